"""studiosr.models-compatible classes whose forward() runs on the MI355X HIP kernels."""
from .common import BaseModule, Model, Upsampler  # noqa: F401
from .edsr import EDSR  # noqa: F401
from .hat import HAT  # noqa: F401
from .rcan import RCAN  # noqa: F401
from .han import HAN  # noqa: F401
from .swinfir import SwinFIR  # noqa: F401
from .swinir import SwinIR  # noqa: F401

__all__ = ["Model", "BaseModule", "Upsampler", "EDSR", "HAN", "HAT", "RCAN", "SwinFIR", "SwinIR"]
