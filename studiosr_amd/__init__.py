"""studiosr_amd -- MI355X (gfx950) native hot path of veritross/studiosr.

`studiosr_amd.models` mirrors `studiosr.models` (same classes, kwargs, state_dict keys, inference API, differentiable forward);
`studiosr_amd.Trainer` / `studiosr_amd.Evaluator` mirror the engine classes;
the math runs in hand-written HIP kernels behind the C ABI of include/studiosr_hip.h.  GPU only: there
is no CPU fallback (use the reference for CPU runs).
"""
from . import _lib, autograd, models, ops, packing, parallel, runtime, strips  # noqa: F401
from .evaluator import Evaluator  # noqa: F401
from .metrics import compute_psnr, compute_ssim  # noqa: F401
from .models import EDSR, HAN, HAT, RCAN, SwinFIR, SwinIR  # noqa: F401
from .trainer import Trainer  # noqa: F401

__version__ = "0.1.0"
