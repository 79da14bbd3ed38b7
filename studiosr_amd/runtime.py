"""Small host-side runtime shared by the model drivers: workspace buffers, precision policy,
HIP-graph capture of a whole forward pass (the launch-bound regime: ~170 kernels per SwinIR
forward at ~3 us each would otherwise be host-bound)."""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch

from . import _lib as L

Tensor = torch.Tensor


def require_device(x: Tensor) -> None:
    if not x.is_cuda:
        raise L.HipLibraryError(
            "studiosr_amd models run only on an MI355X (ROCm) device tensor: there is no CPU fallback. "
            "Move the model and input with .to('cuda'), or use the reference / oracle for CPU runs."
        )
    L.lib()  # raises if libstudiosr_hip.so is missing


def compute_dtype(precision: str) -> torch.dtype:
    """'fp32' (reference inference semantics, exact-fp32 MFMA), 'bf16' (bf16 operands, fp32 accumulate,
    fp32 residual stream / LayerNorm / softmax), or 'auto' = bf16 under torch.autocast(bfloat16) as in the
    reference Trainer (studiosr/engine/trainer.py:80,102), fp32 otherwise."""
    if precision == "auto":
        if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            return torch.bfloat16
        return torch.float32
    if precision == "bf16":
        return torch.bfloat16
    if precision == "fp32":
        return torch.float32
    raise ValueError(f"precision must be 'auto', 'fp32' or 'bf16', got {precision!r}")


def sr_dtype(dt: torch.dtype) -> int:
    return L.SR_BF16 if dt == torch.bfloat16 else L.SR_F32


class Workspace:
    """Named device buffers, reused across forwards of the same geometry (graph-capture friendly:
    a captured forward only touches buffers that were allocated before capture)."""

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.bufs: Dict[Tuple, Tensor] = {}

    def get(self, name: str, shape, dtype: torch.dtype) -> Tensor:
        key = (name, tuple(shape), dtype)
        t = self.bufs.get(key)
        if t is None:
            t = torch.zeros(tuple(shape), dtype=dtype, device=self.device)
            self.bufs[key] = t
        return t


class GraphedForward:
    """Capture `fn(static_input) -> static_output` once into a HIP graph and replay it.

    fn must only enqueue work on the current stream (all studiosr_amd ops do) and use workspace buffers
    that already exist (one eager warm-up call is made before capture to allocate them)."""

    def __init__(self, fn: Callable[[Tensor], Tensor], example: Tensor, warmup: int = 2) -> None:
        self.static_in = example.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                fn(self.static_in)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = fn(self.static_in)

    def __call__(self, x: Tensor) -> Tensor:
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out

    def replay(self) -> Tensor:
        self.graph.replay()
        return self.static_out
