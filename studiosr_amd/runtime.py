"""Small host-side runtime shared by the model drivers: workspace buffers, precision policy,
HIP-graph capture of a whole forward pass (the launch-bound regime: 51 kernels per SwinIR
forward at ~3 us of host time each would otherwise be host-bound)."""
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


_KNOBS: Dict[str, str] = {}


def knob(name: str, default: str) -> str:
    """An A/B environment switch (SR_SWIN_QKV, SR_QKV_FRAG, SR_SWIN_TAIL, SR_TAIL_QKV, ...), read ONCE per forward: the launches of one forward must
    agree on the layouts they hand each other (a tail kernel writes the next block's q / k / v^T in the layout that block's attention launch
    will assume), so a variable that changes mid-forward (A/B tooling does that) must not split them.  Workspace.begin_forward() re-arms the reads."""
    v = _KNOBS.get(name)
    if v is None:
        import os

        v = _KNOBS[name] = os.environ.get(name, default)
    return v


def reset_knobs() -> None:
    """Forget the cached A/B switches (Workspace.begin_forward() does this at the start of every model forward; callers that drive the block
    helpers directly -- tests, tools -- call it after changing the environment)."""
    _KNOBS.clear()


X3_KEY = "bf16x3"  # key of the split-operand weight packing
_x3_depth = 0      # > 0 while a forward with precision "fp32x3" is enqueueing (one forward at a time per process: SURVEY 8b)


class x3_mode:
    """While active, ops.gemm / ops.conv3x3 turn an fp32 compute request into SR_BF16X3 (split-operand bf16 on fp32 tensors)."""

    def __init__(self, on: bool) -> None:
        self.on = on

    def __enter__(self):
        global _x3_depth
        _x3_depth += int(self.on)

    def __exit__(self, *exc):
        global _x3_depth
        _x3_depth -= int(self.on)


def x3_active() -> bool:
    return _x3_depth > 0


def compute_dtype(precision: str) -> torch.dtype:
    """'fp32' (reference inference semantics, exact-fp32 MFMA), 'bf16' (bf16 operands, fp32 accumulate,
    fp32 residual stream / LayerNorm / softmax), 'fp32x3' (fp32 tensors and op order as 'fp32', but every Linear / conv
    contraction runs as hi*hi + hi*lo + lo*hi on the bf16 matrix cores: fp32-class accuracy several times faster), or
    'auto' = bf16 under torch.autocast(bfloat16) as in the reference Trainer (studiosr/engine/trainer.py:80,102), fp32 otherwise."""
    if precision == "fp32x3":
        return torch.float32
    if precision == "auto":
        if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            return torch.bfloat16
        return torch.float32
    if precision == "bf16":
        return torch.bfloat16
    if precision == "fp32":
        return torch.float32
    raise ValueError(f"precision must be 'auto', 'fp32', 'fp32x3' or 'bf16', got {precision!r}")


def sr_dtype(dt: torch.dtype) -> int:
    return L.SR_BF16 if dt == torch.bfloat16 else L.SR_F32


class Workspace:
    """Named device buffers, reused across forwards of the same geometry (graph-capture friendly:
    a captured forward only touches buffers that were allocated before capture).

    The cache is bounded: an evaluation loop over images of many sizes (Set14, Urban100, DIV2K) asks for a new buffer set
    per size, so buffers are kept least-recently-used under a byte budget (SR_WS_BUDGET_MB, default 16 GiB of the 288 GB)
    and the rest is released.  Buffers used by the forward in progress (same epoch) and buffers a HIP graph recorded
    (requested while the stream was capturing: the graph holds their raw pointers) are never evicted."""

    GUARD_BYTES = 1 << 16  # SR_WS_GUARD=1: every buffer sits between two 64 KiB guard zones (see check_guards)

    def __init__(self, device: torch.device, budget_bytes: int = 0) -> None:
        import os

        device = torch.device(device)
        if device.type == "cuda" and device.index is None:  # "cuda" must compare equal to a tensor's "cuda:N" (Model._workspace)
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.bufs: Dict[Tuple, Tensor] = {}
        self.guard = bool(os.environ.get("SR_WS_GUARD"))
        self._raw: Dict[Tuple, Tensor] = {}
        self.budget = budget_bytes or int(os.environ.get("SR_WS_BUDGET_MB", str(16 * 1024))) << 20
        self.epoch = 0
        self._used: Dict[Tuple, int] = {}    # key -> epoch of the last request
        self._pinned: set = set()            # keys recorded by a HIP graph
        self.bytes = 0

    def begin_forward(self) -> None:
        """Called once at the start of every forward: buffers requested from here on belong to this forward."""
        self.epoch += 1
        _KNOBS.clear()

    @staticmethod
    def _nbytes(shape, dtype) -> int:
        n = 1
        for d in shape:
            n *= int(d)
        return n * torch.empty((), dtype=dtype).element_size()

    def _evict(self, need: int) -> None:
        if self.bytes + need <= self.budget:
            return
        for key in sorted((k for k in self.bufs if k not in self._pinned and self._used.get(k, 0) < self.epoch), key=lambda k: self._used.get(k, 0)):
            self.bytes -= self._nbytes(key[1], key[2])
            del self.bufs[key]
            self._raw.pop(key, None)
            self._used.pop(key, None)
            if self.bytes + need <= self.budget:
                break

    def get(self, name: str, shape, dtype: torch.dtype) -> Tensor:
        key = (name, tuple(int(d) for d in shape), dtype)
        t = self.bufs.get(key)
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if t is None:
            nbytes = self._nbytes(key[1], dtype)
            if not capturing:
                self._evict(nbytes)
            if self.guard:
                raw = torch.full((nbytes + 2 * self.GUARD_BYTES,), 0x5A, dtype=torch.uint8, device=self.device)
                raw[self.GUARD_BYTES : self.GUARD_BYTES + nbytes] = 0
                self._raw[key] = raw
                t = raw[self.GUARD_BYTES : self.GUARD_BYTES + nbytes].view(dtype).view(key[1])
            else:
                t = torch.zeros(key[1], dtype=dtype, device=self.device)
            self.bufs[key] = t
            self.bytes += nbytes
        self._used[key] = self.epoch
        if capturing:
            self._pinned.add(key)
        return t

    def release(self) -> None:
        """Drop every buffer no HIP graph depends on (the reference's empty_cache() after inference, common.py:47)."""
        for key in [k for k in self.bufs if k not in self._pinned]:
            self.bytes -= self._nbytes(key[1], key[2])
            del self.bufs[key]
            self._raw.pop(key, None)
            self._used.pop(key, None)

    def check_guards(self):
        """Debug aid: names of buffers whose guard zones were written (out-of-bounds stores by a kernel)."""
        bad = []
        for key, raw in self._raw.items():
            g = self.GUARD_BYTES
            if not bool((raw[:g] == 0x5A).all()) or not bool((raw[-g:] == 0x5A).all()):
                lo = int((raw[:g] != 0x5A).sum())
                hi = int((raw[-g:] != 0x5A).sum())
                bad.append((key, lo, hi))
        return bad


class WorkspaceView:
    """A Workspace seen through a name prefix: two concurrent sub-forwards (the two half batches of RCAN.forward) get disjoint buffers of
    the same names and shapes from one bounded workspace."""

    def __init__(self, ws: Workspace, prefix: str) -> None:
        self.ws, self.prefix, self.device = ws, prefix, ws.device

    def get(self, name: str, shape, dtype: torch.dtype) -> Tensor:
        return self.ws.get(self.prefix + name, shape, dtype)


_capture_warmup = 0  # > 0 while GraphedForward runs its eager warm-up calls


def capturing_or_warming_up() -> bool:
    """True inside a HIP-graph capture AND inside GraphedForward's eager warm-up: a forward that takes another launch sequence under capture
    (part batches on several streams) must take it in the warm-up too, so that its workspace buffers exist before the capture begins
    (a buffer first requested inside a capture is a zero-fill node replayed with the graph, in the graph's private pool)."""
    return _capture_warmup > 0 or torch.cuda.is_current_stream_capturing()


class GraphedForward:
    """Capture `fn(static_input) -> static_output` once into a HIP graph and replay it.

    fn must only enqueue work on the current stream (all studiosr_amd ops do) and use workspace buffers
    that already exist (one eager warm-up call is made before capture to allocate them)."""

    def __init__(self, fn: Callable[[Tensor], Tensor], example: Tensor, warmup: int = 2) -> None:
        # The graph records raw device pointers: everything `fn` closes over (model, workspace buffers, packed weights)
        # must outlive the graph, so the callable itself is kept alive here.
        self.fn = fn
        self.static_in = example.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        global _capture_warmup
        _capture_warmup += 1
        try:
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    fn(self.static_in)
        finally:
            _capture_warmup -= 1
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
