"""Differentiable ops of the training path: every forward AND backward is a launch of the HIP training engine
(studiosr_amd/csrc/sr_train.hip through the C ABI); torch supplies device memory, the autograd graph and the RNG stream.

The op set is the closed set the reference's training step reaches (studiosr/engine/trainer.py:97-109):
nn.Linear, 3x3 / 1x1 nn.Conv2d, nn.LayerNorm, GELU / ReLU / LeakyReLU / Sigmoid, window partition / reverse with the cyclic
shift, (shifted-)window attention with the relative-position bias table and the -100 mask, HAT's overlapping cross attention
(nn.Unfold), AdaptiveAvgPool2d(1) + the channel-attention gate, nn.PixelShuffle, DropPath, residual adds and the final
un-normalise + crop.  Tensors are fp32, NHWC / token-major and unpadded; parameters are used and their gradients produced in the
reference's state_dict layouts, so torch.optim and DistributedDataParallel see ordinary `.grad`s.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Tuple

import torch

from . import _lib as L

Tensor = torch.Tensor
Fn = torch.autograd.Function


_LAUNCH_TLS = threading.local()  # device of the tensors of the op being launched on this thread


def _st():
    """Stream handle for a C-ABI launch: the current stream OF THE DEVICE THE OPERANDS LIVE ON (recorded by _chk), not of whatever device
    the calling thread happens to have current."""
    dev = getattr(_LAUNCH_TLS, "dev", None)
    return (torch.cuda.current_stream(dev) if dev is not None else torch.cuda.current_stream()).cuda_stream


def _chk(t: Tensor) -> Tensor:
    if not t.is_cuda:
        raise L.HipLibraryError("studiosr_amd training ops need ROCm device tensors (there is no CPU path)")
    # kernels launch on the current HIP device: make it the operands' device (autograd's worker thread of device i normally has it
    # current already, Model.__call__ pins it in forward; a model on cuda:3 driven from a thread whose current device is 0 would
    # otherwise launch on the wrong device's stream)
    if t.device.index != torch.cuda.current_device():
        torch.cuda.set_device(t.device)
    _LAUNCH_TLS.dev = t.device
    if t.dtype != torch.float32:
        raise TypeError(f"training ops are fp32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------- raw launches
def bgemm(A: Tensor, B: Tensor, Cc: Tensor, M: int, N: int, K: int, sa: Tuple[int, int], sb: Tuple[int, int], sc: Tuple[int, int], *,
          a_off: int = 0, b_off: int = 0, c_off: int = 0, bias: Optional[Tensor] = None, alpha: float = 1.0, accumulate: bool = False, ksplit: int = 1,
          nb: Tuple[int, int] = (1, 1), sab=(0, 0), sbb=(0, 0), scb=(0, 0), bf16_ok: bool = True) -> None:
    """C[b][m,n] (=|+=) alpha * sum_k A[b][m,k] B[b][k,n] + bias[n]; strides in elements, offsets in elements from data_ptr.
    Under torch.autocast(bfloat16) (the reference Trainer's context) large contractions round their operands to bf16 and run on the
    bf16 matrix cores with fp32 accumulation, exactly the reference's autocast contract; `bf16_ok=False` pins a call to exact fp32
    (the DFT matrices of SwinFIR: torch.fft is not autocast to bf16 either)."""
    # the fp32 kernels carry (batch x ksplit) on grid.z (<= 65535): chunk the first batch level beyond that (SwinIR at batch 64 of
    # 128 x 128 patches is nb = (16,384, 6); a grad-enabled eval forward of an 830 x 830 image likewise)
    if nb[0] * nb[1] * ksplit > 65535 and nb[0] > 1:
        step = max(1, 65535 // (nb[1] * ksplit))
        for b0 in range(0, nb[0], step):
            n1 = min(step, nb[0] - b0)
            bgemm(A, B, Cc, M, N, K, sa, sb, sc, a_off=a_off + b0 * sab[0], b_off=b_off + b0 * sbb[0], c_off=c_off + b0 * scb[0], bias=bias, alpha=alpha,
                  accumulate=accumulate, ksplit=ksplit, nb=(n1, nb[1]), sab=sab, sbb=sbb, scb=scb, bf16_ok=bf16_ok)
        return
    g = L.SrBgemm()
    g.A, g.B, g.C = A.data_ptr() + 4 * a_off, B.data_ptr() + 4 * b_off, Cc.data_ptr() + 4 * c_off
    g.bias = None if bias is None else bias.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.sa_m, g.sa_k = sa
    g.sb_k, g.sb_n = sb
    g.sc_m, g.sc_n = sc
    g.nb1, g.nb2 = nb
    g.sa_b1, g.sa_b2 = sab
    g.sb_b1, g.sb_b2 = sbb
    g.sc_b1, g.sc_b2 = scb
    g.alpha, g.accumulate, g.ksplit = float(alpha), int(accumulate), int(ksplit)
    g.compute_dtype = L.SR_BF16 if (bf16_ok and _autocast_bf16()) else L.SR_F32
    L.check(L.lib().sr_bgemm(C.byref(g), _st()), "sr_bgemm")


_AUTOCAST_STATE = threading.local()  # per thread: autograd runs backward nodes on its own worker threads


def _autocast_bf16() -> bool:
    """bf16 autocast active for THIS launch: read live in forward; backward runs outside the context manager, so every Function records
    the state it was built under (torch does the same: autocast backward ops run in the dtype their forward ran in)."""
    return getattr(_AUTOCAST_STATE, "on", False)


class autocast_state:
    """with autocast_state(flag): ... -- sets what bgemm() sees.  Functions wrap forward (flag = torch state) and backward (flag = saved)."""

    def __init__(self, flag: bool) -> None:
        self.flag = flag

    def __enter__(self):
        self.prev = getattr(_AUTOCAST_STATE, "on", False)
        _AUTOCAST_STATE.on = bool(self.flag)

    def __exit__(self, *exc):
        _AUTOCAST_STATE.on = self.prev


def torch_autocast_bf16() -> bool:
    """What a Function records at forward time: torch's bf16 autocast context, or an enclosing autocast_state(True) (the bf16 inference
    paths of SwinFIR / HAN run their generic-engine modules under it without touching torch's autocast)."""
    return _autocast_bf16() or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def _ksplit(rows: int, cols: int, K: int) -> int:
    """Workgroups for a weight-gradient GEMM whose contraction runs over K tokens: enough slices to fill 256 CUs, each >= 256 long."""
    tiles = ((rows + 63) // 64) * ((cols + 63) // 64)
    ks = max(1, min(1024 // max(tiles, 1), (K + 255) // 256, 512))
    if ks >= 2:  # token counts are B * H * W: the nearest power-of-two slice count usually divides them (wgrad's atomic-free path)
        p2 = 1 << (ks.bit_length() - 1)
        if ks * ks > 2 * p2 * p2:
            p2 *= 2
        if K % p2 == 0 and (K // p2) % 32 == 0:
            return p2
    return ks


def wgrad(dy: Tensor, x: Tensor, rows: int, cols: int, T: int) -> Tensor:
    """dW[rows, cols] = dy[T, rows]^T x[T, cols]: the contraction runs over the T tokens.  Split over token slices so that the launch fills
    the chip; when T divides into equal slices each slice is one batch entry of a plain GEMM into a scratch stack that sr_batch_sum adds
    up -- no atomics (each workgroup's result tile goes out in 16-byte stores; same-address fp32 atomics cost ~14 us of a 55-us launch) and
    a deterministic gradient; otherwise split-K with atomics into a zeroed accumulator."""
    ks = _ksplit(rows, cols, T)
    chunk = T // ks if ks > 1 and T % ks == 0 else 0
    if chunk and chunk % 32 == 0:
        stack = torch.empty(ks, rows, cols, device=dy.device, dtype=torch.float32)
        bgemm(dy, x, stack, rows, cols, chunk, (1, rows), (cols, 1), (cols, 1), nb=(ks, 1), sab=(chunk * rows, 0), sbb=(chunk * cols, 0), scb=(rows * cols, 0))
        dw = torch.empty(rows, cols, device=dy.device, dtype=torch.float32)
        L.check(L.lib().sr_batch_sum(stack.data_ptr(), dw.data_ptr(), ks, rows * cols, rows * cols, _st()), "sr_batch_sum")
        return dw
    dw = _zeros((rows, cols), dy.device)
    bgemm(dy, x, dw, rows, cols, T, (1, rows), (cols, 1), (cols, 1), ksplit=ks)
    return dw


def eltwise(op: int, x: Optional[Tensor], out: Tensor, *, y: Optional[Tensor] = None, s: Optional[Tensor] = None, inner: int = 1, Cn: int = 1, a: float = 0.0, b: float = 0.0) -> Tensor:
    L.check(L.lib().sr_eltwise(op, None if x is None else x.data_ptr(), None if y is None else y.data_ptr(), None if s is None else s.data_ptr(), out.data_ptr(),
                               out.numel(), inner, Cn, float(a), float(b), _st()), "sr_eltwise")
    return out


def colsum(x: Tensor, out: Tensor, nb: int, P: int, Cn: int, alpha: float = 1.0) -> Tensor:
    """out[b][c] += alpha * sum_p x[b][p][c]"""
    L.check(L.lib().sr_colsum(x.data_ptr(), out.data_ptr(), nb, P, Cn, float(alpha), _st()), "sr_colsum")
    return out


class _ZeroArena:
    """Zero-initialised fp32 scratch for the accumulators of a backward pass (split-K weight gradients, bias / LayerNorm / bias-table
    gradients that kernels add into): one fill per 32 MiB chunk instead of one `torch.zeros` launch per accumulator (a HAT step made
    ~1,400 of them, 5.6 ms of 4-us fills).  Slices are views of the chunk; a chunk lives as long as any of its views (e.g. a
    parameter's .grad until zero_grad) and is never handed out twice."""

    CHUNK = 8 << 20  # floats

    def __init__(self) -> None:
        self.buf: Optional[Tensor] = None
        self.off = 0
        self.stream = None

    def take(self, shape, device) -> Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        if n == 0 or n > self.CHUNK // 8 or torch.cuda.is_current_stream_capturing():
            # under HIP-graph capture the chunk's fill would not be part of the graph: every replay would accumulate onto the last result
            return torch.zeros(tuple(shape), device=device, dtype=torch.float32)
        dev = torch.device(device)
        stream = torch.cuda.current_stream(dev)
        if self.buf is None or self.buf.device != dev or self.stream != stream or self.off + n > self.CHUNK:
            # one chunk per (device, stream): the fill and every kernel that adds into a slice are ordered on that stream
            self.buf, self.off, self.stream = torch.zeros(self.CHUNK, device=dev, dtype=torch.float32), 0, stream
        v = self.buf[self.off:self.off + n].view(tuple(shape))
        self.off += (n + 63) // 64 * 64  # 256-byte granules
        return v


_ARENA_TLS = threading.local()  # one arena per thread (autograd's backward worker of a device)


def _arena() -> _ZeroArena:
    a = getattr(_ARENA_TLS, "arena", None)
    if a is None:
        a = _ARENA_TLS.arena = _ZeroArena()
    return a


def _zeros(shape, device) -> Tensor:
    return _arena().take(shape, device)


def _zeros_like(t: Tensor) -> Tensor:
    return _arena().take(t.shape, t.device)


# --------------------------------------------------------------------------- nn.Linear / conv
# --------------------------------------------------------------------------- independent backward work on a side stream
_SIDE_STREAMS = {}
_OVERLAP = False  # (the SR_WGRAD_STREAM switch left in round 5) off: measured 73.4 -> 81-84 ms per HAT step (two cross-stream event waits per layer cost more than the overlap buys)


class _Side:
    """`with _Side(t) as sd:` -- launches inside the block go to a per-device SIDE stream, ordered after everything enqueued on the current
    (main) stream so far; `sd.join(*tensors)` after the block makes the main stream wait for them and hands the tensors they produced to
    it.  A training step at the reference's per-rank batch (4 x 64 x 64: 16,384 tokens) is ~5,000 launches of 10-50 us that each fill a
    fraction of the chip; the weight / bias / bias-table gradients of a layer do not feed its data gradient, so they can run beside it.
    The join happens INSIDE the same backward: what autograd (AccumulateGrad, DDP's bucket hooks) sees is complete on the main stream.
    EXPERIMENT, off by default (SR_WGRAD_STREAM=1 enables it): on one MI355X the HAT x4 step went from 73.4 to 81-84 ms -- every fork / join is a
    pair of cross-stream event waits (a barrier packet each, ~5-10 us), ~600 per step, more than the overlapped kernels save."""

    def __init__(self, t: Tensor, enable: bool = True) -> None:
        self.on = _OVERLAP and enable and t.is_cuda and not torch.cuda.is_current_stream_capturing()
        self.dev = t.device

    def __enter__(self):
        if self.on:
            self.main = torch.cuda.current_stream(self.dev)
            key = self.dev.index
            side = _SIDE_STREAMS.get(key)
            if side is None:
                side = _SIDE_STREAMS[key] = torch.cuda.Stream(device=self.dev)
            self.side = side
            ready = torch.cuda.Event()
            ready.record(self.main)
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
            side.wait_event(ready)
        return self

    def __exit__(self, *exc):
        if self.on:
            self.done = torch.cuda.Event()
            self.done.record(self.side)
            self.ctx.__exit__(*exc)

    def join(self, *tensors) -> None:
        if self.on:
            self.main.wait_event(self.done)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.main)  # allocated under the side stream, consumed (and freed) on the main stream


class _Linear(Fn):
    """y[M,N] = x[M,K] @ w[N,K]^T + b (swinir.py:69-71, common.py:184-195; 1x1 convs of the channel attention, common.py:161-167)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.ac = torch_autocast_bf16()
        with autocast_state(ctx.ac):
            x, w = _chk(x), _chk(w)
            K = x.shape[-1]
            M = x.numel() // K
            N = w.shape[0]
            y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
            bgemm(x, w, y, M, N, K, (K, 1), (1, K), (N, 1), bias=None if b is None else _chk(b))
            ctx.save_for_backward(x, w)
            ctx.has_bias = b is not None
            return y

    @staticmethod
    def backward(ctx, dy):
        with autocast_state(ctx.ac):
            x, w = ctx.saved_tensors
            dy = _chk(dy)
            K = x.shape[-1]
            M = x.numel() // K
            N = w.shape[0]
            dx = dw = db = None
            with _Side(dy, ctx.needs_input_grad[0]) as sd:  # weight / bias gradient beside the data gradient
                if ctx.needs_input_grad[1]:
                    dw = wgrad(dy, x, N, K, M).view_as(w)
                if ctx.has_bias and ctx.needs_input_grad[2]:
                    db = colsum(dy, _zeros((N,), dy.device), 1, M, N)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                bgemm(dy, w, dx, M, K, N, (N, 1), (K, 1), (K, 1))
            sd.join(dw, db)
            return dx, dw, db


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    return _Linear.apply(x, w.reshape(w.shape[0], -1), b)


def _im2col(x: Tensor, strides=None, Cn: Optional[int] = None) -> Tensor:
    B, H, W = x.shape[:3]
    Cn = x.shape[3] if Cn is None else Cn
    sb, sy, sx, sc = strides if strides is not None else (H * W * Cn, W * Cn, Cn, 1)
    col = torch.empty(B * H * W, 9 * Cn, device=x.device, dtype=torch.float32)
    L.check(L.lib().sr_im2col3x3(x.data_ptr(), col.data_ptr(), B, H, W, Cn, sb, sy, sx, sc, _st()), "sr_im2col3x3")
    return col


class _Conv3x3(Fn):
    """nn.Conv2d(Cin, Cout, 3, padding=1) on NHWC (common.py:104-105): y = im2col(x) @ w2^T + b with the column order (tap, c) and
    w2 = the weight permuted to [Cout, 3, 3, Cin] (a 1 MB layout copy; the gradient is permuted back).  The column buffer is rebuilt in
    backward instead of being kept (a 256x256 conv_last buffer is 0.6 GB)."""

    @staticmethod
    def forward(ctx, x, w, b, cin):
        ctx.ac = torch_autocast_bf16()
        with autocast_state(ctx.ac):
            x, w = _chk(x), _chk(w)  # w: [Cout, Cin, 3, 3]
            B, H, W = x.shape[:3]
            ld = x.shape[3]  # cin < ld: the input is a channel-padded buffer (the ingest kernel's NHWC-32 image)
            Cout = w.shape[0]
            w2 = w.permute(0, 2, 3, 1).contiguous()  # [Cout, (ky, kx), Cin]
            col = _im2col(x, (H * W * ld, W * ld, ld, 1), cin)
            y = torch.empty(B, H, W, Cout, device=x.device, dtype=torch.float32)
            bgemm(col, w2, y, B * H * W, Cout, 9 * cin, (9 * cin, 1), (1, 9 * cin), (Cout, 1), bias=None if b is None else _chk(b))
            ctx.save_for_backward(x, w2)
            ctx.cin, ctx.has_bias = cin, b is not None
            return y

    @staticmethod
    def backward(ctx, dy):
        with autocast_state(ctx.ac):
            x, w2 = ctx.saved_tensors
            dy = _chk(dy)
            B, H, W = x.shape[:3]
            ld, cin = x.shape[3], ctx.cin
            Cout, M, K = w2.shape[0], B * H * W, 9 * cin
            dx = dw = db = None
            with _Side(dy, ctx.needs_input_grad[0]) as sd:  # im2col + weight gradient + bias gradient beside the data gradient
                if ctx.needs_input_grad[1]:
                    col = _im2col(x, (H * W * ld, W * ld, ld, 1), cin)
                    dw2 = wgrad(dy, col, Cout, K, M)
                    dw = dw2.view(Cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous()  # back to [Cout, Cin, 3, 3]
                if ctx.has_bias and ctx.needs_input_grad[2]:
                    db = colsum(dy, _zeros((Cout,), dy.device), 1, M, Cout)
            if ctx.needs_input_grad[0]:
                assert ld == cin, "no input gradient through a channel-padded input buffer"
                dcol = torch.empty(M, K, device=dy.device, dtype=torch.float32)
                bgemm(dy, w2, dcol, M, K, Cout, (Cout, 1), (K, 1), (K, 1))
                dx = torch.empty_like(x)
                L.check(L.lib().sr_col2im3x3(dcol.data_ptr(), dx.data_ptr(), B, H, W, cin, _st()), "sr_col2im3x3")
            sd.join(dw, db)
            return dx, dw, db, None


def conv3x3(x: Tensor, w: Tensor, b: Optional[Tensor], cin: Optional[int] = None) -> Tensor:
    """x [B,H,W,C] -> [B,H,W,Cout]; w is the nn.Conv2d weight [Cout, Cin, 3, 3]."""
    cin = w.shape[1] if cin is None else cin
    return _Conv3x3.apply(x, w, b, cin)


# --------------------------------------------------------------------------- LayerNorm
class _LayerNorm(Fn):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        x = _chk(x)
        Cn = x.shape[-1]
        M = x.numel() // Cn
        y = torch.empty_like(x)
        stats = torch.empty(M, 2, device=x.device, dtype=torch.float32)
        L.check(L.lib().sr_layernorm_fwd_train(x.data_ptr(), _chk(g).data_ptr(), _chk(b).data_ptr(), y.data_ptr(), stats.data_ptr(), M, Cn, float(eps), _st()), "sr_layernorm_fwd_train")
        ctx.save_for_backward(x, g, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, stats = ctx.saved_tensors
        dy = _chk(dy)
        Cn = x.shape[-1]
        M = x.numel() // Cn
        dx = torch.empty_like(x)
        dg, db = _zeros_like(g), _zeros_like(g)
        L.check(L.lib().sr_layernorm_bwd(x.data_ptr(), stats.data_ptr(), g.data_ptr(), dy.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), M, Cn, _st()), "sr_layernorm_bwd")
        return dx, dg, db, None


def layer_norm(x: Tensor, g: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    return _LayerNorm.apply(x, g, b, eps)


# --------------------------------------------------------------------------- elementwise
class _Act(Fn):
    FWD = {"gelu": L.EW_GELU_FWD, "relu": L.EW_RELU_FWD, "lrelu": L.EW_LRELU_FWD, "sigmoid": L.EW_SIGMOID_FWD}
    BWD = {"gelu": L.EW_GELU_BWD, "relu": L.EW_RELU_BWD, "lrelu": L.EW_LRELU_BWD, "sigmoid": L.EW_SIGMOID_BWD}

    @staticmethod
    def forward(ctx, x, kind, slope):
        x = _chk(x)
        y = eltwise(_Act.FWD[kind], x, torch.empty_like(x), a=slope)
        ctx.save_for_backward(x if kind in ("gelu", "lrelu", "relu") else y)  # sigmoid's derivative is written in its output
        ctx.kind, ctx.slope = kind, slope
        return y

    @staticmethod
    def backward(ctx, dy):
        (t,) = ctx.saved_tensors
        dy = _chk(dy)
        return eltwise(_Act.BWD[ctx.kind], dy, torch.empty_like(dy), y=t, a=ctx.slope), None, None


def gelu(x):
    return _Act.apply(x, "gelu", 0.0)


def relu(x):
    return _Act.apply(x, "relu", 0.0)


def leaky_relu(x, slope: float = 0.01):
    return _Act.apply(x, "lrelu", slope)


def sigmoid(x):
    return _Act.apply(x, "sigmoid", 0.0)


class _Axpby(Fn):
    @staticmethod
    def forward(ctx, x, y, a, b):
        x, y = _chk(x), _chk(y)
        assert x.shape == y.shape
        ctx.a, ctx.b = a, b
        return eltwise(L.EW_AXPBY, x, torch.empty_like(x), y=y, a=a, b=b)

    @staticmethod
    def backward(ctx, d):
        d = _chk(d)
        # a coefficient of 1 (every plain residual add) passes the incoming gradient through: no launch, no copy (as torch's own add)
        scaled = lambda c: d if c == 1.0 else eltwise(L.EW_AXPBY, d, torch.empty_like(d), a=c)  # noqa: E731
        dx = scaled(ctx.a) if ctx.needs_input_grad[0] else None
        dy = scaled(ctx.b) if ctx.needs_input_grad[1] else None
        return dx, dy, None, None


def add(x: Tensor, y: Tensor, a: float = 1.0, b: float = 1.0) -> Tensor:
    """a*x + b*y (residual adds, res_scale: common.py:152, hat.py:192)."""
    return _Axpby.apply(x, y, a, b)


class _ScaleSample(Fn):
    @staticmethod
    def forward(ctx, x, s):
        x = _chk(x)
        ctx.save_for_backward(s)
        return eltwise(L.EW_SCALE_SAMPLE, x, torch.empty_like(x), s=s, inner=x.numel() // x.shape[0])

    @staticmethod
    def backward(ctx, d):
        (s,) = ctx.saved_tensors
        d = _chk(d)
        return eltwise(L.EW_SCALE_SAMPLE, d, torch.empty_like(d), s=s, inner=d.numel() // d.shape[0]), None


def drop_path(x: Tensor, p: float, training: bool) -> Tensor:
    """timm's DropPath(drop_prob, scale_by_keep=True) (swinir.py:7,137,171-172; hat.py:148,192-193): per-sample Bernoulli(keep) / keep
    in training, identity otherwise.  The mask comes from torch's device RNG (same distribution as the reference's bernoulli_)."""
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    mask = torch.empty(x.shape[0], device=x.device, dtype=torch.float32).bernoulli_(keep)
    if keep > 0.0:
        mask.div_(keep)
    return _ScaleSample.apply(x, mask)


# --------------------------------------------------------------------------- index maps
class _WindowCopy(Fn):
    @staticmethod
    def forward(ctx, x, ws, shift, to_windows, shape):
        x = _chk(x)
        B, H, W, Cn = shape
        ctx.meta = (ws, shift, to_windows, shape)
        out = torch.empty((B * (H // ws) * (W // ws) * ws * ws, Cn) if to_windows else (B, H, W, Cn), device=x.device, dtype=torch.float32)
        L.check(L.lib().sr_window_copy(x.data_ptr(), out.data_ptr(), B, H, W, Cn, ws, shift, int(to_windows), _st()), "sr_window_copy")
        return out

    @staticmethod
    def backward(ctx, d):
        ws, shift, to_windows, shape = ctx.meta
        return _WindowCopy.apply(d, ws, shift, not to_windows, shape), None, None, None, None


def window_partition(x: Tensor, ws: int, shift: int) -> Tensor:
    """[B,H,W,C] -> window-order tokens [B*nW*ws*ws, C] = window_partition(roll(x, (-shift, -shift))) (swinir.py:154-158)."""
    return _WindowCopy.apply(x, ws, shift, True, tuple(x.shape))


def window_reverse(t: Tensor, ws: int, shift: int, shape) -> Tensor:
    """inverse of window_partition: roll(window_reverse(t), (+shift, +shift)) (swinir.py:164-168)."""
    return _WindowCopy.apply(t, ws, shift, False, tuple(shape))


class _OcaUnfold(Fn):
    @staticmethod
    def forward(ctx, x, ws, wse):
        x = _chk(x)
        B, H, W, Cn = x.shape
        ctx.meta = (ws, wse, tuple(x.shape))
        out = torch.empty(B * (H // ws) * (W // ws) * wse * wse, Cn, device=x.device, dtype=torch.float32)
        L.check(L.lib().sr_oca_unfold(x.data_ptr(), out.data_ptr(), B, H, W, Cn, ws, wse, 1, _st()), "sr_oca_unfold")
        return out

    @staticmethod
    def backward(ctx, d):
        ws, wse, (B, H, W, Cn) = ctx.meta
        d = _chk(d)
        dx = torch.empty(B, H, W, Cn, device=d.device, dtype=torch.float32)
        L.check(L.lib().sr_oca_unfold(dx.data_ptr(), d.data_ptr(), B, H, W, Cn, ws, wse, 0, _st()), "sr_oca_unfold")
        return dx, None, None


def oca_unfold(x: Tensor, ws: int, wse: int) -> Tensor:
    """nn.Unfold(kernel=wse, stride=ws, padding=(wse-ws)//2) as window-order tokens [B*nW*wse*wse, C] (hat.py:217-221,255-263)."""
    return _OcaUnfold.apply(x, ws, wse)


class _PixelShuffle(Fn):
    @staticmethod
    def forward(ctx, x, r, forward_dir):
        x = _chk(x)
        B, H, W, Cn = x.shape
        ctx.r, ctx.fwd = r, forward_dir
        if forward_dir:
            out = torch.empty(B, H * r, W * r, Cn // (r * r), device=x.device, dtype=torch.float32)
            L.check(L.lib().sr_pixel_shuffle_nhwc(x.data_ptr(), out.data_ptr(), B, H, W, Cn // (r * r), r, 1, _st()), "sr_pixel_shuffle_nhwc")
        else:
            out = torch.empty(B, H // r, W // r, Cn * r * r, device=x.device, dtype=torch.float32)
            L.check(L.lib().sr_pixel_shuffle_nhwc(x.data_ptr(), out.data_ptr(), B, H // r, W // r, Cn, r, 0, _st()), "sr_pixel_shuffle_nhwc")
        return out

    @staticmethod
    def backward(ctx, d):
        return _PixelShuffle.apply(d, ctx.r, not ctx.fwd), None, None


def pixel_shuffle(x: Tensor, r: int) -> Tensor:
    """nn.PixelShuffle(r) on NHWC: out[b, y*r+i, x*r+j, c] = in[b, y, x, c*r*r + i*r + j] (common.py:129,133,136)."""
    return _PixelShuffle.apply(x, r, True)


# --------------------------------------------------------------------------- attention
def _gather_bias(table: Tensor, rpi: Tensor, heads: int) -> Tensor:
    NN = rpi.numel()
    bias = torch.empty(heads, NN, device=table.device, dtype=torch.float32)
    L.check(L.lib().sr_bias_gather(table.data_ptr(), rpi.data_ptr(), bias.data_ptr(), None, table.shape[0], heads, NN, 1, _st()), "sr_bias_gather")
    return bias


class _Attention(Fn):
    """softmax(scale * q k^T + table[rpi] + mask) v per (window, head) (swinir.py:83-102; hat.py:90-107; OCAB hat.py:266-283).
    q is the column slice [q_off, q_off + C) of the token-major tensor `qs` (Nq consecutive rows per window), k and v are slices of
    `kvs` (Nk rows per window); `qs` and `kvs` may be ONE packed qkv tensor (`same`), which then gets ONE packed gradient."""

    @staticmethod
    def forward(ctx, qs, kvs, table, rpi, mask, meta):
        ctx.ac = torch_autocast_bf16()
        with autocast_state(ctx.ac):
            heads, Nq, Nk, Cn, scale, q_off, k_off, v_off, same = meta
            qs = _chk(qs)
            kvs = qs if same else _chk(kvs)
            table = _chk(table)
            hd = Cn // heads
            ldq, ldk = qs.shape[-1], kvs.shape[-1]
            nbw = qs.numel() // ldq // Nq
            dev = qs.device
            P = torch.empty(nbw, heads, Nq, Nk, device=dev, dtype=torch.float32)
            nbh = (nbw, heads)
            bgemm(qs, kvs, P, Nq, Nk, hd, (ldq, 1), (1, ldk), (Nk, 1), a_off=q_off, b_off=k_off, alpha=scale, nb=nbh,
                  sab=(Nq * ldq, hd), sbb=(Nk * ldk, hd), scb=(heads * Nq * Nk, Nq * Nk))
            bias = _gather_bias(table, rpi, heads)
            nW = mask.shape[0] if mask is not None else 0
            L.check(L.lib().sr_softmax_fwd(P.data_ptr(), bias.data_ptr(), None if mask is None else mask.data_ptr(), nbw * heads * Nq, heads, Nq, Nk, nW, _st()), "sr_softmax_fwd")
            O = torch.empty(nbw * Nq, Cn, device=dev, dtype=torch.float32)
            bgemm(P, kvs, O, Nq, hd, Nk, (Nk, 1), (ldk, 1), (Cn, 1), b_off=v_off, nb=nbh, sab=(heads * Nq * Nk, Nq * Nk), sbb=(Nk * ldk, hd), scb=(Nq * Cn, hd))
            ctx.save_for_backward(qs, kvs, P, rpi, table)
            ctx.meta = meta
            return O

    @staticmethod
    def backward(ctx, dO):
        with autocast_state(ctx.ac):
            qs, kvs, P, rpi, table = ctx.saved_tensors
            heads, Nq, Nk, Cn, scale, q_off, k_off, v_off, same = ctx.meta
            dO = _chk(dO)
            hd = Cn // heads
            ldq, ldk = qs.shape[-1], kvs.shape[-1]
            nbw = qs.numel() // ldq // Nq
            nbh = (nbw, heads)
            sP = (heads * Nq * Nk, Nq * Nk)
            dev = dO.device
            # columns outside the q / k / v slices receive no gradient: zero-filled unless the three slices tile the packed tensor
            dq = torch.empty_like(qs) if (same and ldq == 3 * Cn) else torch.zeros_like(qs)
            dkv = dq if same else torch.zeros_like(kvs)
            # the q / k / v gradients land in disjoint column slices of the packed gradient: dv runs beside dP -> dS, then dk and the bias-table
            # gradient beside dq (side stream, joined before the tensors are handed back)
            with _Side(dO) as sd1:
                # dv = P^T dO
                bgemm(P, dO, dkv, Nk, hd, Nq, (1, Nk), (Cn, 1), (ldk, 1), c_off=v_off, nb=nbh, sab=sP, sbb=(Nq * Cn, hd), scb=(Nk * ldk, hd))
            # dP = dO v^T
            dP = torch.empty_like(P)
            bgemm(dO, kvs, dP, Nq, Nk, hd, (Cn, 1), (1, ldk), (Nk, 1), b_off=v_off, nb=nbh, sab=(Nq * Cn, hd), sbb=(Nk * ldk, hd), scb=sP)
            # dS = P * (dP - rowsum(dP * P))
            L.check(L.lib().sr_softmax_bwd(P.data_ptr(), dP.data_ptr(), nbw * heads * Nq, Nk, _st()), "sr_softmax_bwd")
            dtable = None
            with _Side(dO) as sd2:  # ordered after dS on the main stream (and after dv on the side stream itself)
                if ctx.needs_input_grad[2]:
                    dbias = torch.empty(heads, Nq * Nk, device=dev, dtype=torch.float32)
                    L.check(L.lib().sr_batch_sum(dP.data_ptr(), dbias.data_ptr(), nbw, heads * Nq * Nk, heads * Nq * Nk, _st()), "sr_batch_sum")
                    dtable = _zeros_like(table)
                    L.check(L.lib().sr_bias_gather(None, rpi.data_ptr(), dbias.data_ptr(), dtable.data_ptr(), table.shape[0], heads, Nq * Nk, 0, _st()), "sr_bias_gather")
                # dk = scale * dS^T q
                bgemm(dP, qs, dkv, Nk, hd, Nq, (1, Nk), (ldq, 1), (ldk, 1), b_off=q_off, c_off=k_off, alpha=scale, nb=nbh, sab=sP, sbb=(Nq * ldq, hd), scb=(Nk * ldk, hd))
            # dq = scale * dS k
            bgemm(dP, kvs, dq, Nq, hd, Nk, (Nk, 1), (ldk, 1), (ldq, 1), b_off=k_off, c_off=q_off, alpha=scale, nb=nbh, sab=sP, sbb=(Nk * ldk, hd), scb=(Nq * ldq, hd))
            sd1.join()
            sd2.join(dtable)
            return dq, (None if same else dkv), dtable, None, None, None


def window_attention_packed(qkv: Tensor, table: Tensor, rpi: Tensor, mask: Optional[Tensor], heads: int, N: int, Cn: int) -> Tensor:
    """qkv [nbw*N, 3C] window-order tokens -> attention output [nbw*N, C] (heads concatenated, swinir.py:102)."""
    hd = Cn // heads
    return _Attention.apply(qkv, None, table, rpi, mask, (heads, N, N, Cn, hd ** -0.5, 0, Cn, 2 * Cn, True))


def cross_window_attention(q_win: Tensor, kv_win: Tensor, table: Tensor, rpi: Tensor, heads: int, Nq: int, Nk: int, Cn: int) -> Tensor:
    """OCAB (hat.py:266-283): q = columns [0, C) of q_win [nbw*Nq, 3C] (window-partitioned qkv), k / v = columns [C, 2C) / [2C, 3C) of
    kv_win [nbw*Nk, 3C] (unfolded qkv)."""
    hd = Cn // heads
    return _Attention.apply(q_win, kv_win, table, rpi, None, (heads, Nq, Nk, Cn, hd ** -0.5, 0, Cn, 2 * Cn, False))


# --------------------------------------------------------------------------- channel attention pieces
class _AvgPool(Fn):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x)
        B, H, W, Cn = x.shape
        ctx.shape = tuple(x.shape)
        return colsum(x, _zeros((B, Cn), x.device), B, H * W, Cn, 1.0 / (H * W))

    @staticmethod
    def backward(ctx, d):
        B, H, W, Cn = ctx.shape
        out = torch.empty(B, H, W, Cn, device=d.device, dtype=torch.float32)
        return eltwise(L.EW_BCAST_BC, None, out, s=_chk(d), inner=H * W * Cn, Cn=Cn, a=1.0 / (H * W))


def avg_pool(x: Tensor) -> Tensor:
    """nn.AdaptiveAvgPool2d(1) on NHWC -> [B, C] (common.py:160, hat.py:31)."""
    return _AvgPool.apply(x)


class _MulBC(Fn):
    @staticmethod
    def forward(ctx, x, s):
        x, s = _chk(x), _chk(s)
        ctx.save_for_backward(x, s)
        B, H, W, Cn = x.shape
        return eltwise(L.EW_MUL_BC, x, torch.empty_like(x), s=s, inner=H * W * Cn, Cn=Cn)

    @staticmethod
    def backward(ctx, d):
        x, s = ctx.saved_tensors
        d = _chk(d)
        B, H, W, Cn = x.shape
        dx = eltwise(L.EW_MUL_BC, d, torch.empty_like(d), s=s, inner=H * W * Cn, Cn=Cn)
        prod = eltwise(L.EW_MUL, d, torch.empty_like(d), y=x)
        ds = colsum(prod, _zeros((B, Cn), d.device), B, H * W, Cn)
        return dx, ds


def mul_bc(x: Tensor, s: Tensor) -> Tensor:
    """x [B,H,W,C] * s [B,C] (the channel-attention gate, common.py:169-170, hat.py:38-39)."""
    return _MulBC.apply(x, s)


def channel_attention(y: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """y * sigmoid(conv1x1(relu(conv1x1(avgpool(y))))) (common.py:156-170; hat.py:25-39)."""
    s = sigmoid(linear(relu(linear(avg_pool(y), w1, b1)), w2, b2))
    return mul_bc(y, s)


# --------------------------------------------------------------------------- model output
class _NhwcOut(Fn):
    @staticmethod
    def forward(ctx, y, scale, shift, Ho, Wo):
        y = _chk(y)
        B, Hs, Ws, Cn = y.shape
        out = torch.empty(B, Cn, Ho, Wo, device=y.device, dtype=torch.float32)
        L.check(L.lib().sr_nhwc_out(y.data_ptr(), out.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, Hs, Ws, Cn, Ho, Wo, 1, _st()), "sr_nhwc_out")
        ctx.save_for_backward(scale, shift)
        ctx.shape = (B, Hs, Ws, Cn, Ho, Wo)
        return out

    @staticmethod
    def backward(ctx, d):
        scale, shift = ctx.saved_tensors
        B, Hs, Ws, Cn, Ho, Wo = ctx.shape
        d = _chk(d)
        dy = torch.empty(B, Hs, Ws, Cn, device=d.device, dtype=torch.float32)
        L.check(L.lib().sr_nhwc_out(d.data_ptr(), dy.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, Hs, Ws, Cn, Ho, Wo, 0, _st()), "sr_nhwc_out")
        return dy, None, None, None, None


def nhwc_out(y: Tensor, scale: Tensor, shift: Tensor, Ho: int, Wo: int) -> Tensor:
    """NHWC features -> the model's NCHW output: y * scale[c] + shift[c], cropped to [Ho, Wo] (common.py:232-233, swinir.py:372)."""
    return _NhwcOut.apply(y, scale, shift, Ho, Wo)


# --------------------------------------------------------------------------- host-side index logic
class _LruCache(dict):
    """Small LRU keyed on geometry (H, W, ...): a grad-enabled evaluation over a dataset of many image sizes must not keep one
    [nW, N, N] mask (268 MB at 1024 x 1024, ws 8) or DFT matrix set per size alive."""

    def __init__(self, max_items: int, max_bytes: int) -> None:
        super().__init__()
        self.max_items, self.max_bytes = max_items, max_bytes

    @staticmethod
    def _nbytes(v) -> int:
        ts = v if isinstance(v, (tuple, list)) else (v,)
        return sum(t.numel() * t.element_size() for t in ts)

    def get(self, key, default=None):
        if key in self:
            v = self.pop(key)
            self[key] = v  # most recently used last
            return v
        return default

    def put(self, key, v) -> None:
        self[key] = v
        while len(self) > 1 and (len(self) > self.max_items or sum(self._nbytes(x) for x in self.values()) > self.max_bytes):
            self.pop(next(iter(self)))


_MASKS = _LruCache(8, 512 << 20)


def shift_mask(H: int, W: int, ws: int, shift: int, device) -> Tensor:
    """calculate_mask (common.py:250-274): [nW, N, N] with -100 where the two tokens of a window carry different region labels."""
    key = (H, W, ws, shift, str(device))
    m = _MASKS.get(key)
    if m is None:
        img = torch.zeros(H, W)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[hs, wsl] = cnt
                cnt += 1
        win = img.reshape(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
        diff = win[:, None, :] - win[:, :, None]
        m = torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff)).contiguous().to(device)
        _MASKS.put(key, m)
    return m


# --------------------------------------------------------------------------- SwinFIR / HAN pieces
def _copy_cols(src: Tensor, dst: Tensor, rows: int, n: int, ld_s: int, off_s: int, ld_d: int, off_d: int, *, s_base: int = 0, d_base: int = 0, accumulate: bool = False) -> None:
    L.check(L.lib().sr_copy_cols(src.data_ptr() + 4 * s_base, dst.data_ptr() + 4 * d_base, rows, n, ld_s, off_s, ld_d, off_d, int(accumulate), _st()), "sr_copy_cols")


class _Concat(Fn):
    """torch.cat along the channel (last) axis of row-major tensors (swinfir.py:25,79; han.py:112)."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [_chk(x) for x in xs]
        widths = [x.shape[-1] for x in xs]
        rows = xs[0].numel() // widths[0]
        out = torch.empty(*xs[0].shape[:-1], sum(widths), device=xs[0].device, dtype=torch.float32)
        off = 0
        for x, w_ in zip(xs, widths):
            assert x.numel() // w_ == rows
            _copy_cols(x, out, rows, w_, w_, 0, sum(widths), off)
            off += w_
        ctx.widths, ctx.shapes = widths, [tuple(x.shape) for x in xs]
        return out

    @staticmethod
    def backward(ctx, d):
        d = _chk(d)
        tot = sum(ctx.widths)
        rows = d.numel() // tot
        outs, off = [], 0
        for w_, shp in zip(ctx.widths, ctx.shapes):
            g = torch.empty(shp, device=d.device, dtype=torch.float32)
            _copy_cols(d, g, rows, w_, tot, off, w_, 0)
            outs.append(g)
            off += w_
        return tuple(outs)


def concat(*xs: Tensor) -> Tensor:
    return _Concat.apply(*xs)


class _Slice(Fn):
    @staticmethod
    def forward(ctx, x, off, n):
        x = _chk(x)
        ld = x.shape[-1]
        rows = x.numel() // ld
        out = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.float32)
        _copy_cols(x, out, rows, n, ld, off, n, 0)
        ctx.meta = (off, n, tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, d):
        off, n, shp = ctx.meta
        d = _chk(d)
        g = torch.zeros(shp, device=d.device, dtype=torch.float32)
        _copy_cols(d, g, d.numel() // n, n, n, 0, shp[-1], off)
        return g, None, None


def slice_channels(x: Tensor, off: int, n: int) -> Tensor:
    """x[..., off : off + n] as a contiguous tensor (torch.split, swinfir.py:30)."""
    return _Slice.apply(x, off, n)


class _ScaleParam(Fn):
    """gamma * x with a learnable scalar gamma (HAN's LAM / CSAM, han.py:16,31,41,50)."""

    @staticmethod
    def forward(ctx, x, gamma):
        x, gamma = _chk(x), _chk(gamma)
        ctx.save_for_backward(x, gamma)
        return eltwise(L.EW_SCALE_SAMPLE, x, torch.empty_like(x), s=gamma, inner=x.numel())

    @staticmethod
    def backward(ctx, d):
        x, gamma = ctx.saved_tensors
        d = _chk(d)
        dx = eltwise(L.EW_SCALE_SAMPLE, d, torch.empty_like(d), s=gamma, inner=d.numel())
        prod = eltwise(L.EW_MUL, d, torch.empty_like(d), y=x)
        dg = colsum(prod, _zeros((1,), d.device), 1, prod.numel(), 1)
        return dx, dg.reshape(gamma.shape)


def scale_param(x: Tensor, gamma: Tensor) -> Tensor:
    return _ScaleParam.apply(x, gamma)


def mul(x: Tensor, y: Tensor) -> Tensor:
    return _Mul.apply(x, y)


class _Mul(Fn):
    @staticmethod
    def forward(ctx, x, y):
        x, y = _chk(x), _chk(y)
        ctx.save_for_backward(x, y)
        return eltwise(L.EW_MUL, x, torch.empty_like(x), y=y)

    @staticmethod
    def backward(ctx, d):
        x, y = ctx.saved_tensors
        d = _chk(d)
        return eltwise(L.EW_MUL, d, torch.empty_like(d), y=y), eltwise(L.EW_MUL, d, torch.empty_like(d), y=x)


class _LayerAttention(Fn):
    """HAN's LAM (han.py:19-33) on x [B, N, D]: softmax(max(E) - E) x with E = x x^T.  The row maximum is a constant shift of a softmax
    argument, so this is softmax(-E) x: a plain self-attention with scale -1 whose contraction runs over D = C*H*W (split over workgroups)."""

    @staticmethod
    def forward(ctx, x):
        ctx.ac = torch_autocast_bf16()
        with autocast_state(ctx.ac):
            x = _chk(x)
            B, N, D = x.shape
            E = torch.zeros(B, N, N, device=x.device, dtype=torch.float32)
            bgemm(x, x, E, N, N, D, (D, 1), (1, D), (N, 1), alpha=-1.0, nb=(B, 1), sab=(N * D, 0), sbb=(N * D, 0), scb=(N * N, 0), ksplit=_ksplit(N, N, D))
            L.check(L.lib().sr_softmax_fwd(E.data_ptr(), None, None, B * N, 1, N, N, 0, _st()), "sr_softmax_fwd")
            out = torch.empty_like(x)
            bgemm(E, x, out, N, D, N, (N, 1), (D, 1), (D, 1), nb=(B, 1), sab=(N * N, 0), sbb=(N * D, 0), scb=(N * D, 0))
            ctx.save_for_backward(x, E)
            return out

    @staticmethod
    def backward(ctx, dO):
        with autocast_state(ctx.ac):
            x, P = ctx.saved_tensors
            dO = _chk(dO)
            B, N, D = x.shape
            nb, sx, sp = (B, 1), (N * D, 0), (N * N, 0)
            dP = torch.zeros_like(P)
            bgemm(dO, x, dP, N, N, D, (D, 1), (1, D), (N, 1), nb=nb, sab=sx, sbb=sx, scb=sp, ksplit=_ksplit(N, N, D))
            dx = torch.empty_like(x)
            bgemm(P, dO, dx, N, D, N, (1, N), (D, 1), (D, 1), nb=nb, sab=sp, sbb=sx, scb=sx)  # P^T dO
            L.check(L.lib().sr_softmax_bwd(P.data_ptr(), dP.data_ptr(), B * N, N, _st()), "sr_softmax_bwd")
            bgemm(dP, x, dx, N, D, N, (N, 1), (D, 1), (D, 1), alpha=-1.0, accumulate=True, nb=nb, sab=sp, sbb=sx, scb=sx)   # -(dS x)
            bgemm(dP, x, dx, N, D, N, (1, N), (D, 1), (D, 1), alpha=-1.0, accumulate=True, nb=nb, sab=sp, sbb=sx, scb=sx)   # -(dS^T x)
            return dx


def layer_attention(x: Tensor) -> Tensor:
    return _LayerAttention.apply(x)


class _Conv3d27(Fn):
    """nn.Conv3d(1, 1, 3, 1, 1) over the (C, H, W) volume of an NHWC tensor (han.py:40,46-47)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w, b = _chk(x), _chk(w), _chk(b)
        B, H, W, Cn = x.shape
        out = torch.empty_like(x)
        L.check(L.lib().sr_conv3d27(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, Cn, 0, _st()), "sr_conv3d27")
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, d):
        x, w = ctx.saved_tensors
        d = _chk(d)
        B, H, W, Cn = x.shape
        dx = torch.empty_like(x)
        L.check(L.lib().sr_conv3d27(d.data_ptr(), w.data_ptr(), None, dx.data_ptr(), B, H, W, Cn, 1, _st()), "sr_conv3d27")
        dw, db = _zeros_like(w), _zeros((1,), d.device)
        L.check(L.lib().sr_conv3d27_wgrad(x.data_ptr(), d.data_ptr(), dw.data_ptr(), db.data_ptr(), B, H, W, Cn, _st()), "sr_conv3d27_wgrad")
        return dx, dw, db


def conv3d_27(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """w: the nn.Conv3d weight [1,1,3,3,3] whose kernel axes are (channel-depth, H, W)."""
    return _Conv3d27.apply(x, w.reshape(27), b)


# ---- 2-D real FFT as matrix products on the fp32 matrix cores (SwinFIR's FourierUnit, swinfir.py:19-35: rfftn / irfftn, norm="ortho")
_DFT = _LruCache(8, 256 << 20)


def _dft_mats(H: int, W: int, device):
    key = (H, W, str(device))
    m = _DFT.get(key)
    if m is None:
        f64 = torch.float64
        Wf = W // 2 + 1
        k, w_ = torch.arange(Wf, dtype=f64)[:, None], torch.arange(W, dtype=f64)[None, :]
        ang = 2 * torch.pi * k * w_ / W
        fw = torch.stack([torch.cos(ang), -torch.sin(ang)], 1).reshape(2 * Wf, W) / W ** 0.5           # rows (k, part) -> [2Wf, W]
        wk = torch.full((Wf,), 2.0, dtype=f64)
        wk[0] = 1.0
        if W % 2 == 0:
            wk[-1] = 1.0
        finv = (torch.stack([torch.cos(ang) * wk[:, None], -torch.sin(ang) * wk[:, None]], 1).reshape(2 * Wf, W) / W ** 0.5).t().contiguous()  # [W, 2Wf]
        hh = torch.arange(H, dtype=f64)
        angh = 2 * torch.pi * hh[:, None] * hh[None, :] / H
        gr, gi = torch.cos(angh) / H ** 0.5, -torch.sin(angh) / H ** 0.5
        m = tuple(t.to(torch.float32).contiguous().to(device) for t in (fw, finv, gr, gi))
        _DFT.put(key, m)
    return m


def _dft_h(Y: Tensor, Gr: Tensor, Gi: Tensor, sgn: float, B: int, H: int, Wf: int, Cn: int) -> Tensor:
    """Z = (Gr + i sgn Gi) Y along H for Y [B, H, Wf, (re | im) x C]: four real GEMMs batched over (b, k)."""
    Z = torch.empty_like(Y)
    ld = Wf * 2 * Cn
    kw = dict(nb=(B, Wf), sab=(0, 0), sbb=(H * ld, 2 * Cn), scb=(H * ld, 2 * Cn), bf16_ok=False)  # the reference never runs its FFT in bf16
    bgemm(Gr, Y, Z, H, Cn, H, (H, 1), (ld, 1), (ld, 1), b_off=0, c_off=0, **kw)                               # Zr  = Gr Yr
    bgemm(Gi, Y, Z, H, Cn, H, (H, 1), (ld, 1), (ld, 1), b_off=Cn, c_off=0, alpha=-sgn, accumulate=True, **kw)  # Zr -= sgn Gi Yi
    bgemm(Gi, Y, Z, H, Cn, H, (H, 1), (ld, 1), (ld, 1), b_off=0, c_off=Cn, alpha=sgn, **kw)                    # Zi  = sgn Gi Yr
    bgemm(Gr, Y, Z, H, Cn, H, (H, 1), (ld, 1), (ld, 1), b_off=Cn, c_off=Cn, accumulate=True, **kw)             # Zi += Gr Yi
    return Z


class _Rfft2(Fn):
    """x [B,H,W,C] -> rfftn over (H, W), norm 'ortho', as [B, H, W/2+1, 2C] with channels (real | imag) = torch.cat((f.real, f.imag), 1)."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x)
        B, H, W, Cn = x.shape
        Wf = W // 2 + 1
        fw, finv, gr, gi = _dft_mats(H, W, x.device)
        Y = torch.empty(B, H, Wf, 2 * Cn, device=x.device, dtype=torch.float32)
        bgemm(fw, x, Y, 2 * Wf, Cn, W, (W, 1), (Cn, 1), (Cn, 1), nb=(B * H, 1), sbb=(W * Cn, 0), scb=(2 * Wf * Cn, 0), bf16_ok=False)
        ctx.shape = (B, H, W, Cn)
        return _dft_h(Y, gr, gi, 1.0, B, H, Wf, Cn)

    @staticmethod
    def backward(ctx, dZ):
        B, H, W, Cn = ctx.shape
        Wf = W // 2 + 1
        dZ = _chk(dZ)
        fw, finv, gr, gi = _dft_mats(H, W, dZ.device)
        dY = _dft_h(dZ, gr, gi, -1.0, B, H, Wf, Cn)  # the DFT matrix is symmetric: adjoint = conjugate
        dx = torch.empty(B, H, W, Cn, device=dZ.device, dtype=torch.float32)
        bgemm(fw, dY, dx, W, Cn, 2 * Wf, (1, W), (Cn, 1), (Cn, 1), nb=(B * H, 1), sbb=(2 * Wf * Cn, 0), scb=(W * Cn, 0), bf16_ok=False)
        return dx


class _Irfft2(Fn):
    """[B, H, W/2+1, (real | imag) x C] -> irfftn(s=(H, W), norm 'ortho') [B,H,W,C] (imaginary parts of the DC / Nyquist bins are ignored,
    as torch's c2r transform does)."""

    @staticmethod
    def forward(ctx, Z, W):
        Z = _chk(Z)
        B, H, Wf, C2 = Z.shape
        Cn = C2 // 2
        assert Wf == W // 2 + 1
        fw, finv, gr, gi = _dft_mats(H, W, Z.device)
        Y = _dft_h(Z, gr, gi, -1.0, B, H, Wf, Cn)
        x = torch.empty(B, H, W, Cn, device=Z.device, dtype=torch.float32)
        bgemm(finv, Y, x, W, Cn, 2 * Wf, (2 * Wf, 1), (Cn, 1), (Cn, 1), nb=(B * H, 1), sbb=(2 * Wf * Cn, 0), scb=(W * Cn, 0), bf16_ok=False)
        ctx.shape = (B, H, W, Cn)
        return x

    @staticmethod
    def backward(ctx, dx):
        B, H, W, Cn = ctx.shape
        Wf = W // 2 + 1
        dx = _chk(dx)
        fw, finv, gr, gi = _dft_mats(H, W, dx.device)
        dY = torch.empty(B, H, Wf, 2 * Cn, device=dx.device, dtype=torch.float32)
        bgemm(finv, dx, dY, 2 * Wf, Cn, W, (1, 2 * Wf), (Cn, 1), (Cn, 1), nb=(B * H, 1), sbb=(W * Cn, 0), scb=(2 * Wf * Cn, 0), bf16_ok=False)
        return _dft_h(dY, gr, gi, 1.0, B, H, Wf, Cn), None


def rfft2(x: Tensor) -> Tensor:
    return _Rfft2.apply(x)


def irfft2(z: Tensor, W: int) -> Tensor:
    return _Irfft2.apply(z, W)
