"""Fast training path (C ABI v7): the training step of the reference Trainer (studiosr/engine/trainer.py:97-109: autocast(bfloat16) forward,
L1 loss, loss.backward(), Adam) for HAT's default block geometry as fused HIP launches instead of the generic one-kernel-per-op engine
(studiosr_amd/autograd.py).  What runs per HAB (hat.py:153-195):

    forward   sr_tr_qkv_fwd -> sr_window_attention || sr_cab_fused -> sr_tr_tail_fwd                                   (4 launches)
    backward  sr_tr_tail_bwd -> sr_tr_attn_bwd (2 passes; the bias-table gradient leaves pass Q as per-workgroup table partials) -> CAB: conv1 recompute, sr_tr_ca_bwd, conv2 dgrad, sr_tr_gelu,
              conv1 dgrad (the dgrads are sr_conv3x3 with flipped / transposed packed weights) -> sr_tr_qkv_bwd -> sr_tr_wgrad (6 jobs)

Host-side design:
  * FlatParams: every parameter becomes a view of ONE fp32 buffer P, its gradient a view of ONE buffer G (torch.optim / DDP see ordinary
    tensors).
  * Arena: the kernels' packed operands (weight streams, packed convolutions, padded vectors, gathered relative-position bias) are
    produced from P by ONE sr_tr_gather launch per arena and optimizer step through index maps built once on the host; the maps are written
    as the same reshape / transpose chains as studiosr_amd/packing.py, applied to parameter INDICES instead of values.
  * The adjoint, sr_tr_finalize, turns the weight-gradient GEMMs' partial sums (and the LayerNorm / channel-attention / bias-table
    partials) into G through index maps of the same kind: one launch per stage.
Activations kept for the backward live in per-block static buffers (one forward in flight per model, checked).

Round 5: the same plan covers SwinIR's default geometry (swinir.py:105-174,258-339: 8 x 8 windows, no conv branch, no OCAB): per block
    forward   sr_tr_qkv_fwd -> sr_window_attention -> sr_tr_tail_fwd                       backward  sr_tr_tail_bwd -> sr_tr_attn_bwd (one pass) -> sr_tr_qkv_bwd -> sr_tr_wgrad (4 jobs)
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from . import ops

Tensor = torch.Tensor
CP, HP, HEADS, HD, HDP, CR = 192, 384, 6, 30, 32, 6
C_REAL, HID = 180, 360
ATTN_BWD_LDS = os.environ.get("SR_TR_ATTN_LDS", "1") != "0"  # A/B knob: window-attention backward as one LDS-form launch (read by the library too)
ATTN_LDS = os.environ.get("SR_ATTN_LDS", "1") != "0"  # A/B knob: window attention forward with K / V^T / distinct bias tiles in LDS
CAB_BWD_FUSED = os.environ.get("SR_TR_CAB_BWD_FUSED", "1") != "0"  # A/B knob: the CAB's data gradient (conv, GELU', conv) as one sr_cab_fused launch in its backward form
CONV_WG_SIDE = os.environ.get("SR_TR_CONV_WG_SIDE", "1") != "0"  # A/B knob: the CAB convs' weight-gradient launch on the backward's side stream
FINALIZE_LONG = int(os.environ.get("SR_TR_FINALIZE_LONG", "64"))  # tuning knob: items with at least this many slices are finalized by eight lanes each (0: never)
OCA_LSE = os.environ.get("SR_TR_OCA_LSE", "1") != "0"  # A/B knob: the OCAB's forward keeps its log-sum-exp; the backward's pass Q then runs tile by tile at two workgroups per CU
BWD_DUAL = os.environ.get("SR_TR_BWD_DUAL", "1") != "0"  # A/B knob: the CAB branch of a HAB's backward on a side stream beside the attention backward
_SIDE = {}


def _side_stream(device) -> "torch.cuda.Stream":
    st = _SIDE.get(device)
    if st is None:
        st = _SIDE[device] = torch.cuda.Stream(device=device)
    return st


CAB_MIDPRE = os.environ.get("SR_TR_MIDPRE", "1") != "0"  # A/B knob: the CAB's conv1 pre-activation kept by the forward (SrCab.mid_pre) instead of a conv1 launch in the backward
HAB_MID = os.environ.get("SR_HAB_MID", "1") != "0"  # A/B knob: window attention + CAB forward as one launch (sr_hab_mid)
WG_KS = int(os.environ.get("SR_WG_KS", "16"))  # token slices of the weight-gradient GEMMs (A/B knob)


def _st():
    return torch.cuda.current_stream().cuda_stream


PLAN = os.environ.get("SR_TR_PLAN", "1") != "0"  # A/B knob: the step's launch sequences recorded once and replayed from C (sr_plan_run); 0 = enqueue from Python every step


WG_SIDE = os.environ.get("SR_TR_WG_SIDE", "0") != "0"  # A/B knob: the blocks' weight-gradient launches (and each stage's sr_tr_finalize_to) on a stream of their own
_WG = {}


def _wg_stream(device) -> "torch.cuda.Stream":
    st = _WG.get(device)
    if st is None:
        st = _WG[device] = torch.cuda.Stream(device=device)
    return st


def _ev_record(stream: "torch.cuda.Stream"):
    """A point on `stream` that another stream can wait for later (_ev_wait): a plan-owned event inside a recording, a torch event otherwise."""
    rec = L.recorder()
    if rec is not None:
        e = rec.event()
        rec.add_event_record(stream.cuda_stream, e)
        return e
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev


def _ev_wait(stream: "torch.cuda.Stream", tok) -> None:
    rec = L.recorder()
    if rec is not None:
        rec.add_stream_wait(stream.cuda_stream, tok)
    else:
        stream.wait_event(tok)


def _edge(src: "torch.cuda.Stream", dst: "torch.cuda.Stream", ev: "torch.cuda.Event") -> None:
    """dst waits for everything enqueued on src so far.  Inside a launch-plan recording the edge is recorded (events owned by the plan); otherwise a torch event."""
    rec = L.recorder()
    if rec is not None:
        e = rec.event()
        rec.add_event_record(src.cuda_stream, e)
        rec.add_stream_wait(dst.cuda_stream, e)
    else:
        ev.record(src)
        dst.wait_event(ev)


# --------------------------------------------------------------------------- flat parameters
class FlatParams:
    """All parameters of a model as views of one fp32 buffer (P) and their gradients as views of another (G)."""

    def __init__(self, model: torch.nn.Module) -> None:
        self.params = [p for p in model.parameters()]
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            assert p.dtype == torch.float32
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4  # 16-byte aligned
        self.P = torch.zeros(n, dtype=torch.float32, device=dev)
        self.G = torch.zeros(n, dtype=torch.float32, device=dev)
        self._off: Dict[int, int] = {}
        self._gv_meta: Dict[int, list] = {}
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                view = self.P[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                self._off[id(p)] = o
        self.n = n

    def off(self, p: Tensor) -> int:
        return self._off[id(p)]

    def intact(self) -> bool:
        base = self.P.data_ptr()
        return all(id(p) in self._off and p.data_ptr() == base + 4 * self._off[id(p)] and p.device == self.P.device for p in self.params)

    def grad_view(self, p: Tensor, G: Optional[Tensor] = None) -> Tensor:
        o = self._off[id(p)]
        return (self.G if G is None else G)[o:o + p.numel()].view_as(p)

    def grad_views(self, params: List[Tensor], G: Tensor) -> Tuple[Tensor, ...]:
        """Fresh views of G for `params` (fresh objects every backward: autograd adopts a gradient without a copy only while nobody else holds it), one
        as_strided call per parameter from cached (shape, stride, offset) triples instead of a slice and a view_as (860 parameters: host enqueue time of a step 15.8 -> 13.5 ms)."""
        key = id(params)
        meta = self._gv_meta.get(key)
        if meta is None:
            meta = self._gv_meta[key] = [(tuple(p.shape), tuple(torch.empty(p.shape, device="meta").stride()), self._off[id(p)]) for p in params]
        return tuple(G.as_strided(sh, st, o) for sh, st, o in meta)

    def grad_target(self) -> Tensor:
        """Where this backward writes: G itself when no parameter holds a gradient yet (autograd then adopts the views: no copy, no add);
        a second buffer when gradients are being accumulated (`zero_grad(set_to_none=False)`, several backwards per step) -- .grad may then BE
        a view of G, which this backward must not overwrite before autograd adds the new gradient to it."""
        if all(p.grad is None for p in self.params):
            return self.G
        if getattr(self, "G2", None) is None:
            self.G2 = torch.zeros_like(self.G)
        return self.G2

    def pass_target(self, pass_id: int) -> Tensor:
        """grad_target() decided ONCE per backward pass, for passes made of several autograd nodes that each complete a part of the parameters (the per-stage nodes of
        SR_FAST_FULL=0 / SR_FAST_OCAB=0): the first node of pass `pass_id` to run picks the buffer that no existing .grad is a view of, the others follow it.  (Asking
        per node would send the second node to G2 because the first node's gradients were just adopted as views of G -- and the next accumulating pass would then
        overwrite G2 under the .grad tensors that alias it.)"""
        if getattr(self, "_pass", (None, None))[0] != pass_id:
            g0 = self.G.untyped_storage().data_ptr()
            on_g = any(p.grad is not None and p.grad.untyped_storage().data_ptr() == g0 for p in self.params)
            if on_g and getattr(self, "G2", None) is None:
                self.G2 = torch.zeros_like(self.G)
            self._pass = (pass_id, self.G2 if on_g else self.G)
        return self._pass[1]

    def pidx(self, p: Tensor) -> np.ndarray:
        return (self._off[id(p)] + np.arange(p.numel(), dtype=np.int64)).reshape(tuple(p.shape))


# --------------------------------------------------------------------------- index-form matrices
class IM:
    """A padded operand matrix in index form: idx (flat parameter index, -1 = constant), scl (multiplier, or the constant), mode (0 value,
    1 leading bf16, 2 remainder)."""

    def __init__(self, shape) -> None:
        self.idx = np.full(shape, -1, dtype=np.int64)
        self.scl = np.zeros(shape, dtype=np.float32)
        self.mode = np.zeros(shape, dtype=np.uint8)

    def put(self, sl, pidx: np.ndarray, scale=1.0, mode: int = 0) -> None:
        self.idx[sl] = pidx
        self.scl[sl] = scale
        self.mode[sl] = mode

    def const(self, sl, value: float) -> None:
        self.idx[sl] = -1
        self.scl[sl] = value
        self.mode[sl] = 0

    def map(self, f) -> "IM":
        out = IM.__new__(IM)
        out.idx, out.scl, out.mode = f(self.idx), f(self.scl), f(self.mode)
        return out

    @staticmethod
    def cat(ms: List["IM"]) -> "IM":
        out = IM.__new__(IM)
        out.idx = np.concatenate([m.idx.reshape(-1) for m in ms])
        out.scl = np.concatenate([m.scl.reshape(-1) for m in ms])
        out.mode = np.concatenate([m.mode.reshape(-1) for m in ms])
        return out


def _slots(m: IM) -> IM:
    """[192 rows, 32 * ns] -> weight-stream slots [ns][12 fragments = 3 w + n][64 lanes = 16 g + i][8]: element (row 48 w + 16 n + i, k = 32 c + 8 g + j)
    (the layout of packing.pack_swin_tail_stream's projection part)."""
    ns = m.idx.shape[1] // 32
    return m.map(lambda a: a.reshape(4, 3, 16, ns, 4, 8).transpose(3, 0, 1, 4, 2, 5).reshape(-1))


def _fragments(m: IM) -> IM:
    """packing.to_fragments: [N_p, K_p] -> [n_tile][k_chunk][lane][8]."""
    n_p, k_p = m.idx.shape
    return m.map(lambda a: a.reshape(n_p // 16, 16, k_p // 32, 4, 8).transpose(0, 2, 3, 1, 4).reshape(-1))


class Arena:
    """One packed operand buffer (bf16 or fp32) + its gather maps."""

    def __init__(self, dtype: torch.dtype) -> None:
        self.dtype = dtype
        self.parts: List[IM] = []
        self.n = 0
        self.buf: Optional[Tensor] = None

    def add(self, m: IM) -> int:
        off = self.n
        cnt = m.idx.size
        pad = (-cnt) % 64
        if pad:
            z = IM((pad,))
            m = IM.cat([m, z])
        self.parts.append(m)
        self.n += cnt + pad
        return off

    def finish(self, device) -> None:
        if not self.parts:
            self.parts.append(IM((64,)))
            self.n = 64
        m = IM.cat(self.parts)
        self.parts = []
        self.idx = torch.from_numpy(m.idx.astype(np.int32)).to(device)
        self.scl = torch.from_numpy(m.scl).to(device)
        self.mode = torch.from_numpy(m.mode).to(device)
        self.buf = torch.zeros(self.n, dtype=self.dtype, device=device)

    def view(self, off: int, n: int) -> Tensor:
        return self.buf[off:off + n]

    def gather(self, P: Tensor) -> None:
        L.check(L.lib().sr_tr_gather(P.data_ptr(), self.idx.data_ptr(), None, self.scl.data_ptr(), self.mode.data_ptr(), self.buf.data_ptr(),
                                     L.SR_BF16 if self.dtype == torch.bfloat16 else L.SR_F32, self.n, _st()), "sr_tr_gather")


class FinalMap:
    """Gradient side: for the parameters [p0, p1) of the flat buffer, grad[p] = scale[p] * sum_s part[src[p] + s * stride[p]]."""

    def __init__(self, fp: FlatParams, p0: int, p1: int) -> None:
        self.p0, self.p1 = p0, p1
        n = p1 - p0
        self.src = np.full(n, -1, dtype=np.int64)
        self.stride = np.zeros(n, dtype=np.int32)
        self.ns = np.ones(n, dtype=np.int32)
        self.scale = np.ones(n, dtype=np.float32)
        self.size = 0  # floats of the partial arena

    def alloc(self, n: int) -> int:
        off = self.size
        self.size += (n + 63) // 64 * 64
        return off

    def put(self, pidx: np.ndarray, src: np.ndarray, stride: int, ns: int, scale=1.0) -> None:
        i = (pidx - self.p0).reshape(-1)
        self.src[i] = np.asarray(src, dtype=np.int64).reshape(-1)
        self.stride[i] = stride
        self.ns[i] = ns
        self.scale[i] = np.asarray(scale, dtype=np.float32).reshape(-1) if np.ndim(scale) else scale

    def finish(self, device) -> None:
        # Work items in ARENA order (sr_tr_finalize_to, ABI v9): sorted by src, so that adjacent lanes read adjacent partials also for 3x3 conv weights (their nine
        # taps are adjacent parameters but a whole [Np][Kp] plane apart in the packed gradient).  src -1 (no source: the gradient is zero) becomes an item
        # with no slices; src -2 (the element belongs to another map) is no item at all.
        idx = np.nonzero(self.src != -2)[0]
        idx = idx[np.argsort(self.src[idx], kind="stable")]
        src, ns = self.src[idx].copy(), self.ns[idx].copy()
        ns[src < 0] = 0
        src[src < 0] = 0
        # Items with many slices (LayerNorm / bias-table partials: one per workgroup of the producing launch) go to a launch of their own where eight lanes share an
        # item (SrTrFinalize.lanes, ABI v11): one thread per item would walk up to 256 dependent strided loads long after the rest of the launch has finished.
        stride, scale = self.stride[idx].copy(), self.scale[idx].copy()
        long_ = ns >= FINALIZE_LONG if FINALIZE_LONG > 0 else np.zeros(ns.shape, dtype=bool)
        self.sets = []
        for sel, lanes in ((~long_, 1), (long_, 8)):
            if not sel.any():
                continue
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a[sel])).to(device)  # noqa: E731
            self.sets.append((int(sel.sum()), lanes, t(src), t(idx.astype(np.int32)), t(stride), t(ns), t(scale)))
        self.n_items = int(idx.size)
        self.part = torch.zeros(max(self.size, 64), dtype=torch.float32, device=device)

    def run(self, G: Tensor) -> None:
        for n, lanes, src, dst, stride, ns, scale in self.sets:
            _call(L.lib().sr_tr_finalize_to_args, L.SrTrFinalize, "sr_tr_finalize_to", arena=self.part.data_ptr(), src=src.data_ptr(), dst=dst.data_ptr(),
                  stride=stride.data_ptr(), ns=ns.data_ptr(), scale=scale.data_ptr(), grad=G.data_ptr() + 4 * self.p0, n=n, lanes=lanes)


# --------------------------------------------------------------------------- index-form packers (cf. studiosr_amd/packing.py)
def pack_qkv_fwd(fp: FlatParams, qkv_w: Tensor, qkv_b: Tensor) -> IM:
    """packing.pack_swin_qkv_stream in index form (unfolded weights)."""
    m = IM((3, HEADS, HDP, CP))
    sc = np.ones((3, 1, 1, 1), dtype=np.float32)
    sc[0] = HD ** -0.5
    m.put((slice(None), slice(None), slice(0, HD), slice(0, C_REAL)), fp.pidx(qkv_w).reshape(3, HEADS, HD, C_REAL), np.broadcast_to(sc, (3, HEADS, HD, C_REAL)))
    b = fp.pidx(qkv_b).reshape(3, HEADS, HD)
    bs = np.broadcast_to(sc[:, :, :, 0], (3, HEADS, HD))
    m.put((slice(None), slice(None), slice(0, HD), C_REAL), b, bs, 1)
    m.put((slice(None), slice(None), slice(0, HD), C_REAL + 1), b, bs, 2)
    return m.map(lambda a: a.reshape(3, 3, 2, 2, 16, 6, 4, 8).transpose(1, 5, 2, 3, 0, 6, 4, 7).reshape(-1))


def _m_proj(fp: FlatParams, proj_w: Tensor) -> IM:
    m = IM((CP, HEADS, HDP))
    m.put((slice(0, C_REAL), slice(None), slice(0, HD)), fp.pidx(proj_w).reshape(C_REAL, HEADS, HD))
    return m.map(lambda a: a.reshape(CP, HEADS * HDP))


def _m_fc1(fp: FlatParams, w: Tensor, b: Tensor) -> IM:
    m = IM((HP, CP))
    m.put((slice(0, HID), slice(0, C_REAL)), fp.pidx(w))
    m.put((slice(0, HID), C_REAL), fp.pidx(b), 1.0, 1)
    m.put((slice(0, HID), C_REAL + 1), fp.pidx(b), 1.0, 2)
    return m


def _m_fc2(fp: FlatParams, w: Tensor, b: Tensor) -> IM:
    m = IM((CP, HP))
    m.put((slice(0, C_REAL), slice(0, HID)), fp.pidx(w))
    m.put((slice(0, C_REAL), HID), fp.pidx(b), 1.0, 1)
    m.put((slice(0, C_REAL), HID + 1), fp.pidx(b), 1.0, 2)
    return m


def _sub(m: IM, rows: slice, cols: slice) -> IM:
    return m.map(lambda a: a[rows, cols])


def _tr(m: IM) -> IM:
    return m.map(lambda a: np.ascontiguousarray(a.T))


def pack_tail_fwd(fp: FlatParams, proj_w, fc1_w, fc1_b, fc2_w, fc2_b) -> IM:
    """sr_tr_tail_fwd's 30 slots: 6 projection, then per hidden half 6 fc1 + 6 fc2 (biases on the constant-one channels / hidden columns)."""
    fc1, fc2 = _m_fc1(fp, fc1_w, fc1_b), _m_fc2(fp, fc2_w, fc2_b)
    parts = [_slots(_m_proj(fp, proj_w))]
    for hf in range(2):
        parts.append(_slots(_sub(fc1, slice(192 * hf, 192 * hf + 192), slice(None))))
        parts.append(_slots(_sub(fc2, slice(None), slice(192 * hf, 192 * hf + 192))))
    return IM.cat(parts)


def _weights_only(m: IM, rows: int, cols: int) -> IM:
    """the matrix with everything outside [0, rows) x [0, cols) (bias columns, pads) zero"""
    out = IM(m.idx.shape)
    out.idx[:rows, :cols] = m.idx[:rows, :cols]
    out.scl[:rows, :cols] = m.scl[:rows, :cols]
    out.mode[:rows, :cols] = m.mode[:rows, :cols]
    return out


def pack_tail_bwd(fp: FlatParams, proj_w, fc1_w, fc1_b, fc2_w, fc2_b) -> IM:
    """sr_tr_tail_bwd's 42 slots: per hidden half 6 fc1 (forward, with bias), 6 W2^T (rows = hidden columns, K = channels), 6 W1^T (rows =
    channels, K = hidden columns); then 6 Wproj^T (rows = (head, feature), K = channels).  The transposed matrices carry weights only."""
    fc1, fc2 = _m_fc1(fp, fc1_w, fc1_b), _m_fc2(fp, fc2_w, fc2_b)
    w1, w2 = _weights_only(fc1, HID, C_REAL), _weights_only(fc2, C_REAL, HID)
    parts = []
    for hf in range(2):
        hs = slice(192 * hf, 192 * hf + 192)
        parts.append(_slots(_sub(fc1, hs, slice(None))))
        parts.append(_slots(_tr(_sub(w2, slice(None), hs))))  # [hidden col][channel]
        parts.append(_slots(_tr(_sub(w1, hs, slice(None)))))  # [channel][hidden col]
    parts.append(_slots(_tr(_m_proj(fp, proj_w))))            # [(head, feature)][channel]
    return IM.cat(parts)


def pack_qkv_bwd(fp: FlatParams, qkv_w: Tensor) -> IM:
    """sr_tr_qkv_bwd's 18 slots: W^T [channel][(part, head, feature)], q part scaled by hd^-0.5."""
    m = IM((3, HEADS, HDP, CP))
    sc = np.ones((3, 1, 1, 1), dtype=np.float32)
    sc[0] = HD ** -0.5
    m.put((slice(None), slice(None), slice(0, HD), slice(0, C_REAL)), fp.pidx(qkv_w).reshape(3, HEADS, HD, C_REAL), np.broadcast_to(sc, (3, HEADS, HD, C_REAL)))
    return _slots(_tr(m.map(lambda a: a.reshape(3 * HEADS * HDP, CP))))


def pack_conv(fp: FlatParams, w: Tensor, cin_p: int, cout_p: int, transpose: bool = False, rows: Optional[np.ndarray] = None) -> IM:
    """packing.pack_conv3x3 in index form: [cout_p, 9 * cin_p] with k = tap * cin_p + c; packed row n takes output channel rows[n] (-1 = zero
    row; None = identity -- a conv feeding nn.PixelShuffle uses packing.pixel_shuffle_rows).  transpose = True: the data-gradient convolution
    (its input = the forward's packed output rows, taps flipped): W'[ci][tap'][n] = W[rows[n]][ci][8 - tap']; then cout_p / cin_p are the
    DGRAD's output / input widths."""
    cout, cin = w.shape[:2]
    pi = fp.pidx(w).reshape(cout, cin, 9)
    n_rows = cin_p if transpose else cout_p  # packed forward output rows
    if rows is None:
        rows = np.full(n_rows, -1, dtype=np.int64)
        rows[:cout] = np.arange(cout)
    rows = np.asarray(rows, dtype=np.int64)
    assert rows.shape[0] == n_rows
    ok = rows >= 0
    src = pi[np.clip(rows, 0, None)]  # [n_rows, cin, 9]
    if transpose:
        m = IM((cout_p, 9, cin_p))  # rows = forward ci, columns = forward packed rows
        sub = IM((cin, 9, n_rows))
        sub.put(slice(None), src[:, :, ::-1].transpose(1, 2, 0))
        sub.idx[:, :, ~ok] = -1
        sub.scl[:, :, ~ok] = 0.0
        m.idx[:cin], m.scl[:cin], m.mode[:cin] = sub.idx, sub.scl, sub.mode
    else:
        m = IM((cout_p, 9, cin_p))
        sub = IM((n_rows, 9, cin))
        sub.put(slice(None), src.transpose(0, 2, 1))
        sub.idx[~ok] = -1
        sub.scl[~ok] = 0.0
        m.idx[:, :, :cin], m.scl[:, :, :cin], m.mode[:, :, :cin] = sub.idx, sub.scl, sub.mode
    return _fragments(m.map(lambda a: a.reshape(a.shape[0], -1)))


def pack_vec(fp: FlatParams, v: Tensor, n_p: int, rows: Optional[np.ndarray] = None) -> IM:
    m = IM((n_p,))
    if rows is None:
        m.put(slice(0, v.numel()), fp.pidx(v).reshape(-1))
    else:
        rows = np.asarray(rows, dtype=np.int64)
        ok = rows >= 0
        m.idx[ok] = fp.pidx(v).reshape(-1)[rows[ok]]
        m.scl[ok] = 1.0
    return m


def pack_bias(fp: FlatParams, table: Tensor, rpi: np.ndarray, nq: int, nk: int) -> Tuple[IM, IM, IM, Optional[IM]]:
    """relative-position bias gathered through rpi (negative indices wrap): [heads][nq][nk], its transpose [heads][nk][nq], the
    accumulator-fragment order of packing.bias_fragments, and (16 x 16 windows with a relative-position index: tile (qt, kt) depends on qt - kt
    only) the 31 distinct tiles of packing.bias_distinct_tiles (SrWindowAttn.bias_tiles: the LDS form of the window attention)."""
    T = table.shape[0]
    r = np.asarray(rpi, dtype=np.int64).reshape(nq, nk)
    r = np.where(r < 0, r + T, r)
    pi = fp.pidx(table)  # [T, heads]
    b = IM((HEADS, nq, nk))
    b.put(slice(None), pi[r].transpose(2, 0, 1))
    bt = b.map(lambda a: np.ascontiguousarray(a.transpose(0, 2, 1)))
    bf = b.map(lambda a: a.reshape(HEADS, nq // 16, 16, nk // 16, 4, 4).transpose(0, 1, 3, 4, 2, 5).reshape(-1))
    b31 = None
    if nq == 256 and nk == 256:
        r5 = r.reshape(16, 16, 16, 16).transpose(0, 2, 1, 3)  # [qt, kt, i, j]
        first = np.stack([r5[max(d - 15, 0), max(15 - d, 0)] for d in range(31)])
        qt = np.arange(16)
        if np.array_equal(first[qt[:, None] - qt[None, :] + 15], r5):
            b31 = b.map(lambda a: np.stack([a.reshape(HEADS, 16, 16, 16, 16)[:, max(d - 15, 0), :, max(15 - d, 0), :] for d in range(31)], 1)
                        .reshape(HEADS, 31, 16, 4, 4).transpose(0, 1, 3, 2, 4).reshape(-1))
    return b, bt, bf, b31


def pack_bias_rel(b: IM, rpi: np.ndarray, T: int) -> Optional[IM]:
    """The OCAB's gathered bias [heads][256][576] as its rotated relative-position table [heads][1521] (SrTrAttnFwd.bias_rel, packing.oca_bias_rel),
    or None when rpi is not a function of the (row, column) differences."""
    from . import packing

    J = packing.oca_rel_index().numpy()  # [256, 576] -> position in the table
    qrep, krep = np.zeros(1521, dtype=np.int64), np.zeros(1521, dtype=np.int64)
    qq, kk = np.meshgrid(np.arange(256), np.arange(576), indexing="ij")
    qrep[J.reshape(-1)], krep[J.reshape(-1)] = qq.reshape(-1), kk.reshape(-1)
    r = np.asarray(rpi, dtype=np.int64).reshape(256, 576)
    r = np.where(r < 0, r + T, r)
    if not np.array_equal(r[qrep, krep][J], r):
        return None
    return b.map(lambda a: np.ascontiguousarray(a[:, qrep, krep]).reshape(-1))


# --------------------------------------------------------------------------- one block (HAB or OCAB)
class BlockPlan:
    """Offsets, maps and static buffers of one HAB (hat.py:95-104,153-195), one OCAB (hat.py:107-118,239-293; oca = True: no CAB, no shift,
    no DropPath, keys / values from the unfolded 24 x 24 neighbourhood) or one SwinTransformerBlock of SwinIR (swinir.py:105-174; ws = 8: 64-token
    windows, no conv branch -- the same four block kernels, which walk 64 window-order tokens per workgroup whatever the window)."""

    def __init__(self, fp: FlatParams, blk, rpi: np.ndarray, rpi_dev: Tensor, conv_scale: float, shift: int, wa: Arena, fa: Arena, fm: "FinalMap", oca: bool = False,
                 ws: int = 16) -> None:
        self.shift, self.conv_scale, self.oca, self.ws = shift, float(conv_scale), oca, ws
        self._ev = None  # (fork, join) events of the backward's side stream
        self.fp, self.blk, self.fm, self.rpi_dev = fp, blk, fm, rpi_dev
        at = blk if oca else blk.attn  # OCAB holds qkv / proj / table itself
        mlp = blk.mlp
        self.at, self.mlp = at, mlp
        self.nq = ws * ws
        self.nk = 576 if oca else self.nq
        self.o_qkvf = wa.add(pack_qkv_fwd(fp, at.qkv.weight, at.qkv.bias))
        self.o_tailf = wa.add(pack_tail_fwd(fp, at.proj.weight, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias))
        self.o_tailb = wa.add(pack_tail_bwd(fp, at.proj.weight, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias))
        self.o_qkvb = wa.add(pack_qkv_bwd(fp, at.qkv.weight))
        self.o_g1, self.o_b1 = fa.add(pack_vec(fp, blk.norm1.weight, CP)), fa.add(pack_vec(fp, blk.norm1.bias, CP))
        self.o_g2, self.o_b2 = fa.add(pack_vec(fp, blk.norm2.weight, CP)), fa.add(pack_vec(fp, blk.norm2.bias, CP))
        self.o_bp = fa.add(pack_vec(fp, at.proj.bias, CP))
        self.table = at.relative_position_bias_table
        b, bt, bf, b31 = pack_bias(fp, self.table, rpi, self.nq, self.nk)
        cy, cx = np.divmod(np.arange(self.nq), ws)  # is rpi the standard relative-position index of a ws x ws window (hat.py:480-492, swinir.py:56-67)?
        std = (cy[:, None] - cy[None, :] + ws - 1) * (2 * ws - 1) + (cx[:, None] - cx[None, :] + ws - 1)
        self.std_rpi = (not oca) and (ws == 16 or np.array_equal(np.asarray(rpi, dtype=np.int64).reshape(self.nq, self.nq), std))
        self.o_bias, self.o_biasT = fa.add(b), fa.add(bt)
        self.o_biasF = None if oca else fa.add(bf)
        self.o_bias31 = None if (oca or b31 is None or not ATTN_LDS) else fa.add(b31)
        rel = pack_bias_rel(b, rpi, self.table.shape[0]) if oca else None
        self.o_bias_rel = None if rel is None else fa.add(rel)
        self.cab = None
        if not oca and hasattr(blk, "conv_block"):
            cab = blk.conv_block.cab
            ca = cab[3].attention
            self.cab, self.ca_mod = cab, ca
            self.o_c1 = wa.add(pack_conv(fp, cab[0].weight, CP, 64))
            self.o_c2 = wa.add(pack_conv(fp, cab[2].weight, 64, CP))
            self.o_c1t = wa.add(pack_conv(fp, cab[0].weight, 64, CP, transpose=True))   # dmid [64] -> dn1 [192]
            self.o_c2t = wa.add(pack_conv(fp, cab[2].weight, CP, 64, transpose=True))   # dy [192] -> dmid [64]
            self.o_bc1, self.o_bc2 = fa.add(pack_vec(fp, cab[0].bias, 64)), fa.add(pack_vec(fp, cab[2].bias, CP))
            self.ca = (ca[1].weight, ca[1].bias, ca[3].weight, ca[3].bias)
        self._final_maps()

    def _final_maps(self) -> None:
        fp, fm, at, mlp, ks = self.fp, self.fm, self.at, self.mlp, WG_KS
        # -- qkv: out [ks][576][192]; n = part * 192 + head * 32 + d
        n_qkv = 3 * HEADS * HDP
        self.f_qkv = fm.alloc(ks * n_qkv * CP)
        nn = (np.arange(3)[:, None, None] * 192 + np.arange(HEADS)[None, :, None] * 32 + np.arange(HD)[None, None, :]).reshape(-1)  # [540] -> padded row
        sc = np.ones(3 * HEADS * HD, dtype=np.float32)
        sc[: HEADS * HD] = HD ** -0.5
        fm.put(fp.pidx(at.qkv.weight), self.f_qkv + nn[:, None] * CP + np.arange(C_REAL)[None, :], n_qkv * CP, ks, np.broadcast_to(sc[:, None], (540, C_REAL)))
        fm.put(fp.pidx(at.qkv.bias), self.f_qkv + nn * CP + C_REAL, n_qkv * CP, ks, sc)
        # -- proj: out [ks][192 c][192 (head, d)]; bias = column 30 (a pad feature of O, read as one)
        self.f_proj = fm.alloc(ks * CP * CP)
        kk = (np.arange(HEADS)[:, None] * 32 + np.arange(HD)[None, :]).reshape(-1)
        fm.put(fp.pidx(at.proj.weight), self.f_proj + np.arange(C_REAL)[:, None] * CP + kk[None, :], CP * CP, ks)
        fm.put(fp.pidx(at.proj.bias), self.f_proj + np.arange(C_REAL) * CP + HD, CP * CP, ks)
        # -- fc1: out [ks][384][192]; fc2: out [ks][192][384]
        self.f_fc1 = fm.alloc(ks * HP * CP)
        fm.put(fp.pidx(mlp.fc1.weight), self.f_fc1 + np.arange(HID)[:, None] * CP + np.arange(C_REAL)[None, :], HP * CP, ks)
        fm.put(fp.pidx(mlp.fc1.bias), self.f_fc1 + np.arange(HID) * CP + C_REAL, HP * CP, ks)
        self.f_fc2 = fm.alloc(ks * CP * HP)
        fm.put(fp.pidx(mlp.fc2.weight), self.f_fc2 + np.arange(C_REAL)[:, None] * HP + np.arange(HID)[None, :], CP * HP, ks)
        fm.put(fp.pidx(mlp.fc2.bias), self.f_fc2 + np.arange(C_REAL) * HP + HID, CP * HP, ks)

    def prepare(self, B: int, H: int, W: int, dev, groups: int) -> None:
        """Geometry-dependent parts: gradient partial buffers that depend on the number of workgroups / images; static activations."""
        fp, fm = self.fp, self.fm
        nbw_ = B * H * W // self.nq
        # HAB with the LDS form of the attention backward (csrc/sr_tr_attn_lds.hip): one bias-table partial per (head, window), i.e. groups * 4 == windows;
        # OCAB (and HABs without it): `groups` window groups whose pass-Q workgroups walk their windows with the gradient tiles in registers;
        # 8 x 8 windows: groups * 4 == windows selects the one-pass kernel (one wave per (window, head), one table partial per four windows of a head);
        # a window count that is not a multiple of four takes the two register passes with one window per group
        if self.ws == 8:
            self.groups = nbw_ // 4 if (nbw_ % 4 == 0 and self.std_rpi) else nbw_
        else:
            self.groups = nbw_ // 4 if (not self.oca and nbw_ % 4 == 0 and ATTN_BWD_LDS) else groups
        nwg = B * H * W // 64
        self.f_ln1, self.f_ln2 = fm.alloc(nwg * 2 * CP), fm.alloc(nwg * 2 * CP)
        for f, norm in ((self.f_ln1, self.blk.norm1), (self.f_ln2, self.blk.norm2)):
            fm.put(fp.pidx(norm.weight), f + np.arange(C_REAL), 2 * CP, nwg)
            fm.put(fp.pidx(norm.bias), f + CP + np.arange(C_REAL), 2 * CP, nwg)
        if self.cab is not None:
            cab, ks = self.cab, conv_ks(B, H, W)  # CAB convs: out [ks][9][Np][Kp]
            self.ks_conv = ks
            c3 = cab[0].weight.shape[0]
            self.f_c1 = fm.alloc(ks * 9 * 64 * CP)
            co, ci, tp = np.arange(c3)[:, None, None], np.arange(C_REAL)[None, :, None], np.arange(9)[None, None, :]
            fm.put(fp.pidx(cab[0].weight).reshape(c3, C_REAL, 9), self.f_c1 + (tp * 64 + co) * CP + ci, 9 * 64 * CP, ks)
            fm.put(fp.pidx(cab[0].bias), self.f_c1 + (4 * 64 + np.arange(c3)) * CP + C_REAL, 9 * 64 * CP, ks)  # centre tap, the ones channel of n1
            self.f_c2 = fm.alloc(ks * 9 * CP * 64)
            co, ci = np.arange(C_REAL)[:, None, None], np.arange(c3)[None, :, None]
            fm.put(fp.pidx(cab[2].weight).reshape(C_REAL, c3, 9), self.f_c2 + (tp * CP + co) * 64 + ci, 9 * CP * 64, ks)
            fm.put(fp.pidx(cab[2].bias), self.f_c2 + (4 * CP + np.arange(C_REAL)) * 64 + c3, 9 * CP * 64, ks)    # ones_col = c3
            ca = self.ca_mod
            cr = ca[1].weight.shape[0]
            self.ca_stride = (2 * cr * C_REAL + cr + C_REAL + 63) // 64 * 64
            self.f_ca = fm.alloc(B * self.ca_stride)
            o = self.f_ca
            fm.put(fp.pidx(ca[1].weight), o + np.arange(cr * C_REAL), self.ca_stride, B)
            fm.put(fp.pidx(ca[1].bias), o + cr * C_REAL + np.arange(cr), self.ca_stride, B)
            fm.put(fp.pidx(ca[3].weight), o + cr * C_REAL + cr + np.arange(C_REAL * cr), self.ca_stride, B)
            fm.put(fp.pidx(ca[3].bias), o + cr * C_REAL + cr + C_REAL * cr + np.arange(C_REAL), self.ca_stride, B)
        tb = self.table  # [T, heads]: one table-sized partial per pass-Q workgroup, a head's groups * 4 workgroups consecutive (sr_tr_attn_bwd)
        self.tpad = (tb.shape[0] + 63) // 64 * 64
        nq_wg = self.groups * (self.nq // 64)
        self.f_tab = fm.alloc(HEADS * nq_wg * self.tpad)
        fm.put(fp.pidx(tb), self.f_tab + np.arange(HEADS)[None, :] * nq_wg * self.tpad + np.arange(tb.shape[0])[:, None], self.tpad, nq_wg)
        T = B * H * W
        bf, f32 = torch.bfloat16, torch.float32
        e = lambda *s, dt=bf: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        self.x1 = e(T, CP, dt=f32)
        self.q, self.qT, self.o = e(T * CP), e(T * CP), e(T, CP)
        if self.oca:
            n = T // 256 * HEADS * 576 * 32
            self.k, self.kT, self.v = e(n), e(n), e(n)  # the unfolded neighbourhoods
            self.lse_fwd = e(T * HEADS, dt=f32) if (self.o_bias_rel is not None and OCA_LSE) else None
        else:
            self.k, self.kT, self.v, self.vT = (e(T * CP) for _ in range(4))
            self.n1 = None
            if self.cab is not None:
                self.n1, self.y = e(T, CP), e(T, CP)
                self.mid_pre = e(T, 64) if CAB_MIDPRE else None  # conv1's pre-activation, kept by the forward (2 MB per block at 4 x 64 x 64) instead of being recomputed
                self.n_tiles = ops.cab_pool_tiles(H, W)
                self.pool = e(B, self.n_tiles, CP, dt=f32)
                self.gate = e(B, CP, dt=f32)

    # ------------------------------------------------------------------ forward
    def forward(self, st: "Stage", cur: Tensor, out: Tensor, sc_i: Optional[Tensor]) -> None:
        B, H, W = st.geo
        wa, fa, sc = st.wa.buf, st.fa.buf, st.sc
        lib = L.lib()
        g = dict(B=B, H=H, W=W, C=C_REAL, Cp=CP, ldx=CP, heads=HEADS, hd_p=HDP, ws=self.ws, eps=1e-5)
        nbw = B * H * W // self.nq
        if self.oca:
            _call(lib.sr_tr_qkv_fwd, L.SrTrQkvFwd, "sr_tr_qkv_fwd", x=cur.data_ptr(), gamma=fa[self.o_g1:].data_ptr(), beta=fa[self.o_b1:].data_ptr(),
                  wstream=wa[self.o_qkvf:].data_ptr(), q=self.q.data_ptr(), qT=self.qT.data_ptr(), k=sc.dk.data_ptr(), kT=sc.dq.data_ptr(), v=sc.dv.data_ptr(),
                  vT=sc.dOT.data_ptr(), n1=None, ldn=CP, shift=0, **g)  # k / v in scratch (dk, dv; the transposes are not needed): only their unfolded copies are kept
            a = L.SrTrOcaFold()
            a.k, a.v, a.kwin, a.vwin, a.kwinT, a.vwinT = sc.dk.data_ptr(), sc.dv.data_ptr(), self.k.data_ptr(), self.v.data_ptr(), self.kT.data_ptr(), sc.vwinT.data_ptr()
            a.B, a.nwy, a.nwx, a.heads, a.wse, a.pad = B, H // 16, W // 16, HEADS, 24, 4
            L.check(lib.sr_tr_oca_fold(C.byref(a), 1, _st()), "sr_tr_oca_fold")
            _call(lib.sr_tr_attn_fwd, L.SrTrAttnFwd, "sr_tr_attn_fwd", q=self.q.data_ptr(), k=self.k.data_ptr(), vT=sc.vwinT.data_ptr(), bias=fa[self.o_bias:].data_ptr(),
                  out=self.o.data_ptr(), n_bwin=nbw, heads=HEADS, hd_p=HDP, Nq=256, Nk=576, ldo=CP,
                  bias_rel=None if self.o_bias_rel is None else fa[self.o_bias_rel:].data_ptr(),
                  lse=None if self.lse_fwd is None else self.lse_fwd.data_ptr())  # (the LDS form keeps every query row's log-sum-exp for the backward's pass Q)
        else:
            _call(lib.sr_tr_qkv_fwd, L.SrTrQkvFwd, "sr_tr_qkv_fwd", x=cur.data_ptr(), gamma=fa[self.o_g1:].data_ptr(), beta=fa[self.o_b1:].data_ptr(),
                  wstream=wa[self.o_qkvf:].data_ptr(), q=self.q.data_ptr(), qT=self.qT.data_ptr(), k=self.k.data_ptr(), kT=self.kT.data_ptr(), v=self.v.data_ptr(),
                  vT=self.vT.data_ptr(), n1=None if self.n1 is None else self.n1.data_ptr(), ldn=CP, shift=self.shift, **g)
            akw = dict(q=self.q.data_ptr(), k=self.k.data_ptr(), vt=self.vT.data_ptr(), bias=fa[self.o_bias:].data_ptr(), out=self.o.data_ptr(), n_bwin=nbw,
                       heads=HEADS, hd_p=HDP, ntok=self.nq, H=H, W=W, ws=self.ws, shift=self.shift, dtype=L.SR_BF16, y_mode=L.Y_ROLL,
                       bias_frag=fa[self.o_biasF:].data_ptr(), qkv_frag=0, bias_tiles=None if self.o_bias31 is None else fa[self.o_bias31:].data_ptr())
        if not self.oca and self.cab is None:  # SwinIR: attention only (swinir.py:146-163)
            ops.window_attention(**akw)
        elif not self.oca:
            ckw = dict(x=self.n1.data_ptr(), w1p=wa[self.o_c1:].data_ptr(), b1=fa[self.o_bc1:].data_ptr(), w2p=wa[self.o_c2:].data_ptr(), b2=fa[self.o_bc2:].data_ptr(),
                       y=self.y.data_ptr(), pool_partial=self.pool.data_ptr(), B=B, H=H, W=W, Cin_p=CP, Cmid_p=64, Cout_p=CP, dtype=L.SR_BF16,
                       mid_pre=None if self.mid_pre is None else self.mid_pre.data_ptr())
            if HAB_MID:  # the two independent launches as one (sr_hab_mid, ABI v8)
                ops.hab_mid(akw, ckw)
            else:
                ops.window_attention(**akw)
                ops.cab_fused(**ckw)
        kw = {}
        if self.cab is not None:
            w1, b1, w2, b2 = self.ca
            kw = dict(y=self.y.data_ptr(), pool_partial=self.pool.data_ptr(), ca_w1=w1.data_ptr(), ca_b1=b1.data_ptr(), ca_w2=w2.data_ptr(), ca_b2=b2.data_ptr(),
                      gate_out=self.gate.data_ptr(), ca_Cr=w1.shape[0], ca_n_tiles=self.n_tiles, y_scale=self.conv_scale)
        _call(lib.sr_tr_tail_fwd, L.SrTrTailFwd, "sr_tr_tail_fwd", x=cur.data_ptr(), out=out.data_ptr(), x1=self.x1.data_ptr(), o=self.o.data_ptr(),
              wstream=wa[self.o_tailf:].data_ptr(), bproj=fa[self.o_bp:].data_ptr(), gamma=fa[self.o_g2:].data_ptr(), beta=fa[self.o_b2:].data_ptr(),
              s_a=None if sc_i is None else sc_i[0].data_ptr(), s_m=None if sc_i is None else sc_i[1].data_ptr(), ldy=CP, shift=self.shift, Hp=HP, **kw, **g)

    # ------------------------------------------------------------------ backward
    def backward(self, st: "Stage", xin: Tensor, d: Tensor, dx: Tensor, sc_i: Optional[Tensor]) -> None:
        B, H, W = st.geo
        T = B * H * W
        wa, fa, sc, fm = st.wa.buf, st.fa.buf, st.sc, st.fm
        lib = L.lib()
        g = dict(B=B, H=H, W=W, C=C_REAL, Cp=CP, ldx=CP, heads=HEADS, hd_p=HDP, ws=self.ws, eps=1e-5)
        pp = lambda off: fm.part.data_ptr() + 4 * off  # noqa: E731
        s_a = None if sc_i is None else sc_i[0].data_ptr()
        s_m = None if sc_i is None else sc_i[1].data_ptr()
        # operand set of this block's weight-gradient launch (see Scratch): wait until the launch that read it two blocks ago is through
        main = torch.cuda.current_stream()
        k = sc.turn & 1
        sc.turn += 1
        op = sc.ops[k]
        if WG_SIDE and sc.wg_busy[k] is not None:
            _ev_wait(main, sc.wg_busy[k])
            sc.wg_busy[k] = None
        kw = {}
        if self.cab is not None:
            kw = dict(y=self.y.data_ptr(), gate=self.gate.data_ptr(), dyc=op.dyc.data_ptr(), dgate_part=sc.dgate_part.data_ptr())
        _call(lib.sr_tr_tail_bwd, L.SrTrTailBwd, "sr_tr_tail_bwd", dout=d.data_ptr(), x1=self.x1.data_ptr(), gamma=fa[self.o_g2:].data_ptr(), beta=fa[self.o_b2:].data_ptr(),
              wstream=wa[self.o_tailb:].data_ptr(), s_a=s_a, s_m=s_m, dx1=sc.dx1.data_ptr(), n2w=op.n2w.data_ptr(), doutw=op.doutw.data_ptr(), gw=op.gw.data_ptr(),
              dhw=op.dhw.data_ptr(), dOw=sc.dOw.data_ptr(), dOT=sc.dOT.data_ptr(), dx1sw=op.dx1sw.data_ptr(), ln_part=pp(self.f_ln2), ldy=CP, shift=self.shift, Hp=HP, **kw, **g)
        # HAB: the CAB branch of the backward (channel attention, the two data-gradient convs, GELU') only shares sr_tr_tail_bwd's outputs with the attention
        # backward and joins it in sr_tr_qkv_bwd: it runs on a side stream beside sr_tr_attn_bwd (two event edges per block; SR_TR_BWD_DUAL=0: one stream)
        dual = BWD_DUAL and self.cab is not None
        if dual:
            side = _side_stream(main.device)
            if self._ev is None:
                self._ev = tuple(torch.cuda.Event() for _ in range(4))
            _edge(main, side, self._ev[0])
            with torch.cuda.stream(side):
                jobs_cab = self._cab_backward(st, B, H, W, T, lib, wa, fa, sc, pp, op)
        dkp, dvp = (sc.dkwin, sc.dvwin) if self.oca else (sc.dk, sc.dv)
        lse_given = self.oca and getattr(self, "lse_fwd", None) is not None
        _call(lib.sr_tr_attn_bwd, L.SrTrAttnBwd, "sr_tr_attn_bwd", q=self.q.data_ptr(), qT=self.qT.data_ptr(), k=self.k.data_ptr(), kT=self.kT.data_ptr(), v=self.v.data_ptr(),
              o=self.o.data_ptr(), dO=sc.dOw.data_ptr(), dOT=sc.dOT.data_ptr(), bias=fa[self.o_bias:].data_ptr(), biasT=fa[self.o_biasT:].data_ptr(), dq=sc.dq.data_ptr(),
              dk=dkp.data_ptr(), dv=dvp.data_ptr(), lse=(self.lse_fwd if lse_given else sc.lse).data_ptr(), lse_given=int(lse_given), delta=sc.delta.data_ptr(), dtab_part=pp(self.f_tab), rpi=self.rpi_dev.data_ptr(), n_bwin=T // self.nq,
              heads=HEADS, hd_p=HDP, Nq=self.nq, Nk=self.nk, ldo=CP, groups=self.groups, T=self.table.shape[0], Tpad=self.tpad, toeplitz16=int(not self.oca and self.std_rpi), H=H, W=W,
              ws=self.ws, shift=self.shift,
              oca_rel=int(self.oca and self.o_bias_rel is not None))
        jobs = []
        ks = WG_KS
        if self.oca:
            a = L.SrTrOcaFold()
            a.k, a.v, a.kwin, a.vwin = sc.dk.data_ptr(), sc.dv.data_ptr(), sc.dkwin.data_ptr(), sc.dvwin.data_ptr()
            a.B, a.nwy, a.nwx, a.heads, a.wse, a.pad = B, H // 16, W // 16, HEADS, 24, 4
            L.check(lib.sr_tr_oca_fold(C.byref(a), 0, _st()), "sr_tr_oca_fold")
        elif dual:
            _edge(side, main, self._ev[1])
            if CONV_WG_SIDE:  # the CAB convs' weight gradients (a 384-workgroup launch of its own since the nn.Linear jobs moved to the wide-tile kernel: 46 us that do not
                # fill the chip) go back to the side stream, beside sr_tr_qkv_bwd and the nn.Linear weight gradients; joined at the end of the block
                _edge(main, side, self._ev[2])
                with torch.cuda.stream(side):
                    _wgrad(jobs_cab)
            else:
                jobs += jobs_cab
        elif self.cab is not None:
            jobs += self._cab_backward(st, B, H, W, T, lib, wa, fa, sc, pp, op)
        _call(lib.sr_tr_qkv_bwd, L.SrTrQkvBwd, "sr_tr_qkv_bwd", dx1=sc.dx1.data_ptr(), x=xin.data_ptr(), dq=sc.dq.data_ptr(), dk=sc.dk.data_ptr(), dv=sc.dv.data_ptr(),
              dn1c=None if self.cab is None else sc.dn1c.data_ptr(), gamma=fa[self.o_g1:].data_ptr(), beta=fa[self.o_b1:].data_ptr(), wstream=wa[self.o_qkvb:].data_ptr(), dx=dx.data_ptr(),
              n1w=op.n1w.data_ptr(), dqkvw=op.dqkvw.data_ptr(), ln_part=pp(self.f_ln1), ldn=CP, shift=self.shift, **g)
        wg = _wg_stream(main.device) if WG_SIDE else None
        if wg is not None:  # the weight gradients only feed the stage's sr_tr_finalize_to: they leave the critical path of the backward pass
            _ev_wait(wg, _ev_record(main))
        with (torch.cuda.stream(wg) if wg is not None else contextlib.nullcontext()):
            _wgrad([
                dict(A=op.dqkvw.data_ptr(), B=op.n1w.data_ptr(), out=pp(self.f_qkv), lda=3 * CP, ldb=CP, Np=3 * CP, Kp=CP, T=T, taps=1, H=H, W=W, ones_col=-1, ks=ks),
                dict(A=op.dx1sw.data_ptr(), B=self.o.data_ptr(), out=pp(self.f_proj), lda=CP, ldb=CP, Np=CP, Kp=CP, T=T, taps=1, H=H, W=W, ones_col=HD, ks=ks),
                dict(A=op.dhw.data_ptr(), B=op.n2w.data_ptr(), out=pp(self.f_fc1), lda=HP, ldb=CP, Np=HP, Kp=CP, T=T, taps=1, H=H, W=W, ones_col=-1, ks=ks),
                dict(A=op.doutw.data_ptr(), B=op.gw.data_ptr(), out=pp(self.f_fc2), lda=CP, ldb=HP, Np=CP, Kp=HP, T=T, taps=1, H=H, W=W, ones_col=-1, ks=ks),
            ] + jobs)
        if wg is not None:
            sc.wg_busy[k] = _ev_record(wg)
        if dual and CONV_WG_SIDE:
            _edge(side, main, self._ev[3])  # the next block's sr_tr_tail_bwd overwrites the operands of the conv weight gradients


    def _cab_backward(self, st: "Stage", B: int, H: int, W: int, T: int, lib, wa, fa, sc, pp, op) -> List[dict]:
        """CAB backward (hat.py:41-52): launches on the current stream; returns its two weight-gradient jobs."""
        w1, b1, w2, b2 = self.ca
        mid_pre = self.mid_pre
        if mid_pre is None:  # (SR_TR_MIDPRE=0: conv1 again)
            mid_pre = sc.mid_pre
            _conv(self.n1, wa[self.o_c1:], fa[self.o_bc1:], mid_pre, B, H, W, CP, 64)
        _call(lib.sr_tr_ca_bwd, L.SrTrCaBwd, "sr_tr_ca_bwd", dgate_part=sc.dgate_part.data_ptr(), pool_partial=self.pool.data_ptr(), w1=w1.data_ptr(), b1=b1.data_ptr(),
              w2=w2.data_ptr(), b2=b2.data_ptr(), dy=op.dyc.data_ptr(), dparam_part=pp(self.f_ca), B=B, H=H, W=W, C=C_REAL, Cp=CP, Cr=w1.shape[0], n_tiles=self.n_tiles,
              parts=H * W // 64, ld=CP, dparam_stride=self.ca_stride, y_scale=self.conv_scale)
        if CAB_BWD_FUSED and ops.cab_supported(CP, 64, CP, L.SR_BF16):
            # conv2's data gradient, GELU', conv1's data gradient as ONE launch: the backward form of sr_cab_fused (SrCab.bwd_pre, ABI v11) -- the same two convs with the
            # flipped / transposed weights around a pointwise step; 49 us of three launches on the longer branch of a HAB's backward -> one
            ops.cab_fused(x=op.dyc.data_ptr(), w1p=wa[self.o_c2t:].data_ptr(), b1=sc.zeros.data_ptr(), w2p=wa[self.o_c1t:].data_ptr(), b2=sc.zeros.data_ptr(),
                          y=sc.dn1c.data_ptr(), pool_partial=None, B=B, H=H, W=W, Cin_p=CP, Cmid_p=64, Cout_p=CP, dtype=L.SR_BF16, tile_rows=0,
                          bwd_pre=mid_pre.data_ptr(), bwd_dmid=op.dmid.data_ptr(), bwd_g=op.mid_g.data_ptr())
        else:
            _conv(op.dyc, wa[self.o_c2t:], None, sc.dmid_g, B, H, W, CP, 64)
            _call(lib.sr_tr_gelu_args, L.SrTrGelu, "sr_tr_gelu", x=mid_pre.data_ptr(), dg=sc.dmid_g.data_ptr(), g=op.mid_g.data_ptr(), dx=op.dmid.data_ptr(), n=T * 64)
            _conv(op.dmid, wa[self.o_c1t:], None, sc.dn1c, B, H, W, 64, CP)
        return [
            dict(A=op.dmid.data_ptr(), B=self.n1.data_ptr(), out=pp(self.f_c1), lda=64, ldb=CP, Np=64, Kp=CP, T=T, taps=9, H=H, W=W, ones_col=-1, ks=self.ks_conv),
            dict(A=op.dyc.data_ptr(), B=op.mid_g.data_ptr(), out=pp(self.f_c2), lda=CP, ldb=64, Np=CP, Kp=64, T=T, taps=9, H=H, W=W, ones_col=60, ks=self.ks_conv),
        ]


class Scratch:
    """Backward scratch shared by all blocks of a model (one block's backward at a time)."""

    def __init__(self, B: int, H: int, W: int, dev, groups: int, oca: bool = True) -> None:
        T = B * H * W
        bf, f32 = torch.bfloat16, torch.float32
        e = lambda *s, dt=bf: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        self.dx1 = e(T, CP, dt=f32)
        self.n2w, self.doutw, self.dOw, self.dx1sw, self.dyc, self.n1w, self.dn1c = (e(T, CP) for _ in range(7))
        self.gw, self.dhw = e(T, HP), e(T, HP)
        self.dOT, self.dq, self.dk, self.dv = (e(T * CP) for _ in range(4))
        self.dqkvw = e(T, 3 * CP)
        self.dgate_part = e(T // 64, CP, dt=f32)
        self.zeros = torch.zeros(CP, dtype=f32, device=dev)  # (bias operands of the CAB's fused data-gradient launch)
        self.lse, self.delta = e(T * HEADS, dt=f32), e(T * HEADS, dt=f32)
        self.groups = groups
        self.mid_pre, self.mid_g, self.dmid_g, self.dmid = (e(T, 64) for _ in range(4))
        if oca:
            n = T // 256 * HEADS * 576 * 32  # OCAB: unfolded neighbourhoods
            self.vwinT, self.dkwin, self.dvwin = e(n), e(n), e(n)
        # The operands of the weight-gradient launches exist twice: consecutive blocks of a backward pass alternate between the two sets, so that a block's
        # sr_tr_wgrad can run on its own stream beside the NEXT block's kernels (which write the other set).  wg_busy[k]: the point on the weight-gradient
        # stream behind the last launch that reads set k (None: free); `turn` counts the blocks of the running backward pass.
        import types

        def opset(first: bool):
            if first:
                return types.SimpleNamespace(n2w=self.n2w, doutw=self.doutw, dx1sw=self.dx1sw, dyc=self.dyc, n1w=self.n1w, gw=self.gw, dhw=self.dhw, dqkvw=self.dqkvw,
                                             mid_g=self.mid_g, dmid=self.dmid)
            return types.SimpleNamespace(n2w=e(T, CP), doutw=e(T, CP), dx1sw=e(T, CP), dyc=e(T, CP), n1w=e(T, CP), gw=e(T, HP), dhw=e(T, HP), dqkvw=e(T, 3 * CP),
                                         mid_g=e(T, 64), dmid=e(T, 64))

        self.ops = [opset(True), opset(False) if WG_SIDE else opset(True)]
        self.wg_busy = [None, None]
        self.turn = 0


def _conv(x: Tensor, wp: Tensor, bias: Optional[Tensor], out: Tensor, B: int, H: int, W: int, cin_p: int, cout_p: int) -> None:
    ops.conv3x3(x=x.data_ptr(), Wp=wp.data_ptr(), bias=None if bias is None else bias.data_ptr(), out=out.data_ptr(), skip=None, pool_partial=None,
                B=B, H=H, W=W, Cin_p=cin_p, Cout_p=cout_p, x_dtype=L.SR_BF16, out_dtype=L.SR_BF16, skip_dtype=0, compute_dtype=L.SR_BF16, act=L.ACT_NONE,
                out_scale=1.0, out_mode=L.OUT_NHWC, ps_r=0, cps_p=0, act_slope=0.0, tile_rows=0)


def conv_ks(B: int, H: int, W: int) -> int:
    """Token slices of a 3x3 weight-gradient job: the halo form runs few, MFMA-dense workgroups whose steps (one 4 x 8 patch each) are a dependent
    chain, so a slice gets ~16 steps instead of the generic form's T / 8 / 32."""
    if HALO and H % 4 == 0 and W % 8 == 0:
        return max(WG_KS, min(256, (B * H * W // 32) // int(os.environ.get("SR_WG_HALO_STEPS", "16"))))
    return WG_KS


HALO = os.environ.get("SR_WG_HALO", "1") != "0"  # A/B knob: 3x3 weight gradients on 2-D patches with one staged halo for all nine taps


def _wgrad(jobs: List[dict]) -> None:
    arr = (L.SrTrWgradJob * len(jobs))()
    for a, j in zip(arr, jobs):
        for k, v in j.items():
            setattr(a, k, v)
        a.halo = int(HALO and j.get("taps") == 9 and j["H"] % 4 == 0 and j["W"] % 8 == 0)
    L.check(L.lib().sr_tr_wgrad(arr, len(jobs), _st()), "sr_tr_wgrad")


def _call(fn, struct, what: str, **kw) -> None:
    a = struct()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(fn(C.byref(a), _st()), what)


class Stage:
    """The blocks of one RHAG (six HABs, then the OCAB) with their gradient map; forward / backward as launch sequences."""

    def __init__(self, fp: FlatParams, habs, ocab, rpi_sa: Optional[Tensor], rpi_oca: Optional[Tensor], conv_scale: float, wa: Arena, fa: Arena, extra=None, ws: int = 16) -> None:
        """extra: a module whose parameters join the stage's gradient map (the RHAG's closing conv, hat.py:356: its weight gradient is finalised with the stage's, so
        that a stage's backward leaves ALL of its layer's gradients complete -- what DistributedDataParallel's buckets need to start reducing)."""
        self.fp, self.wa, self.fa = fp, wa, fa
        mods = list(habs) + ([ocab] if ocab is not None else []) + ([extra] if extra is not None else [])
        params = [p for b in mods for p in b.parameters()]
        p0 = min(fp.off(p) for p in params)
        p1 = max(fp.off(p) + (p.numel() + 3) // 4 * 4 for p in params)
        self.params = params
        self.fm = FinalMap(fp, p0, p1)
        self.blocks = []
        for b in habs:  # the relative-position index: one buffer per model (HAT, hat.py:480-492) or per attention module (SwinIR, swinir.py:56-67)
            r = rpi_sa if rpi_sa is not None else b.attn.relative_position_index
            self.blocks.append(BlockPlan(fp, b, r.detach().cpu().numpy(), r.detach().to(torch.int32).contiguous(), conv_scale, b.shift_size, wa, fa, self.fm, ws=ws))
        if ocab is not None:
            self.blocks.append(BlockPlan(fp, ocab, rpi_oca.detach().cpu().numpy(), rpi_oca.detach().to(torch.int32).contiguous(), 0.0, 0, wa, fa, self.fm, oca=True))
        self.n_habs = len(habs)
        self.geo = None
        self.gen = 0
        self.first = False  # the model's first stage (set by the plan): its forward opens a pass

    def prepare(self, B: int, H: int, W: int, dev, scratch: Scratch) -> None:
        if self.geo == (B, H, W):
            return
        assert self.geo is None, "one geometry per fast-training plan (rebuild the plan for another batch / patch size)"
        for h in self.blocks:
            h.prepare(B, H, W, dev, scratch.groups)
        self.fm.finish(dev)
        self.geo = (B, H, W)
        self.sc = scratch
        self.ts = [torch.empty(B, H, W, CP, dtype=torch.float32, device=dev) for _ in self.blocks]  # block outputs
        self.dxs = [torch.empty(B, H, W, CP, dtype=torch.float32, device=dev) for _ in range(2)]  # gradient w.r.t. a block's input: two static buffers in turn

    def forward(self, x: Tensor, scales: Optional[Tensor]) -> Tensor:
        cur = x
        for i, h in enumerate(self.blocks):
            h.forward(self, cur, self.ts[i], None if (scales is None or h.oca) else scales[i])
            cur = self.ts[i]
        return cur

    def backward(self, x: Tensor, dout: Tensor, scales: Optional[Tensor], G: Optional[Tensor] = None, join: bool = True) -> Tensor:
        """dout: gradient of the stage output; returns the gradient of x (one of the stage's two static buffers); fills G for the stage's parameters.
        join = False: the caller waits for the weight-gradient stream itself (backward_model: once, behind the last stage)."""
        B, H, W = self.geo
        if WG_SIDE:  # a stage is self-contained (it is recorded as a launch plan of its own: no event may cross its border): the operand sets are free again
            main = torch.cuda.current_stream()
            _ev_wait(main, _ev_record(_wg_stream(main.device)))
        self.sc.turn, self.sc.wg_busy = 0, [None, None]
        d = dout.contiguous()
        for i in range(len(self.blocks) - 1, -1, -1):
            h = self.blocks[i]
            dx = self.dxs[i & 1]  # static (a recorded launch plan holds the pointers); the caller consumes the result before this stage's next backward
            h.backward(self, x if i == 0 else self.ts[i - 1], d, dx, None if (scales is None or h.oca) else scales[i])
            d = dx
        if WG_SIDE:  # the stage's gradients: partial sums -> G, behind the stage's weight-gradient launches on their stream
            main = torch.cuda.current_stream()
            wg = _wg_stream(main.device)
            _ev_wait(wg, _ev_record(main))  # (LayerNorm / bias-table / channel-attention partials come from the main-stream kernels)
            with torch.cuda.stream(wg):
                self.fm.run(self.fp.G if G is None else G)
            if join:
                _ev_wait(main, _ev_record(wg))
        else:
            self.fm.run(self.fp.G if G is None else G)
        return d


class _StageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stage, scales, *params):
        x = x.contiguous()
        stage.gen += 1
        if stage.first:  # a forward pass begins: its backward pass decides once where the gradients go (FlatParams.pass_target)
            stage.fp.pass_id = getattr(stage.fp, "pass_id", 0) + 1
        ctx.stage, ctx.gen, ctx.scales, ctx.pass_id = stage, stage.gen, scales, getattr(stage.fp, "pass_id", 0)
        ctx.save_for_backward(x)
        out = stage.forward(x, scales)
        return out.view(out.shape)  # a fresh alias of the stage's static output buffer

    @staticmethod
    def backward(ctx, dout):
        stage = ctx.stage
        if stage.gen != ctx.gen:
            raise RuntimeError("studiosr_amd fast training path: another forward ran before this backward (one forward in flight per model)")
        (x,) = ctx.saved_tensors
        G = stage.fp.pass_target(ctx.pass_id)
        dx = stage.backward(x, dout, ctx.scales, G).clone()  # (the stage's own buffer is static: autograd gets a tensor of its own)
        grads = stage.fp.grad_views(stage.params, G)
        return (dx, None, None) + grads


def run_stage(stage: Stage, x: Tensor, scales: Optional[Tensor]) -> Tensor:
    return _StageFn.apply(x, stage, scales, *stage.params)



# --------------------------------------------------------------------------- 3x3 convolutions outside the blocks
class ConvPlan:
    """One nn.Conv2d(3x3) of the model (common.py:104-105): packed forward weights, packed data-gradient weights (flipped / transposed),
    padded bias, and the gradient map of its weight-gradient GEMM (sr_tr_wgrad, 9 taps).  rows: packed output row -> output channel (the
    PixelShuffle permutation of packing.pixel_shuffle_rows)."""

    def __init__(self, fp: FlatParams, conv, cin_p: int, cout_p: int, wa: Arena, fa: Arena, fm: FinalMap, rows: Optional[np.ndarray] = None, dgrad: bool = True,
                 dgrad_in_p: Optional[int] = None) -> None:
        w, b = conv.weight, conv.bias
        cout, cin = w.shape[:2]
        self.cin, self.cout, self.cin_p, self.cout_p = cin, cout, cin_p, cout_p
        if rows is None:
            rows = np.full(cout_p, -1, dtype=np.int64)
            rows[:cout] = np.arange(cout)
        rows = np.asarray(rows, dtype=np.int64)
        self.o_w = wa.add(pack_conv(fp, w, cin_p, cout_p, rows=rows))
        self.dg_in = dgrad_in_p or cout_p
        if dgrad:
            r2 = np.full(self.dg_in, -1, dtype=np.int64)
            r2[:cout_p] = rows
            self.o_wt = wa.add(pack_conv(fp, w, self.dg_in, cin_p, transpose=True, rows=r2))
        self.o_b = fa.add(pack_vec(fp, b, cout_p, rows))
        self.fp, self.w, self.b, self.rows = fp, w, b, rows

    def prepare(self, fm: FinalMap, B: int, H: int, W: int) -> None:
        """Gradient map of this conv's weight-gradient job at its (own) geometry: [ks][9][cout_p][cin_p] partial sums."""
        fp, w, b, rows, cout, cin, cout_p, cin_p = self.fp, self.w, self.b, self.rows, self.cout, self.cin, self.cout_p, self.cin_p
        self.fm = fm  # the gradient map (and partial-sum arena) this conv's weight-gradient job writes into
        ks = self.ks = conv_ks(B, H, W)
        n_of = np.zeros(cout, dtype=np.int64)
        n_of[rows[rows >= 0]] = np.nonzero(rows >= 0)[0]
        self.f_w = fm.alloc(ks * 9 * cout_p * cin_p)
        tp = np.arange(9)[None, None, :]
        fm.put(fp.pidx(w).reshape(cout, cin, 9), self.f_w + (tp * cout_p + n_of[:, None, None]) * cin_p + np.arange(cin)[None, :, None], 9 * cout_p * cin_p, ks)
        if cin < cin_p:  # the input's first pad channel reads as one: the bias gradient is that column of the centre tap
            self.ones_col, self.f_b = cin, None
            fm.put(fp.pidx(b), self.f_w + (4 * cout_p + n_of) * cin_p + cin, 9 * cout_p * cin_p, ks)
        else:            # no spare input channel: a second job against a constant-one operand
            self.ones_col, self.f_b = -1, fm.alloc(WG_KS * cout_p * 8)
            fm.put(fp.pidx(b), self.f_b + n_of * 8, cout_p * 8, WG_KS)

    def fwd(self, plan, x: Tensor, out: Tensor, B: int, H: int, W: int, *, act: int = L.ACT_NONE, skip: Optional[Tensor] = None, out_mode: int = L.OUT_NHWC, ps_r: int = 0,
            cps_p: int = 0, fin=None) -> None:
        kw = dict(x=x.data_ptr(), Wp=plan.wa.buf[self.o_w:].data_ptr(), bias=plan.fa.buf[self.o_b:].data_ptr(), out=out.data_ptr(), skip=None if skip is None else skip.data_ptr(),
                  pool_partial=None, B=B, H=H, W=W, Cin_p=self.cin_p, Cout_p=self.cout_p, x_dtype=ops._dt(x), out_dtype=ops._dt(out), skip_dtype=0 if skip is None else ops._dt(skip),
                  compute_dtype=L.SR_BF16, act=act, out_scale=1.0, out_mode=out_mode, ps_r=ps_r, cps_p=cps_p, act_slope=0.0, tile_rows=0)
        if fin is not None:
            fs, fb, fc, fh, fw = fin
            kw.update(fin_scale=fs.data_ptr(), fin_bias=fb.data_ptr(), fin_c=fc, fin_h=fh, fin_w=fw)
        ops.conv3x3(**kw)

    def dgrad(self, plan, dy: Tensor, dx: Tensor, B: int, H: int, W: int) -> None:
        ops.conv3x3(x=dy.data_ptr(), Wp=plan.wa.buf[self.o_wt:].data_ptr(), bias=None, out=dx.data_ptr(), skip=None, pool_partial=None, B=B, H=H, W=W, Cin_p=self.dg_in,
                    Cout_p=self.cin_p, x_dtype=ops._dt(dy), out_dtype=ops._dt(dx), skip_dtype=0, compute_dtype=L.SR_BF16, act=L.ACT_NONE, out_scale=1.0, out_mode=L.OUT_NHWC,
                    ps_r=0, cps_p=0, act_slope=0.0, tile_rows=0)

    def wgrad_jobs(self, plan, dy: Tensor, lda: int, x: Tensor, B: int, H: int, W: int) -> List[dict]:
        pp = lambda off: self.fm.part.data_ptr() + 4 * off  # noqa: E731
        T = B * H * W
        f32 = torch.float32
        jobs = [dict(A=dy.data_ptr(), B=x.data_ptr(), out=pp(self.f_w), lda=lda, ldb=self.cin_p, Np=self.cout_p, Kp=self.cin_p, T=T, taps=9, H=H, W=W, ones_col=self.ones_col,
                     ks=self.ks, a_f32=int(dy.dtype == f32), b_f32=int(x.dtype == f32))]
        if self.f_b is not None:
            jobs.append(dict(A=dy.data_ptr(), B=plan.ones.data_ptr(), out=pp(self.f_b), lda=lda, ldb=8, Np=self.cout_p, Kp=8, T=T, taps=1, H=H, W=W, ones_col=-1, ks=WG_KS,
                             a_f32=int(dy.dtype == f32), b_f32=0))
        return jobs


class _ModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan, *params):
        plan.gen += 1
        ctx.plan, ctx.gen = plan, plan.gen
        return plan.forward_model(x)

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        if plan.gen != ctx.gen:
            raise RuntimeError("studiosr_amd fast training path: another forward ran before this backward (one forward in flight per model)")
        G = plan.fp.grad_target()
        plan.backward_model(dout, G)
        return (None, None) + plan.fp.grad_views(plan.fp.params, G)


# The same step as a CHAIN of autograd nodes -- head (conv_first, patch_embed.norm), one per RHAG, tail (norm ... conv_last) -- each replaying its own launch
# plan.  What it buys is the reference's data-parallel overlap (trainer.py:89-91: DistributedDataParallel): a node's backward returns ITS parameters' gradients
# complete (every part of the backward pass finalises its own gradient map), so DDP's bucket hooks fire part by part, last RHAG first, and the bucketed
# all-reduce runs on DDP's stream beside the launches of the RHAGs still to come -- with ONE node all 83 MB became ready at once, behind the last launch.
# Tensors between the nodes are fresh aliases of the plan's static buffers; autograd hands a node's returned gradient tensor to the next node unchanged
# (every intermediate has one consumer), which is checked (and repaired by a copy) against the buffer the recorded plans read.
NODES = os.environ.get("SR_FAST_NODES", "1") != "0"  # A/B knob: 0 = the whole step as ONE autograd node (round 4)


def _check_gen(plan, ctx) -> None:
    if plan.gen != ctx.gen:
        raise RuntimeError("studiosr_amd fast training path: another forward ran before this backward (one forward in flight per model)")


def _into(buf: Tensor, t: Tensor) -> None:
    if t.data_ptr() != buf.data_ptr() or t.dtype != buf.dtype:
        buf.copy_(t.reshape(buf.shape))


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan, *params):
        plan.gen += 1
        ctx.plan, ctx.gen = plan, plan.gen
        plan.begin_forward(x)
        plan.run_fwd_head()
        return plan.t0.view(plan.t0.shape)

    @staticmethod
    def backward(ctx, dt0):
        plan = ctx.plan
        _check_gen(plan, ctx)
        G = plan.pass_target()
        _into(plan.dt_in(-1), dt0)
        plan.run_bwd_head(G)
        return (None, None) + plan.fp.grad_views(plan.head_params, G)


class _StageNodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cur, plan, li, *params):
        ctx.plan, ctx.gen, ctx.li = plan, plan.gen, li
        plan.run_fwd_stage(li)
        return plan.tl[li].view(plan.tl[li].shape)

    @staticmethod
    def backward(ctx, dt):
        plan, li = ctx.plan, ctx.li
        _check_gen(plan, ctx)
        G = plan.pass_target()
        _into(plan.dt_in(li), dt)
        plan.run_bwd_stage(li, G, True)
        nxt = plan.dt_in(li - 1)
        return (nxt.view(nxt.shape), None, None) + plan.fp.grad_views(plan.stages[li].params, G)


class _TailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cur, plan, *params):
        ctx.plan, ctx.gen = plan, plan.gen
        plan.run_fwd_tail()
        return plan.finish_forward()

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        _check_gen(plan, ctx)
        G = plan.begin_backward(dout)
        plan.run_bwd_tail(G)
        dt = plan.dt_in(len(plan.stages) - 1)
        return (dt.view(dt.shape), None) + plan.fp.grad_views(plan.tail_params, G)


def run_model(plan: "HatPlan", x: Tensor) -> Tensor:
    if not NODES:
        return _ModelFn.apply(x, plan, *plan.fp.params)
    t = _HeadFn.apply(x, plan, *plan.head_params)
    for li, st in enumerate(plan.stages):
        t = _StageNodeFn.apply(t, plan, li, *st.params)
    return _TailFn.apply(t, plan, *plan.tail_params)

# --------------------------------------------------------------------------- whole-model plan
class HatPlan:
    """Fast-path plan of a HAT model (FlatParams, the two arenas, one Stage per RHAG: its six HABs and its OCAB) or -- round 5 -- of a SwinIR model of the
    default geometry (swinir.py:258-339: one Stage per RSTB, 8 x 8 windows, no conv branch, no OCAB; everything outside the blocks is the same sequence)."""

    def __init__(self, model) -> None:
        self.model = model
        self.swin = type(model).__name__ == "SwinIR"
        self.ws = int(model.window_size)
        self.fp = FlatParams(model)
        dev = self.fp.P.device
        self.wa, self.fa = Arena(torch.bfloat16), Arena(torch.float32)
        with_oca = not self.swin and os.environ.get("SR_FAST_OCAB", "1") != "0"
        self.with_oca = with_oca
        self.full = (with_oca or self.swin) and os.environ.get("SR_FAST_FULL", "1") != "0"  # the whole model on fused launches (forward_model / backward_model, or the node chain)
        if self.swin:
            self.stages = [Stage(self.fp, list(layer.residual_group.blocks), None, None, None, 0.0, self.wa, self.fa, extra=layer.conv if self.full else None, ws=self.ws)
                           for layer in model.layers]
        else:
            self.stages = [Stage(self.fp, list(layer.residual_group.blocks), layer.residual_group.overlap_attn if with_oca else None, model.relative_position_index_SA,
                                 model.relative_position_index_OCA, model.conv_scale, self.wa, self.fa, extra=layer.conv if self.full else None) for layer in model.layers]
        self.stages[0].first = True
        # ---- everything outside the blocks (hat.py:519-554): conv_first, patch_embed.norm, the RHAG convs, norm, conv_after_body, the upsampling tail
        from . import packing
        from .models.swinir import final_affine, ingest_affine

        m = model
        # Gradient maps outside the blocks, one per part of the backward pass that completes a set of parameters: head (conv_first, patch_embed.norm: last to run),
        # tail (norm, conv_after_body, conv_before_upsample, upsample, conv_last: first to run); a RHAG's closing conv belongs to its stage's map.  Each map spans
        # the whole flat buffer and leaves what it does not own untouched (src = -2).
        def model_map():
            f = FinalMap(self.fp, 0, self.fp.n)
            f.src[:] = -2
            return f

        self.fm_head, self.fm_tail = model_map(), model_map()
        fm = self.fm_tail
        self.c_first = ConvPlan(self.fp, m.conv_first, 32, CP, self.wa, self.fa, self.fm_head, dgrad=False)
        self.c_layers = [ConvPlan(self.fp, layer.conv, CP, CP, self.wa, self.fa, st.fm if self.full else fm) for layer, st in zip(m.layers, self.stages)]
        self.c_after = ConvPlan(self.fp, m.conv_after_body, CP, CP, self.wa, self.fa, fm)
        self.c_before = ConvPlan(self.fp, m.conv_before_upsample[0], CP, 64, self.wa, self.fa, fm)
        self.c_up = []
        for idx, r, c_ps in m.upsample.stages:
            cps_p = packing.round_up(c_ps, 32)
            rows = packing.pixel_shuffle_rows(c_ps, cps_p, r).numpy()
            self.c_up.append((ConvPlan(self.fp, m.upsample[idx], 64, r * r * cps_p, self.wa, self.fa, fm, rows=rows), r, cps_p))
        self.c_last = ConvPlan(self.fp, m.conv_last, 64, 16, self.wa, self.fa, fm, dgrad_in_p=32)
        self.o_pe = (self.fa.add(pack_vec(self.fp, m.patch_embed.norm.weight, CP)), self.fa.add(pack_vec(self.fp, m.patch_embed.norm.bias, CP)))
        self.o_nm = (self.fa.add(pack_vec(self.fp, m.norm.weight, CP)), self.fa.add(pack_vec(self.fp, m.norm.bias, CP)))
        self.fin = final_affine(m.img_range, m.n_colors, dev)
        self.ing = ingest_affine(m.img_range, m.n_colors, dev)
        self.zero3 = torch.zeros(m.n_colors, dtype=torch.float32, device=dev)
        self.wa.finish(dev)
        self.fa.finish(dev)
        self.scratch = None
        self.geo = None
        self.packed_version = None
        self.scales_override = None
        self.gen = 0
        self._plans = {}  # recorded launch plans: (part, DropPath on / off[, gradient buffer]) -> _lib.LaunchPlan
        # parameters by the part of the step that owns (and completes the gradients of) them
        self.head_params = list(m.conv_first.parameters()) + list(m.patch_embed.parameters())
        self.tail_params = (list(m.norm.parameters()) + list(m.conv_after_body.parameters()) + list(m.conv_before_upsample.parameters()) + list(m.upsample.parameters()) +
                            list(m.conv_last.parameters()))
        if self.full:
            owned = [id(p) for p in self.head_params + self.tail_params + [p for st in self.stages for p in st.params]]
            assert len(owned) == len(set(owned)) == len(self.fp.params) and set(owned) == {id(p) for p in self.fp.params}, "every parameter belongs to exactly one part"
    @staticmethod
    def supported(model) -> bool:
        try:
            if type(model).__name__ == "SwinIR":  # the default (classical SR) geometry: embed 180, six heads, 8 x 8 windows, mlp ratio 2, 3x3 convs, pixelshuffle tail
                return (len(set(model.depths)) == 1 and model.embed_dim == C_REAL and model.window_size == 8 and all(h == HEADS for h in model.num_heads) and
                        int(model.embed_dim * model.mlp_ratio) == HID and model.upsampler == "pixelshuffle" and isinstance(model.conv_after_body, torch.nn.Conv2d) and
                        all(isinstance(layer.conv, torch.nn.Conv2d) for layer in model.layers) and next(model.parameters()).is_cuda)
            return (type(model).__name__ == "HAT" and len(set(model.depths)) == 1 and model.embed_dim == C_REAL and model.window_size == 16 and all(h == HEADS for h in model.num_heads) and
                    int(model.embed_dim * model.mlp_ratio) == HID and int(model.window_size * model.overlap_ratio) == 8 and model.embed_dim // model.compress_ratio == 60 and model.embed_dim // model.squeeze_factor == CR and
                    next(model.parameters()).is_cuda)
        except Exception:
            return False

    def prepare(self, B: int, H: int, W: int) -> None:
        if self.geo == (B, H, W):
            return
        assert self.geo is None, "one geometry per fast-training plan"
        dev = self.fp.P.device
        fp, m = self.fp, self.model
        fm, fm_head = self.fm_tail, self.fm_head
        nbw = B * H * W // 256
        # window groups of the OCAB's pass Q (one 4-wave workgroup per (group, head, 64 queries), one workgroup per CU): the largest divisor of the window count
        # that keeps the launch within one residency round of the 256 CUs (HAT x4 step at groups 8 / 16 / 32 / 64: 20.07 / 20.46 / 20.61 / 21.58 ms; SR_TR_GROUPS: A/B knob)
        groups = max(1, min(nbw, int(os.environ.get("SR_TR_GROUPS", str(max(1, (512 if OCA_LSE else 256) // (HEADS * 4)))))))
        while nbw % groups:
            groups -= 1
        self.scratch = Scratch(B, H, W, dev, groups, oca=self.with_oca)
        for c, st in zip(self.c_layers, self.stages):  # (before the stage finishes its map: the RHAG's closing conv is part of it)
            c.prepare(st.fm if self.full else fm, B, H, W)
        for s in self.stages:
            s.prepare(B, H, W, dev, self.scratch)
        self.c_first.prepare(fm_head, B, H, W)
        for c in [self.c_after, self.c_before]:
            c.prepare(fm, B, H, W)
        h, w = H, W
        for cp, r, _ in self.c_up:
            cp.prepare(fm, B, h, w)
            h, w = h * r, w * r
        self.c_last.prepare(fm, B, h, w)
        # LayerNorm partials of patch_embed.norm / norm
        nwg = B * H * W // 64
        self.f_pe, self.f_nm = fm_head.alloc(nwg * 2 * CP), fm.alloc(nwg * 2 * CP)
        for mp, f, norm in ((fm_head, self.f_pe, m.patch_embed.norm), (fm, self.f_nm, m.norm)):
            mp.put(fp.pidx(norm.weight), f + np.arange(C_REAL), 2 * CP, nwg)
            mp.put(fp.pidx(norm.bias), f + CP + np.arange(C_REAL), 2 * CP, nwg)
        fm.finish(dev)
        fm_head.finish(dev)
        T = B * H * W
        bf, f32 = torch.bfloat16, torch.float32
        e = lambda *s, dt=bf: torch.empty(*s, dtype=dt, device=dev)  # noqa: E731
        self.xin = e(B, H, W, 32)
        self.first, self.t0 = e(B, H, W, CP, dt=f32), e(B, H, W, CP, dt=f32)
        self.tl = [e(B, H, W, CP, dt=f32) for _ in self.stages]
        self.tn, self.body, self.feat = e(B, H, W, CP), e(B, H, W, CP), e(B, H, W, 64)
        self.ups = []
        h, w = H, W
        for _, r, cps_p in self.c_up:
            h, w = h * r, w * r
            self.ups.append(e(B, h, w, cps_p))
        self.hw_out = (h, w)
        T2 = B * h * w
        self.ones = torch.ones(T2, 8, dtype=bf, device=dev)
        # backward scratch of the tail
        self.dY, self.dupA, self.dupB = e(B, h, w, 32), e(T2 * 64), e(T2 * 64)
        self.dps = e(T2 * 64)  # un-shuffled gradient: [B, h / r, w / r, r * r * 64] has the same element count as the shuffled tensor
        self.dpre, self.dbody = e(B, H, W, 64), e(B, H, W, CP)
        self.dtn, self.dtA, self.dtB, self.dcv = (e(B, H, W, CP, dt=f32) for _ in range(4))
        self.geo = (B, H, W)

    # ------------------------------------------------------------------ whole-model launch sequences (hat.py:519-554), in the parts the node chain runs them in
    def begin_forward(self, x: Tensor) -> None:
        """Everything of a forward pass that is NOT a recorded launch: the ingest of this batch, the DropPath draw."""
        from .models.train import _drop_rates

        m = self.model
        B, H, W = self.geo
        Hin, Win = x.shape[2], x.shape[3]
        ops.ingest_nchw(x, self.xin, L.PAD_REFLECT if (H != Hin or W != Win) else L.PAD_NONE, *self.ing)
        # DropPath (hat.py:148,192-193): per block, per branch, per image Bernoulli(keep) / keep -- drawn with torch ops into a STATIC buffer (the launches that
        # read it are recorded once)
        dpr = _drop_rates(m)
        nb = len(self.stages[0].blocks) - int(self.with_oca)
        new = None
        if m.training and any(r > 0.0 for r in dpr):
            if getattr(self, "_keep", None) is None:
                self._keep = (1.0 - torch.tensor(dpr, dtype=torch.float32, device=x.device)).reshape(len(self.stages), nb, 1, 1)
            mask = (torch.rand(self._keep.shape[0], self._keep.shape[1], 2, B, device=x.device) < self._keep).to(torch.float32)
            new = mask / self._keep.clamp_min(1e-30)  # keep == 0: the mask is all zero and stays unscaled, as timm's DropPath
        if self.scales_override is not None:  # testing hook: [stages, blocks, 2, B]
            new = self.scales_override.to(torch.float32)
        if new is None:
            self.scales = None
        else:
            if getattr(self, "_scales_buf", None) is None:
                self._scales_buf = torch.empty(len(self.stages), nb, 2, B, dtype=torch.float32, device=x.device)
            self._scales_buf.copy_(new)
            self.scales = self._scales_buf
        for st in self.stages:
            st.gen += 1
        self._out_hw = (Hin, Win)

    def finish_forward(self) -> Tensor:
        """conv_last into a FRESH output tensor (the one launch of the forward whose arguments change from step to step)."""
        m = self.model
        B = self.geo[0]
        Hin, Win = self._out_hw
        s = m.scale
        t = self.ups[-1] if self.ups else self.feat
        h, w = self.hw_out
        out = torch.empty(B, m.n_colors, Hin * s, Win * s, dtype=torch.float32, device=t.device)
        self.c_last.fwd(self, t, out, B, h, w, out_mode=L.OUT_FINAL_NCHW, fin=(*self.fin, m.n_colors, Hin * s, Win * s))
        return out

    def run_fwd_head(self) -> None:
        self._planned(("fwd_head",), self._fwd_head)

    def run_fwd_stage(self, li: int) -> None:
        self._planned(("fwd_stage", li, self.scales is None), lambda: self._fwd_stage(li))

    def run_fwd_tail(self) -> None:
        self._planned(("fwd_tail",), self._fwd_tail)

    def forward_model(self, x: Tensor) -> Tensor:
        self.begin_forward(x)
        self.run_fwd_head()
        for li in range(len(self.stages)):
            self.run_fwd_stage(li)
        self.run_fwd_tail()
        return self.finish_forward()

    def _planned(self, key, body) -> None:
        """Run a launch sequence whose arguments are static: the first time it is recorded (studiosr_amd/_lib.py PlanRecorder -> sr_plan_create), every time
        it is enqueued by sr_plan_run on the current stream.  SR_TR_PLAN=0: enqueue from Python as round 4 did."""
        if not PLAN:
            body()
            return
        plan = self._plans.get(key)
        if plan is None:
            rec = L.PlanRecorder(_st())
            with L.recording(rec):
                body()
            plan = self._plans[key] = rec.finish()
        plan.run(_st())

    def _fwd_head(self) -> None:
        """conv_first + patch_embed.norm (hat.py:531-536): static arguments only."""
        B, H, W = self.geo
        fa = self.fa.buf
        self.c_first.fwd(self, self.xin, self.first, B, H, W)
        ops.layernorm(self.first, self.t0, fa[self.o_pe[0]:self.o_pe[0] + CP], fa[self.o_pe[1]:self.o_pe[1] + CP], C_REAL)

    def _fwd_stage(self, li: int) -> None:
        """One RHAG (hat.py:350-357): its blocks, its closing conv, the residual."""
        B, H, W = self.geo
        st = self.stages[li]
        cur = self.t0 if li == 0 else self.tl[li - 1]
        o = st.forward(cur, None if self.scales is None else self.scales[li])
        self.c_layers[li].fwd(self, o, self.tl[li], B, H, W, skip=cur)

    def _fwd_tail(self) -> None:
        """norm, conv_after_body (+ conv_first's output), conv_before_upsample, the upsampling convs (hat.py:537-552 up to conv_last)."""
        B, H, W = self.geo
        fa = self.fa.buf
        cur = self.tl[-1] if self.tl else self.t0
        ops.layernorm(cur, self.tn, fa[self.o_nm[0]:self.o_nm[0] + CP], fa[self.o_nm[1]:self.o_nm[1] + CP], C_REAL)
        self.c_after.fwd(self, self.tn, self.body, B, H, W, skip=self.first)
        self.c_before.fwd(self, self.body, self.feat, B, H, W, act=L.ACT_LRELU)
        t, h, w = self.feat, H, W
        for (cp, r, cps_p), up in zip(self.c_up, self.ups):
            cp.fwd(self, t, up, B, h, w, out_mode=L.OUT_PIXEL_SHUFFLE, ps_r=r, cps_p=cps_p)
            t, h, w = up, h * r, w * r

    def begin_backward(self, dout: Tensor, G: Optional[Tensor] = None) -> Tensor:
        """Everything of a backward pass that is NOT a recorded launch: WHERE this pass writes the gradients (decided once per pass: FlatParams.grad_target looks at
        the parameters' .grad, which the pass itself sets part by part) and the ingest of the output gradient."""
        G = self.fp.grad_target() if G is None else G
        self._G_pass = (self.gen, G)
        B, H, W = self.geo
        h, w = self.hw_out
        dout = dout.contiguous().to(torch.float32)
        if dout.shape[2] != h or dout.shape[3] != w:  # reflect-padded input (hat.py:544): the output was cropped, its gradient is zero beyond the crop
            dout = torch.nn.functional.pad(dout, (0, w - dout.shape[3], 0, h - dout.shape[2]))
        # d(conv_last output) = dout * range, NHWC, zero beyond the cropped size and in the pad channels
        ops.ingest_nchw(dout, self.dY, L.PAD_NONE, self.fin[0], self.zero3)
        return G

    def pass_target(self) -> Tensor:
        gen, G = getattr(self, "_G_pass", (None, None))
        assert gen == self.gen and G is not None, "the backward pass has not begun (the tail node runs first)"
        return G

    def dt_in(self, li: int) -> Tensor:
        """Gradient of RHAG li's OUTPUT (two static buffers in turn; li = -1: the gradient of the first RHAG's input, which the head part reads)."""
        return (self.dtA, self.dtB)[(len(self.stages) - 1 - li) & 1]

    def run_bwd_tail(self, G: Tensor) -> None:
        self._planned(("bwd_tail", G.data_ptr()), lambda: self._bwd_tail(G))

    def run_bwd_stage(self, li: int, G: Tensor, join: bool) -> None:
        self._planned(("bwd_stage", li, self.scales is None, G.data_ptr(), join), lambda: self._bwd_stage(li, G, join))

    def run_bwd_head(self, G: Tensor) -> None:
        self._planned(("bwd_head", G.data_ptr()), lambda: self._bwd_head(G))

    def backward_model(self, dout: Tensor, G: Optional[Tensor] = None) -> None:
        G = self.begin_backward(dout, G)
        self.run_bwd_tail(G)
        for li in range(len(self.stages) - 1, -1, -1):
            self.run_bwd_stage(li, G, False)
        self.run_bwd_head(G)

    def _bwd_tail(self, G: Tensor) -> None:
        """Backward of conv_last ... norm: leaves the gradient of the last RHAG's output in dt_in(last) and the tail parameters' gradients complete in G."""
        B, H, W = self.geo
        lib = L.lib()
        fa = self.fa.buf
        h, w = self.hw_out
        T = B * H * W
        self.scratch.turn, self.scratch.wg_busy = 0, [None, None]
        jobs = self.c_last.wgrad_jobs(self, self.dY, 32, self.ups[-1] if self.ups else self.feat, B, h, w)
        d_cur = self.dupA[: B * h * w * 64].view(B, h, w, 64)
        self.c_last.dgrad(self, self.dY, d_cur, B, h, w)
        _wgrad(jobs)
        other = self.dupB
        for si in range(len(self.c_up) - 1, -1, -1):
            cp, r, cps_p = self.c_up[si]
            hi, wi = h // r, w // r
            dps = self.dps[: B * hi * wi * r * r * cps_p].view(B, hi, wi, r * r * cps_p)
            _call(lib.sr_tr_unshuffle_args, L.SrTrUnshuffle, "sr_tr_unshuffle", src=d_cur.data_ptr(), dst=dps.data_ptr(), B=B, H=hi, W=wi, cps=cps_p, r=r)
            xin = self.ups[si - 1] if si > 0 else self.feat
            jobs = cp.wgrad_jobs(self, dps, r * r * cps_p, xin, B, hi, wi)
            d_prev = other[: B * hi * wi * 64].view(B, hi, wi, 64)
            cp.dgrad(self, dps, d_prev, B, hi, wi)
            _wgrad(jobs)
            other = self.dupA if other is self.dupB else self.dupB
            d_cur, h, w = d_prev, hi, wi
        _call(lib.sr_tr_lrelu_bwd_args, L.SrTrLreluBwd, "sr_tr_lrelu_bwd", dy=d_cur.data_ptr(), y=self.feat.data_ptr(), dx=self.dpre.data_ptr(), slope=0.01, n=T * 64)
        jobs = self.c_before.wgrad_jobs(self, self.dpre, 64, self.body, B, H, W)
        self.c_before.dgrad(self, self.dpre, self.dbody, B, H, W)
        jobs += self.c_after.wgrad_jobs(self, self.dbody, CP, self.tn, B, H, W)
        self.c_after.dgrad(self, self.dbody, self.dtn, B, H, W)
        _wgrad(jobs)
        last = self.tl[-1] if self.tl else self.t0
        _call(lib.sr_tr_ln_bwd, L.SrTrLnBwd, "sr_tr_ln_bwd", x=last.data_ptr(), dy=self.dtn.data_ptr(), gamma=fa[self.o_nm[0]:].data_ptr(), dskip=None,
              dx=self.dt_in(len(self.stages) - 1).data_ptr(), ln_part=self.fm_tail.part.data_ptr() + 4 * self.f_nm, M=T, C=C_REAL, Cp=CP, ld=CP, dy_bf16=0, dskip_bf16=0, eps=1e-5)
        self.fm_tail.run(G)

    def _bwd_stage(self, li: int, G: Tensor, join: bool) -> None:
        """Backward of RHAG li: dt_in(li) -> dt_in(li - 1); the RHAG's gradients (blocks AND closing conv) complete in G (join: also waited for, when the
        weight gradients run on their own stream)."""
        B, H, W = self.geo
        lib = L.lib()
        T = B * H * W
        st = self.stages[li]
        dt, nxt = self.dt_in(li), self.dt_in(li - 1)
        cur_in = self.t0 if li == 0 else self.tl[li - 1]
        jobs = self.c_layers[li].wgrad_jobs(self, dt, CP, st.ts[-1], B, H, W)
        self.c_layers[li].dgrad(self, dt, self.dcv, B, H, W)
        _wgrad(jobs)
        dx = st.backward(cur_in, self.dcv, None if self.scales is None else self.scales[li], G, join=join)
        _call(lib.sr_tr_add_args, L.SrTrAdd, "sr_tr_add", a=dx.data_ptr(), b=dt.data_ptr(), b_dtype=L.SR_F32, out=nxt.data_ptr(), n=T * CP)

    def _bwd_head(self, G: Tensor) -> None:
        """Backward of patch_embed.norm + conv_first (the gradient of conv_first's output also arrives from conv_after_body's skip: dbody)."""
        B, H, W = self.geo
        lib = L.lib()
        fa = self.fa.buf
        T = B * H * W
        dfirst = self.dtn  # (free again)
        _call(lib.sr_tr_ln_bwd, L.SrTrLnBwd, "sr_tr_ln_bwd", x=self.first.data_ptr(), dy=self.dt_in(-1).data_ptr(), gamma=fa[self.o_pe[0]:].data_ptr(), dskip=self.dbody.data_ptr(),
              dx=dfirst.data_ptr(), ln_part=self.fm_head.part.data_ptr() + 4 * self.f_pe, M=T, C=C_REAL, Cp=CP, ld=CP, dy_bf16=0, dskip_bf16=1, eps=1e-5)
        _wgrad(self.c_first.wgrad_jobs(self, dfirst, CP, self.xin, B, H, W))
        self.fm_head.run(G)
        if WG_SIDE:  # the stages' weight gradients and finalize launches are through before the gradients are handed to autograd
            main = torch.cuda.current_stream()
            _ev_wait(main, _ev_record(_wg_stream(main.device)))

    def pack(self) -> None:
        """Packed operands <- current parameters: two sr_tr_gather launches (~130 MB of traffic) on EVERY recorded forward.  (Until round 5 the
        parameters' version counters decided; torch.optim.Adam(fused=True) updates parameters without bumping them, so the second step of such a run
        trained on the first step's weights.  The launches cost less than reading 860 version counters on the host.)"""
        self.wa.gather(self.fp.P)
        self.fa.gather(self.fp.P)
        self.packed_version = True


def get_plan(model) -> Optional[HatPlan]:
    plan = getattr(model, "_fast_plan", None)
    if plan is not None and plan.fp.intact():
        return plan
    if not HatPlan.supported(model):
        return None
    plan = HatPlan(model)
    object.__setattr__(model, "_fast_plan", plan)
    return plan
