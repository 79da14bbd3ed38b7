"""Row-strip sharding of ONE large image across the GPUs of a node (SwinIR x4 on a
2048x2048 LR image over 8 GPUs), with the per-layer neighbour exchange the receptive field needs:

  * every 3x3 conv (conv_first, RSTB.conv, conv_after_body, conv_before_upsample, Upsampler, conv_last -- swinir.py:241,
    290,316-326) reads a 1-row halo from the strip above and below (zeros at the true image border = the conv's padding);
  * an unshifted window block (W-MSA) needs nothing: strips are cut on window-row boundaries;
  * a shifted block (SW-MSA, `torch.roll(x, (-s, -s))`, swinir.py:154) needs `s` rows from the strip BELOW (cyclically:
    the last strip receives the first rows of strip 0, exactly the rows the reference's roll wraps around).  The strip
    buffer then holds rows [r0+s, r1+s) of the image, i.e. this rank's slice of the rolled tensor; the kernels run on it
    with the row roll switched off (SR_Y_STRIP) and the row half of the -100 mask only on the last strip
    (SR_Y_STRIP_LAST, common.py:250-274); the column half of roll and mask is untouched.  Afterwards the last `s` rows go
    back down (`torch.roll(+s)`, swinir.py:168).  LayerNorm / MLP / Linear are per token and need nothing.

One process per GPU: `DistStripComm` moves the halos with paired isend/irecv between strip neighbours (RCCL point to
point over the direct xGMI link; gloo in the CPU tests).  `LocalStripComm(n)` keeps all n strips in ONE process and
"exchanges" by local copies -- same orchestration code, used to check n-strip results against the unsharded forward on a
single GPU.  The orchestration (`swinir_forward_strips`) is written once, over a list of local strips.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops, packing
from .models.common import conv_call
from .runtime import Workspace, compute_dtype

Tensor = torch.Tensor


def strip_partition(n_window_rows: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous balanced [first, last) window-row range of every rank (the first n % world ranks get one more)."""
    if n_window_rows < world_size:
        raise ValueError(f"{n_window_rows} window rows cannot be cut into {world_size} non-empty strips")
    base, extra = divmod(n_window_rows, world_size)
    out, start = [], 0
    for r in range(world_size):
        stop = start + base + (1 if r < extra else 0)
        out.append((start, stop))
        start = stop
    return out


# --------------------------------------------------------------------------- communicators
class StripComm:
    """`local_ranks` strips live in this process.  All methods take one tensor per local strip, in that order."""

    world: int
    local_ranks: List[int]

    def shift_up(self, send: Sequence[Tensor], recv: Sequence[Tensor], cyclic: bool) -> None:
        """send[i] travels to strip rank-1; recv[i] is filled with what strip rank+1 sent (zeros when there is none)."""
        raise NotImplementedError

    def shift_down(self, send: Sequence[Tensor], recv: Sequence[Tensor], cyclic: bool) -> None:
        """send[i] travels to strip rank+1; recv[i] is filled with what strip rank-1 sent (zeros when there is none)."""
        raise NotImplementedError

    def gather_rows(self, parts: Sequence[Tensor], dim: int) -> Tensor:
        """Concatenate every strip's tensor along `dim` in rank order; the result is returned on every process."""
        raise NotImplementedError


class LocalStripComm(StripComm):
    """All `world` strips in this process (single GPU): exchanges are local copies."""

    def __init__(self, world: int) -> None:
        self.world = world
        self.local_ranks = list(range(world))

    def _shift(self, send, recv, cyclic: bool, step: int) -> None:
        n = self.world
        for r in range(n):
            src = r - step  # the strip whose `send` lands in strip r
            if 0 <= src < n:
                recv[r].copy_(send[src])
            elif cyclic:
                recv[r].copy_(send[src % n])
            else:
                recv[r].zero_()

    def shift_up(self, send, recv, cyclic):
        self._shift(send, recv, cyclic, -1)

    def shift_down(self, send, recv, cyclic):
        self._shift(send, recv, cyclic, +1)

    def gather_rows(self, parts, dim):
        return torch.cat(list(parts), dim=dim)


class DistStripComm(StripComm):
    """One strip per process (one process per GPU); neighbours exchange with paired isend/irecv."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None) -> None:
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("DistStripComm needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local_ranks = [self.rank]

    def _peer(self, r: int) -> int:
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _host_staged(self, t: Tensor) -> bool:
        """RCCL ("nccl") p2p is ordered on the device stream.  Any other backend (gloo: CPU tests, several processes on one GPU) moves
        device tensors through the host without ordering against the compute stream, so the halo is staged explicitly: wait for
        the producing kernels, send from / receive into host buffers, copy back on the compute stream."""
        return t.is_cuda and dist.get_backend(self.group) != "nccl"

    def _shift(self, send, recv, cyclic: bool, step: int) -> None:
        n, r = self.world, self.rank
        (s,), (d,) = send, recv
        if n == 1:
            d.copy_(s) if cyclic else d.zero_()
            return
        dst, src = r + step, r - step
        if cyclic:
            dst, src = dst % n, src % n
        staged = self._host_staged(s)
        if staged:
            torch.cuda.current_stream(s.device).synchronize()
            s_buf, d_buf = s.detach().cpu(), torch.empty(d.shape, dtype=d.dtype)
        else:
            s_buf, d_buf = s, d
        p2p = []
        if 0 <= dst < n:
            p2p.append(dist.P2POp(dist.isend, s_buf, self._peer(dst), self.group))
        if 0 <= src < n:
            p2p.append(dist.P2POp(dist.irecv, d_buf, self._peer(src), self.group))
        else:
            d.zero_()
        if p2p:
            for req in dist.batch_isend_irecv(p2p):
                req.wait()
        if staged and 0 <= src < n:
            d.copy_(d_buf)

    def shift_up(self, send, recv, cyclic):
        self._shift(send, recv, cyclic, -1)

    def shift_down(self, send, recv, cyclic):
        self._shift(send, recv, cyclic, +1)

    def gather_rows(self, parts, dim):
        (mine,) = parts
        out_dev = mine.device
        if self._host_staged(mine):
            torch.cuda.current_stream(out_dev).synchronize()
            mine = mine.cpu()
        mine = mine.movedim(dim, 0).contiguous()
        n_rows = torch.tensor([mine.shape[0]], dtype=torch.int64, device=mine.device)
        counts = [torch.empty_like(n_rows) for _ in range(self.world)]
        dist.all_gather(counts, n_rows, group=self.group)
        counts = [int(c.item()) for c in counts]
        pad = torch.zeros((max(counts),) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        pad[: mine.shape[0]] = mine
        gathered = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(gathered, pad, group=self.group)
        return torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0).movedim(0, dim).to(out_dev)


_SIDE_STREAMS = {}


def _side_stream(device) -> "torch.cuda.Stream":
    """One halo-exchange stream per device (created on first use; streams are cheap but not free to create per forward)."""
    key = torch.device(device).index
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


# --------------------------------------------------------------------------- strip buffers
class _Strip:
    """Rows [r0, r1) of a [1, H, W, C] NHWC image, stored with `top` margin rows above and `bot` below so that halo rows
    land next to the strip and every kernel input is one contiguous row range (batch 1: a row slice is contiguous)."""

    def __init__(self, ws: Workspace, name: str, rows: int, W: int, C: int, dtype: torch.dtype, top: int = 1, bot: int = 1) -> None:
        self.rows, self.top, self.bot = rows, top, bot
        self.buf = ws.get(name, (1, top + rows + bot, W, C), dtype)

    def view(self, a: int, b: int) -> Tensor:
        """Rows [a, b) relative to the strip's first own row (a may be negative: the top margin)."""
        assert -self.top <= a <= b <= self.rows + self.bot
        return self.buf[:, self.top + a : self.top + b]

    @property
    def own(self) -> Tensor:
        return self.view(0, self.rows)


def _exchange_conv_halo(comm: StripComm, strips: List[_Strip]) -> None:
    """Row -1 <- last row of the strip above, row `rows` <- first row of the strip below; zeros at the image border."""
    comm.shift_down([s.view(s.rows - 1, s.rows) for s in strips], [s.view(-1, 0) for s in strips], cyclic=False)
    comm.shift_up([s.view(0, 1) for s in strips], [s.view(s.rows, s.rows + 1) for s in strips], cyclic=False)


def _conv_strips(comm: StripComm, src: List[_Strip], dst: List[Tensor], wp: Tensor, b: Tensor, cdt: torch.dtype, skip: Optional[List[Tensor]] = None, **kw) -> None:
    """3x3 conv of every local strip on rows [-1, rows+1): the halo rows are exchanged first; output rows computed from
    beyond the halo (the first and last output row of the extended range) are garbage and never read."""
    _exchange_conv_halo(comm, src)
    for i, s in enumerate(src):
        conv_call(s.view(-1, s.rows + 1), wp, b, dst[i], cdt, skip=None if skip is None else skip[i], **kw)


# --------------------------------------------------------------------------- SwinIR
def swinir_forward_strips(model, x: Tensor, comm: StripComm) -> Tensor:
    """SwinIR.forward (swinir.py:342-372) of ONE image [1, n_colors, H, W], row-strip sharded over `comm.world` strips.
    Every process passes the full LR image (it is tiny next to the activations) and gets the full HR image back."""
    from .models import swinir as SW

    x = model._check_input(x)
    if x.shape[0] != 1:
        raise ValueError("strip sharding handles one image (batch 1); batches of tiles go through TileParallel")
    cdt = compute_dtype(model.precision)
    P = model._get_packed(cdt)
    ws_ = model._workspace(x.device)
    _, _, H, W = x.shape
    w8, C, s = model.window_size, model.embed_dim, model.scale
    if model.training:
        Hp, Wp, pad_mode = H + (w8 - H % w8) % w8, W + (w8 - W % w8) % w8, L.PAD_REFLECT
        if Hp - H >= H or Wp - W >= W:
            raise RuntimeError("Padding size should be less than the corresponding input dimension (reflect pad)")
    else:
        Hp, Wp, pad_mode = (H // w8 + 1) * w8, (W // w8 + 1) * w8, L.PAD_EVAL_MIRROR
    Cp = P["layers"][0]["geo"].Cp if P["layers"] else packing.round_up(C, 64)
    parts = strip_partition(Hp // w8, comm.world)
    ranks = comm.local_ranks
    rows = [(parts[r][1] - parts[r][0]) * w8 for r in ranks]
    row0 = [parts[r][0] * w8 for r in ranks]
    shifts = [bp["shift"] for lp in P["layers"] for bp in lp["blocks"]]
    bot = max([1] + shifts)
    if min(rows) < bot:
        raise ValueError("a strip is shorter than the cyclic shift")

    # the whole padded LR image is ingested on every rank (3 channels); one zero row above and below = the conv padding
    xin = ws_.get("strip.xin", (1, Hp + 2, Wp, 32), cdt)
    xin[:, 0].zero_()
    xin[:, Hp + 1].zero_()
    ops.ingest_nchw(x, xin[:, 1 : Hp + 1], pad_mode, *P["ing"])

    def strips(name: str, ch: int, dtype: torch.dtype, scale: int = 1, top: int = 1, bot_: int = 1) -> List[_Strip]:
        return [_Strip(ws_, f"strip.{name}.{r}", n * scale, Wp * scale, ch, dtype, top, bot_) for r, n in zip(ranks, rows)]

    first, ta, tb = strips("first", Cp, torch.float32), strips("ta", Cp, torch.float32, bot_=bot), strips("tb", Cp, torch.float32, bot_=bot)
    for i, n in enumerate(rows):  # conv_first: the halo rows come straight out of the full ingested image
        conv_call(xin[:, row0[i] : row0[i] + n + 2], *P["first"], first[i].view(-1, n + 1), cdt)
        ops.layernorm(first[i].own, ta[i].own, *P["pe_norm"], C)

    # Halo exchanges of the SW-MSA blocks run on a SIDE stream beside the windows that do not need them: the rolled
    # strip's interior window rows are launched first, the last window row (the only one that reads the rows received from the strip
    # below) after the exchange has landed; the roll-back exchange of its result overlaps with the interior window rows of the NEXT block,
    # whose first window row waits for it.  Same windows, same kernels, same operands: results stay bit-identical to the unsharded forward.
    overlap = os.environ.get("SR_STRIPS_OVERLAP", "1") != "0" and x.is_cuda
    main = torch.cuda.current_stream(x.device) if x.is_cuda else None
    side = _side_stream(x.device) if overlap else None
    pending = [None]  # event of a roll-back exchange whose rows [0, shift) of `tb` the next consumer must wait for

    def on_side(fn):
        """Run the exchange `fn` on the side stream, ordered after everything enqueued on the compute stream so far; returns its completion event."""
        if not overlap:
            fn()
            return None
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            fn()
            done = torch.cuda.Event()
            done.record(side)
        return done

    def wait(ev) -> None:
        if ev is not None:
            main.wait_event(ev)

    def settle() -> None:  # before any consumer that reads whole strips (convs, LayerNorm)
        wait(pending[0])
        pending[0] = None

    for lp in P["layers"]:
        geo = lp["geo"]
        cur = ta
        for bp in lp["blocks"]:
            sh = bp["shift"]
            if sh == 0:
                if pending[0] is None:
                    for i in range(len(ranks)):
                        SW.run_swin_block(bp, geo, cur[i].own, tb[i].own, ws_, cdt, 0)
                else:  # rows [0, shift) of `cur` are still in flight: every window row but the first goes ahead
                    for i in range(len(ranks)):
                        if cur[i].rows > w8:
                            SW.run_swin_block(bp, geo, cur[i].view(w8, cur[i].rows), tb[i].view(w8, tb[i].rows), ws_, cdt, 0)
                    settle()
                    for i in range(len(ranks)):
                        SW.run_swin_block(bp, geo, cur[i].view(0, w8), tb[i].view(0, w8), ws_, cdt, 0)
            else:
                settle()
                # rows [sh, rows+sh) of the buffer = this strip of roll(x, -sh): own rows sh.. + the first sh rows of the strip below
                got = on_side(lambda: comm.shift_up([c.view(0, sh) for c in cur], [c.view(c.rows, c.rows + sh) for c in cur], cyclic=True))
                for i, r in enumerate(ranks):  # interior window rows: own rows only, never the wrapped window row
                    n = cur[i].rows
                    if overlap and n > w8:
                        SW.run_swin_block(bp, geo, cur[i].view(sh, n + sh - w8), tb[i].view(sh, n + sh - w8), ws_, cdt, sh, y_mode=L.Y_STRIP)
                wait(got)
                for i, r in enumerate(ranks):
                    y_mode = L.Y_STRIP_LAST if r == comm.world - 1 else L.Y_STRIP
                    n = cur[i].rows
                    a0 = n + sh - w8 if overlap else sh  # without overlap: the whole rolled strip in one launch
                    SW.run_swin_block(bp, geo, cur[i].view(a0, n + sh), tb[i].view(a0, n + sh), ws_, cdt, sh, y_mode=y_mode)
                # roll back: the last sh rows of the rolled strip are the first sh rows of the strip below
                pending[0] = on_side(lambda: comm.shift_down([t.view(t.rows, t.rows + sh) for t in tb], [t.view(0, sh) for t in tb], cyclic=True))
            cur = tb
        settle()
        if cur is ta:  # zero-depth RSTB: convolve a copy, `ta` stays the skip
            tc = strips("tc", Cp, torch.float32, bot_=bot)
            for i in range(len(ranks)):
                tc[i].own.copy_(ta[i].own)
            cur = tc
        ext = [t.view(-1, t.rows + 1) for t in ta]
        _conv_strips(comm, cur, ext, *lp["conv"], cdt, skip=ext)  # ta = conv(cur) + ta   (swinir.py:245-246)

    normed = strips("normed", Cp, cdt)
    for i in range(len(ranks)):
        ops.layernorm(ta[i].own, normed[i].own, *P["norm"], C)
    body = strips("body", Cp, cdt)
    _conv_strips(comm, normed, [b_.view(-1, b_.rows + 1) for b_ in body], *P["after_body"], cdt, skip=[f.view(-1, f.rows + 1) for f in first])

    fin_scale, fin_bias = P["fin"]
    hr: List[Tensor] = []
    if model.upsampler == "pixelshuffle":
        feat = strips("feat", 64, cdt)
        _conv_strips(comm, body, [f.view(-1, f.rows + 1) for f in feat], *P["before_up"], cdt, act=L.ACT_LRELU)
        cur, mult = feat, 1
        for si, (wp, b, r, cps_p) in enumerate(P["up"]):
            mult *= r
            nxt = strips(f"up{si}", cps_p, cdt, scale=mult, top=r, bot_=r)  # conv over rows [-1, n+1) shuffles to r*(n+2) rows
            _conv_strips(comm, cur, [t.buf for t in nxt], wp, b, cdt, out_mode=L.OUT_PIXEL_SHUFFLE, ps_r=r, cps_p=cps_p)
            cur = nxt
        outs = [torch.empty(1, model.n_colors, c.rows + 2, W * s, dtype=torch.float32, device=x.device) for c in cur]
        _exchange_conv_halo(comm, cur)
        for i, c in enumerate(cur):
            conv_call(c.view(-1, c.rows + 1), *P["last"], outs[i], cdt, out_mode=L.OUT_FINAL_NCHW,
                      fin=(fin_scale, fin_bias, model.n_colors, c.rows + 2, W * s), cout_p=16)
            hr.append(outs[i][:, :, 1 : c.rows + 1])
    elif model.upsampler == "pixelshuffledirect":
        wp, b, r, cps_p = P["up"][0]
        _exchange_conv_halo(comm, body)
        for i, bd in enumerate(body):
            n_out = r * (bd.rows + 2)
            out = torch.empty(1, model.n_colors, n_out, W * s, dtype=torch.float32, device=x.device)
            conv_call(bd.view(-1, bd.rows + 1), wp, b, out, cdt, out_mode=L.OUT_FINAL_NCHW, ps_r=r, cps_p=cps_p,
                      fin=(fin_scale, fin_bias, model.n_colors, n_out, W * s), cout_p=r * r * cps_p)
            hr.append(out[:, :, r : r + r * bd.rows])
    else:
        raise NotImplementedError(f"strip sharding of upsampler {model.upsampler!r}")
    full = comm.gather_rows(hr, dim=2)
    return full[:, :, : H * s].contiguous()
