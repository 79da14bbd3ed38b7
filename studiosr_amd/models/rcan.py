"""RCAN on the MI355X HIP hot path (reference: studiosr/models/rcan.py).

Per RCAB (rcan.py:11-24): conv+ReLU -> conv (its epilogue also emits per-tile channel sums) ->
channel-attention gate kernel: mean -> 1x1 -> ReLU -> 1x1 -> sigmoid -> y*s + x (common.py:156-170).
Per ResidualGroup (:27-36): 20 RCABs + conv + skip.  Head / tail as EDSR.
"""
from __future__ import annotations

import os
from typing import Dict

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops, packing
from ..runtime import capturing_or_warming_up, compute_dtype, knob, sr_dtype
from .common import Model, Upsampler, conv2d, conv_call, pack_upsampler, run_upsampler
from .edsr import MeanShift, mean_shift_affine

Tensor = torch.Tensor


class ChannelAttention(nn.Module):
    """Parameters of common.py:156-165 (keys conv_du.0 / conv_du.2)."""

    def __init__(self, channel: int, reduction: int = 16) -> None:
        super().__init__()
        self.conv_du = nn.Sequential(nn.Conv2d(channel, channel // reduction, 1), nn.ReLU(True), nn.Conv2d(channel // reduction, channel, 1), nn.Sigmoid())


class RCAB(nn.Module):
    def __init__(self, n_feat: int, kernel_size: int, reduction: int) -> None:
        super().__init__()
        self.body = nn.Sequential(conv2d(n_feat, n_feat, kernel_size), nn.ReLU(True), conv2d(n_feat, n_feat, kernel_size), ChannelAttention(n_feat, reduction))


class ResidualGroup(nn.Module):
    def __init__(self, n_feat: int, kernel_size: int, reduction: int, n_resblocks: int) -> None:
        super().__init__()
        self.body = nn.Sequential(*[RCAB(n_feat, kernel_size, reduction) for _ in range(n_resblocks)], conv2d(n_feat, n_feat, kernel_size))


def pack_ca(w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor):
    f = lambda t: t.detach().to(torch.float32).reshape(t.shape[0], -1).contiguous()  # noqa: E731
    return f(w1), b1.detach().float().contiguous(), f(w2), b2.detach().float().contiguous()


def run_channel_attention(ca, y: Tensor, pool: Tensor, n_tiles: int, C: int, out: Tensor, skip=None, skip2=None, y_scale: float = 1.0) -> None:
    w1, b1, w2, b2 = ca
    B, H, W, Cp = y.shape
    ops.channel_attention(
        y=y.data_ptr(), pool_partial=pool.data_ptr(), w1=w1.data_ptr(), b1=b1.data_ptr(), w2=w2.data_ptr(), b2=b2.data_ptr(),
        skip=None if skip is None else skip.data_ptr(), out=out.data_ptr(), B=B, H=H, W=W, C=C, C_p=Cp, Cr=w1.shape[0], n_tiles=n_tiles,
        y_dtype=sr_dtype(y.dtype), skip_dtype=sr_dtype(skip.dtype) if skip is not None else 0, out_dtype=sr_dtype(out.dtype), y_scale=y_scale,
        skip2=None if skip2 is None else skip2.data_ptr(), skip2_dtype=sr_dtype(skip2.dtype) if skip2 is not None else 0,
    )


class RCAN(Model):
    def __init__(self, scale: int = 4, n_colors: int = 3, img_range: float = 1.0, n_feats: int = 64, n_resblocks: int = 20,
                 n_resgroups: int = 10, reduction: int = 16) -> None:
        super().__init__(scale, n_colors, img_range)
        self.n_feats, self.n_resblocks, self.n_resgroups, self.reduction = n_feats, n_resblocks, n_resgroups, reduction
        self.sub_mean = MeanShift(img_range)
        self.add_mean = MeanShift(img_range, sign=1)
        k = 3
        self.head = nn.Sequential(conv2d(n_colors, n_feats, k))
        self.body = nn.Sequential(*[ResidualGroup(n_feats, k, reduction, n_resblocks) for _ in range(n_resgroups)], conv2d(n_feats, n_feats, k))
        self.tail = nn.Sequential(Upsampler(scale, n_feats), conv2d(n_feats, n_colors, k))

    def _pack(self, dt: torch.dtype) -> Dict:
        F = self.n_feats
        Fp = packing.round_up(F, 32)
        ident = packing.identity_idx(F, Fp)
        pc = lambda m, cin_p=Fp: packing.pack_conv3x3(m.weight, m.bias, cin_p, ident, dt)  # noqa: E731
        P: Dict = {"Fp": Fp, "ing": mean_shift_affine(self.sub_mean), "fin": mean_shift_affine(self.add_mean)}
        P["head"] = pc(self.head[0], 32)
        P["groups"] = []
        for gi in range(self.n_resgroups):
            grp = self.body[gi]
            blocks = []
            for bi in range(self.n_resblocks):
                b = grp.body[bi].body
                du = b[3].conv_du
                blocks.append((pc(b[0]), pc(b[2]), pack_ca(du[0].weight, du[0].bias, du[2].weight, du[2].bias)))
            P["groups"].append((blocks, pc(grp.body[self.n_resblocks])))
        P["body_last"] = pc(self.body[self.n_resgroups])
        P["up"] = pack_upsampler(self.tail[0], Fp, dt)
        P["tail"] = packing.pack_conv3x3(self.tail[1].weight, self.tail[1].bias, P["up"][-1][3], packing.identity_idx(self.n_colors, 16), dt)
        return P

    def _run_groups(self, P: Dict, h: Tensor, ws_, cdt, keep_all: bool = False):
        """The residual groups (rcan.py:27-36) on the fp32 NHWC stream h; returns (last group output, [every group output]).
        keep_all: every group output gets its own buffer (HAN's layer attention reads all of them)."""
        B, H, W, Fp = h.shape
        f32 = torch.float32
        # bf16 or split operands ("fp32x3", round 5), 64 (padded) channels: conv-ReLU-conv of an RCAB is ONE launch (sr_rcab_conv_pair, the intermediate stays in LDS)
        from ..runtime import x3_active

        fused_pair = (cdt == torch.bfloat16 or (x3_active() and knob("SR_RCAB_X3", "1") != "0")) and Fp == 64
        n_tiles = ops.rcab_pool_tiles(H, W) if fused_pair else ops.conv_pool_tiles(H, W, Fp, sr_dtype(cdt))
        pool = ws_.get("pool", (B, n_tiles, Fp), f32)
        mid = ws_.get("mid", (B, H, W, Fp), cdt)
        y = ws_.get("y", (B, H, W, Fp), f32)
        ga, gb = (None, None) if keep_all else (ws_.get("ga", (B, H, W, Fp), f32), ws_.get("gb", (B, H, W, Fp), f32))
        ra, rb = ws_.get("ra", (B, H, W, Fp), f32), ws_.get("rb", (B, H, W, Fp), f32)
        # one launch per RCAB: block n+1 applies block n's channel-attention tail (gate * y + skip) while it stages its halo
        # (sr_rcab_conv_pair's gated input); y / pool alternate because block n+1 reads block n's while writing its own
        chained = fused_pair and self.n_feats // self.reduction <= 8
        if chained:
            # the branch output y is stored in the compute dtype (bf16; its pool sums are taken from the fp32 accumulators first): it is
            # read once, multiplied by the gate and added to the fp32 skip stream -- 25 % less HBM traffic per block, >= 50 dB kept
            y, y2, pool2 = ws_.get("yc", (B, H, W, Fp), cdt), ws_.get("y2", (B, H, W, Fp), cdt), ws_.get("pool2", (B, n_tiles, Fp), f32)
        g = h
        feats = []
        for gi, (blocks, gconv) in enumerate(P["groups"]):
            r = g
            if chained:
                ys, pools, prev = (y, y2), (pool, pool2), None  # prev = (skip, y, pool, ca) of the block whose tail is still pending
                for i, (c1, c2, ca) in enumerate(blocks):
                    yc, pc = ys[i & 1], pools[i & 1]
                    kw = dict(w1p=c1[0].data_ptr(), b1=c1[1].data_ptr(), w2p=c2[0].data_ptr(), b2=c2[1].data_ptr(), y=yc.data_ptr(),
                              pool_partial=pc.data_ptr(), B=B, H=H, W=W, C_p=Fp, x_dtype=L.SR_F32, y_dtype=sr_dtype(yc.dtype))
                    if prev is None:
                        ops.rcab_conv_pair(x=r.data_ptr(), **kw)
                    else:
                        pr, py, pp, (w1, b1, w2, b2) = prev
                        nxt = ra if (pr is not ra) else rb
                        ops.rcab_conv_pair(x=pr.data_ptr(), gate_y=py.data_ptr(), gate_pool=pp.data_ptr(), gate_w1=w1.data_ptr(), gate_b1=b1.data_ptr(),
                                           gate_w2=w2.data_ptr(), gate_b2=b2.data_ptr(), x_out=nxt.data_ptr(), gate_C=self.n_feats, gate_Cr=w1.shape[0], **kw)
                        r = nxt
                    prev = (r, yc, pc, ca)
                if prev is not None:  # the last block's tail: standalone channel attention
                    pr, py, pp, ca = prev
                    nxt = ra if (pr is not ra) else rb
                    run_channel_attention(ca, py, pp, n_tiles, self.n_feats, nxt, skip=pr)
                    r = nxt
            else:
                for (c1, c2, ca) in blocks:  # RCAB: r = CA(conv2(relu(conv1(r)))) + r
                    if fused_pair:
                        ops.rcab_conv_pair(x=r.data_ptr(), w1p=c1[0].data_ptr(), b1=c1[1].data_ptr(), w2p=c2[0].data_ptr(), b2=c2[1].data_ptr(),
                                           y=y.data_ptr(), pool_partial=pool.data_ptr(), B=B, H=H, W=W, C_p=Fp, x_dtype=sr_dtype(r.dtype),
                                           y_dtype=sr_dtype(y.dtype))
                    else:
                        conv_call(r, *c1, mid, cdt, act=L.ACT_RELU)
                        conv_call(mid, *c2, y, cdt, pool=pool)
                    nxt = ra if (r is not ra) else rb
                    run_channel_attention(ca, y, pool, n_tiles, self.n_feats, nxt, skip=r)
                    r = nxt
            gn = ws_.get(f"feat{gi}", (B, H, W, Fp), f32) if keep_all else (ga if (g is not ga) else gb)
            conv_call(r, *gconv, gn, cdt, skip=g)  # group conv + skip (rcan.py:33-36)
            g = gn
            feats.append(gn)
        return g, feats

    def forward(self, x: Tensor) -> Tensor:
        y = self._train_forward(x)
        if y is not None:
            return y
        x = self._check_input(x)
        cdt = compute_dtype(self.precision)
        P = self._get_packed(cdt)
        ws_ = self._workspace(x.device)
        B, _, H, W = x.shape
        s = self.scale
        out = torch.empty(B, self.n_colors, H * s, W * s, dtype=torch.float32, device=x.device)
        # part batches on several streams pay off only when the launches cost no CPU time, i.e. inside a HIP-graph capture: an eager forward
        # is launch-bound at 4 x 222 launches (b16 eager: 12.5 ms with four parts, 7.2 with two, 6.4 with one)
        if self.pipeline_halves and B >= 16 and B % 2 == 0 and (capturing_or_warming_up() or os.environ.get("SR_RCAN_PARTS")):
            # Every RCAB launch is whole residency rounds whose load, MFMA and store phases run one after the other chip-wide; two half
            # batches on two streams are out of phase, so one's HBM phases run under the other's MFMAs.  Measured (tools/rcan_ab.py,
            # same box): b8 4.81 -> 5.58 ms (worse: off below 16), b16 6.43 -> 6.04 ms, b32 12.08 -> 9.13 ms.
            from ..runtime import WorkspaceView

            main = torch.cuda.current_stream(x.device)
            # parts (SR_RCAN_PARTS): b16 5.89 (2) / 5.70 (4) / 9.63 ms (8: the captured graph's cross-queue signalling takes over); b32 8.50 / 8.26 / 11.86
            # with the XCD-aware tile order of sr_rcab (round 4): parts of EIGHT images -- b16 4.86 (4 parts) -> 4.73 ms (2), b32 7.37 (2) / 7.26 (4)
            parts = int(os.environ.get("SR_RCAN_PARTS", "0")) or min(4, max(2, B // 8))  # (b64: 4 parts 14.0 ms, 8 parts 14.4; b32: 4 parts 7.27, 8 parts 9.5)
            if parts < 2 or B % parts:
                parts = 2
            h = B // parts
            sides = [self._side_stream(x.device, i) for i in range(parts - 1)]
            for i, side in enumerate(sides):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._forward_into(P, x[(i + 1) * h:(i + 2) * h], out[(i + 1) * h:(i + 2) * h], WorkspaceView(ws_, f"h{i + 1}."), cdt)
            self._forward_into(P, x[:h], out[:h], WorkspaceView(ws_, "h0."), cdt)
            for side in sides:
                main.wait_stream(side)
        else:
            self._forward_into(P, x, out, ws_, cdt)
        return out

    pipeline_halves = True

    def _side_stream(self, device, i: int = 0) -> "torch.cuda.Stream":
        sts = getattr(self, "_side", None)
        if not isinstance(sts, dict) or sts.get("device") != torch.device(device):
            sts = {"device": torch.device(device)}
            object.__setattr__(self, "_side", sts)
        if i not in sts:
            sts[i] = torch.cuda.Stream(device=device)
        return sts[i]

    def _forward_into(self, P: Dict, x: Tensor, out: Tensor, ws_, cdt) -> None:
        B, _, H, W = x.shape
        Fp = P["Fp"]
        f32 = torch.float32
        xin = ws_.get("xin", (B, H, W, 32), cdt)
        ops.ingest_nchw(x, xin, L.PAD_NONE, *P["ing"])
        h = ws_.get("head", (B, H, W, Fp), f32)
        conv_call(xin, *P["head"], h, cdt)
        g, _ = self._run_groups(P, h, ws_, cdt)
        res = ws_.get("res", (B, H, W, Fp), cdt)
        conv_call(g, *P["body_last"], res, cdt, skip=h)
        up = run_upsampler(P["up"], res, ws_, cdt, "rcan")
        s = self.scale
        conv_call(up, *P["tail"], out, cdt, out_mode=L.OUT_FINAL_NCHW, fin=(*P["fin"], self.n_colors, H * s, W * s), cout_p=16)

    def get_model_config(self) -> Dict:
        config = super().get_model_config()
        config.update(dict(scale=self.scale, n_colors=self.n_colors, img_range=self.img_range, n_feats=self.n_feats,
                           n_resblocks=self.n_resblocks, n_resgroups=self.n_resgroups, reduction=self.reduction))
        return config

    def get_training_config(self) -> Dict:  # rcan.py:93-104
        return dict(batch_size=16, learning_rate=0.0001, beta1=0.9, beta2=0.99, weight_decay=0.0, max_iters=1000000, gamma=0.5,
                    milestones=[200000, 400000, 600000, 800000])

    @classmethod
    def from_pretrained(cls, scale: int = 4) -> "RCAN":
        """rcan.py:107-119: RCAN_BIX{scale}.pt with img_range 255, read from ./pretrained (no network here)."""
        path = os.path.join("pretrained", "models_ECCV2018RCAN", f"RCAN_BIX{scale}.pt")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found (no network access here; place the official checkpoint there)")
        model = cls(scale=scale, img_range=255.0)
        model.load_state_dict(torch.load(path, map_location="cpu"), False)
        return model
