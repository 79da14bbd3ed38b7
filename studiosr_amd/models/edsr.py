"""EDSR on the MI355X HIP hot path (reference: studiosr/models/edsr.py).

forward = ingest (MeanShift sub_mean folded into the NCHW->NHWC ingest, edsr.py:40) -> head conv (:41)
-> n_resblocks x [conv+ReLU, conv*res_scale + skip] (common.py:150-153) -> conv + long skip (:43-44)
-> Upsampler convs storing through PixelShuffle (:46) -> tail conv + add_mean + NCHW store (:46-47).
Every FLOP is the implicit-GEMM 3x3 kernel (csrc/sr_conv.hip).
"""
from __future__ import annotations

import os
from typing import Dict

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops, packing
from ..runtime import compute_dtype
from .common import RGB_MEAN, Model, Upsampler, conv2d, conv_call, pack_upsampler, run_upsampler

Tensor = torch.Tensor


class MeanShift(nn.Conv2d):
    """Frozen 1x1 conv of the reference (common.py:108-121); kept for state_dict compatibility, folded into
    the ingest / final-store affine at run time."""

    def __init__(self, img_range: float, rgb_mean=RGB_MEAN, rgb_std=(1.0, 1.0, 1.0), sign: int = -1) -> None:
        super().__init__(3, 3, kernel_size=1)
        std = torch.tensor(rgb_std)
        self.weight.data = torch.eye(3).view(3, 3, 1, 1) / std.view(3, 1, 1, 1)
        self.bias.data = sign * img_range * torch.tensor(rgb_mean) / std
        for p in self.parameters():
            p.requires_grad = False


class ResBlock(nn.Module):
    """Parameters of common.py:140-148 (keys body.0 / body.2)."""

    def __init__(self, n_feats: int, kernel_size: int, res_scale: float = 1.0) -> None:
        super().__init__()
        self.body = nn.Sequential(conv2d(n_feats, n_feats, kernel_size), nn.ReLU(True), conv2d(n_feats, n_feats, kernel_size))
        self.res_scale = res_scale


def mean_shift_affine(ms: MeanShift):
    """(scale[c], bias[c]) of a MeanShift whose weight is diagonal (it always is: eye/std)."""
    w = ms.weight.detach().reshape(3, 3)
    return torch.diagonal(w).to(torch.float32).contiguous(), ms.bias.detach().to(torch.float32).contiguous()


class EDSR(Model):
    def __init__(self, scale: int = 4, n_colors: int = 3, img_range: float = 1.0, n_feats: int = 256, n_resblocks: int = 32,
                 res_scale: float = 0.1) -> None:
        super().__init__(scale, n_colors, img_range)
        self.n_feats = n_feats
        self.n_resblocks = n_resblocks
        self.res_scale = res_scale
        self.sub_mean = MeanShift(img_range)
        self.add_mean = MeanShift(img_range, sign=1)
        k = 3
        self.head = nn.Sequential(conv2d(n_colors, n_feats, k))
        self.body = nn.Sequential(*[ResBlock(n_feats, k, res_scale) for _ in range(n_resblocks)], conv2d(n_feats, n_feats, k))
        self.tail = nn.Sequential(Upsampler(scale, n_feats), conv2d(n_feats, n_colors, k))

    def _pack(self, dt: torch.dtype) -> Dict:
        F = self.n_feats
        Fp = packing.round_up(F, 32)
        ident = packing.identity_idx(F, Fp)
        P: Dict = {"Fp": Fp}
        P["ing"] = mean_shift_affine(self.sub_mean)
        P["fin"] = mean_shift_affine(self.add_mean)
        P["head"] = packing.pack_conv3x3(self.head[0].weight, self.head[0].bias, 32, ident, dt)
        P["blocks"] = []
        for i in range(self.n_resblocks):
            rb = self.body[i]
            P["blocks"].append(
                (packing.pack_conv3x3(rb.body[0].weight, rb.body[0].bias, Fp, ident, dt), packing.pack_conv3x3(rb.body[2].weight, rb.body[2].bias, Fp, ident, dt))
            )
        bl = self.body[self.n_resblocks]
        P["body_last"] = packing.pack_conv3x3(bl.weight, bl.bias, Fp, ident, dt)
        P["up"] = pack_upsampler(self.tail[0], Fp, dt)
        P["tail"] = packing.pack_conv3x3(self.tail[1].weight, self.tail[1].bias, P["up"][-1][3], packing.identity_idx(self.n_colors, 16), dt)
        return P

    def forward(self, x: Tensor) -> Tensor:
        y = self._train_forward(x)
        if y is not None:
            return y
        x = self._check_input(x)
        cdt = compute_dtype(self.precision)
        P = self._get_packed(cdt)
        ws_ = self._workspace(x.device)
        B, _, H, W = x.shape
        Fp = P["Fp"]
        xin = ws_.get("xin", (B, H, W, 32), cdt)
        ops.ingest_nchw(x, xin, L.PAD_NONE, *P["ing"])
        h = ws_.get("head", (B, H, W, Fp), torch.float32)
        conv_call(xin, *P["head"], h, cdt)
        ra = ws_.get("ra", (B, H, W, Fp), torch.float32)
        rb = ws_.get("rb", (B, H, W, Fp), torch.float32)
        mid = ws_.get("mid", (B, H, W, Fp), cdt)
        cur = h
        for i, (c1, c2) in enumerate(P["blocks"]):  # res = conv2(relu(conv1(x))) * res_scale + x
            conv_call(cur, *c1, mid, cdt, act=L.ACT_RELU)
            nxt = ra if (cur is not ra) else rb
            conv_call(mid, *c2, nxt, cdt, out_scale=self.res_scale, skip=cur)
            cur = nxt
        res = ws_.get("res", (B, H, W, Fp), cdt)
        conv_call(cur, *P["body_last"], res, cdt, skip=h)  # body(x) + x  (edsr.py:43-44)
        up = run_upsampler(P["up"], res, ws_, cdt, "edsr")
        s = self.scale
        out = torch.empty(B, self.n_colors, H * s, W * s, dtype=torch.float32, device=x.device)
        conv_call(up, *P["tail"], out, cdt, out_mode=L.OUT_FINAL_NCHW, fin=(*P["fin"], self.n_colors, H * s, W * s), cout_p=16)
        return out

    def get_model_config(self) -> Dict:
        config = super().get_model_config()
        config.update(dict(scale=self.scale, n_colors=self.n_colors, img_range=self.img_range, n_feats=self.n_feats,
                           n_resblocks=self.n_resblocks, res_scale=self.res_scale))
        return config

    def get_training_config(self) -> Dict:  # edsr.py:64-75
        return dict(batch_size=16, learning_rate=0.0001, beta1=0.9, beta2=0.99, weight_decay=0.0, max_iters=1000000, gamma=0.5,
                    milestones=[200000, 400000, 600000, 800000])

    @classmethod
    def from_pretrained(cls, scale: int = 4, dataset: str = "DIV2K") -> "EDSR":
        """Checkpoint names of edsr.py:77-112, read from ./pretrained (no network here)."""
        assert scale in [2, 3, 4]
        assert dataset in ["DIV2K", "DF2K"]
        if dataset == "DIV2K":
            model, file_name = cls(scale=scale, img_range=255.0), f"r32f256x{scale}.pth"
        else:
            model, file_name = cls(scale=scale), f"EDSRx{scale}.pth"
        path = os.path.join("pretrained", file_name)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found (no network access here; place the official checkpoint there)")
        model.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
        return model
