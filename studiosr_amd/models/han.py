"""HAN on the HIP path (reference: studiosr/models/han.py): RCAN's residual groups plus a layer-attention module over the 11 group
outputs (LAM, han.py:12-33) and a channel-spatial attention module (CSAM: a 3x3x3 Conv3d over the (C, H, W) volume, han.py:36-53).

Same constructor kwargs / state_dict keys as the reference.  Inference runs RCAN's launch sequence for the head and the residual groups
(one fused launch per RCAB in bf16) and for the upsampler / tail, and the generic engine (studiosr_amd/autograd.py) for the attention
tail between them (LAM, CSAM, the two fusing convs); whenever autograd is recording the whole model runs on that engine and is
differentiable end to end.
"""
from __future__ import annotations

import os
from typing import Dict

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..runtime import compute_dtype
from .common import Model, Upsampler, conv2d, conv_call, run_upsampler
from .edsr import MeanShift
from .rcan import RCAN, ResidualGroup


class LAM_Module(nn.Module):
    def __init__(self, in_dim: int) -> None:
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1))


class CSAM_Module(nn.Module):
    def __init__(self, in_dim: int) -> None:
        super().__init__()
        self.conv = nn.Conv3d(1, 1, 3, 1, 1)
        self.gamma = nn.Parameter(torch.zeros(1))


class HAN(Model):
    def __init__(self, scale: int = 4, n_colors: int = 3, img_range: float = 1.0, n_feats: int = 64, n_resblocks: int = 20, n_resgroups: int = 10,
                 reduction: int = 16) -> None:
        super().__init__(scale, n_colors, img_range)
        self.n_feats, self.n_resblocks, self.n_resgroups, self.reduction = n_feats, n_resblocks, n_resgroups, reduction
        self.sub_mean = MeanShift(img_range)
        self.add_mean = MeanShift(img_range, sign=1)
        k = 3
        self.head = nn.Sequential(conv2d(n_colors, n_feats, k))
        self.body = nn.Sequential(*[ResidualGroup(n_feats, k, reduction, n_resblocks) for _ in range(n_resgroups)], conv2d(n_feats, n_feats, k))
        self.tail = nn.Sequential(Upsampler(scale, n_feats), conv2d(n_feats, n_colors, k))
        self.csa = CSAM_Module(n_feats)
        self.la = LAM_Module(n_feats)
        self.last_conv = nn.Conv2d(n_feats * (n_resgroups + 1), n_feats, 3, 1, 1)  # han.py:87 hard-codes 11 = the default n_resgroups + 1
        self.last = nn.Conv2d(n_feats * 2, n_feats, 3, 1, 1)

    _pack = RCAN._pack            # head / residual groups / body conv / upsampler / tail are RCAN's modules under the same names
    _run_groups = RCAN._run_groups

    def forward(self, x):
        y = self._train_forward(x)
        if y is not None:
            return y
        from .. import autograd as A
        from . import train

        x = self._check_input(x)
        cdt = compute_dtype(self.precision)
        P = self._get_packed(cdt)
        ws_ = self._workspace(x.device)
        B, _, H, W = x.shape
        Fp, F, f32 = P["Fp"], self.n_feats, torch.float32
        xin = ws_.get("xin", (B, H, W, 32), cdt)
        ops.ingest_nchw(x, xin, L.PAD_NONE, *P["ing"])
        h = ws_.get("head", (B, H, W, Fp), f32)
        conv_call(xin, *P["head"], h, cdt)
        g, feats = self._run_groups(P, h, ws_, cdt, keep_all=True)
        last = ws_.get(f"feat{len(feats)}", (B, H, W, Fp), f32)
        conv_call(g, *P["body_last"], last, cdt)  # han.py:99: no long skip here, it follows the attention tail
        unpad = (lambda t: t) if Fp == F else (lambda t: t[..., :F].contiguous())
        with torch.no_grad(), A.autocast_state(cdt == torch.bfloat16):  # LAM, CSAM, last_conv, last (han.py:96-113)
            r = train.han_attention(self, [unpad(t) for t in feats + [last]], unpad(h))
        res = ws_.get("res", (B, H, W, Fp), cdt)
        res[..., :F] = r
        if Fp != F:
            res[..., F:] = 0
        up = run_upsampler(P["up"], res, ws_, cdt, "han")
        s = self.scale
        out = torch.empty(B, self.n_colors, H * s, W * s, dtype=f32, device=x.device)
        conv_call(up, *P["tail"], out, cdt, out_mode=L.OUT_FINAL_NCHW, fin=(*P["fin"], self.n_colors, H * s, W * s), cout_p=16)
        return out

    def get_model_config(self) -> Dict:
        config = super().get_model_config()
        config.update(dict(n_feats=self.n_feats, n_resblocks=self.n_resblocks, n_resgroups=self.n_resgroups, reduction=self.reduction))
        return config

    def get_training_config(self) -> Dict:  # han.py:129-140
        return dict(batch_size=16, learning_rate=0.0001, beta1=0.9, beta2=0.99, weight_decay=0.0, max_iters=1000000, gamma=0.5,
                    milestones=[200000, 400000, 600000, 800000])

    @classmethod
    def from_pretrained(cls, scale: int = 4) -> "HAN":
        """han.py:142-161: HAN_BIX{scale}.pt with img_range 255; read from ./pretrained (no network here)."""
        model = cls(scale=scale, img_range=255.0)
        path = os.path.join("pretrained", f"HAN_BIX{scale}.pt")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found (no network access here; place the official checkpoint there)")
        model.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
        return model
