"""HAN on the HIP path (reference: studiosr/models/han.py): RCAN's residual groups plus a layer-attention module over the 11 group
outputs (LAM, han.py:12-33) and a channel-spatial attention module (CSAM: a 3x3x3 Conv3d over the (C, H, W) volume, han.py:36-53).

Same constructor kwargs / state_dict keys as the reference.  The forward runs on the generic fp32 engine (studiosr_amd/autograd.py) in
eval and train mode and is differentiable end to end.
"""
from __future__ import annotations

import os
from typing import Dict

import torch
import torch.nn as nn

from .common import Model, Upsampler, conv2d
from .edsr import MeanShift
from .rcan import ResidualGroup


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

    def forward(self, x):
        from . import train

        return train.han_forward(self, self._check_input(x))

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
