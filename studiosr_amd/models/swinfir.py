"""SwinFIR on the HIP path (reference: studiosr/models/swinfir.py): SwinIR whose RSTB convs and conv_after_body are SFB blocks --
a spatial conv branch plus a spectral branch (1x1 conv -> rfftn -> 1x1 conv on (real | imag) -> irfftn -> 1x1 conv), fused by a 1x1 conv.

Same constructor kwargs and state_dict keys as the reference (`layers.i.conv.S.body.0.weight`, `...F.fu.conv_layer.weight`, ...).
Inference is SwinIR's launch sequence -- the 36 Swin blocks are the one-launch block kernel (bf16) or the exact-fp32 GEMM / attention
kernels -- with the 7 SFBs evaluated on the generic engine (studiosr_amd/autograd.py: the 2-D real FFT is two DFT matrix products on
the fp32 matrix cores) between them; whenever autograd is recording the whole model runs on that engine and is differentiable end to end.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .. import _lib as L
from .. import packing
from .common import conv_call
from .swinir import SwinIR


class FourierUnit(nn.Module):
    def __init__(self, embed_dim: int) -> None:
        super().__init__()
        self.conv_layer = nn.Conv2d(embed_dim * 2, embed_dim * 2, 1, 1, 0)


class SpectralTransform(nn.Module):
    def __init__(self, embed_dim: int) -> None:
        super().__init__()
        self.conv_before_fft = nn.Sequential(nn.Conv2d(embed_dim, embed_dim // 2, 1, 1, 0), nn.LeakyReLU(0.2, inplace=True))
        self.fu = FourierUnit(embed_dim // 2)
        self.conv_after_fft = nn.Conv2d(embed_dim // 2, embed_dim, 1, 1, 0)


class SpatialB(nn.Module):
    def __init__(self, embed_dim: int, red: int = 1) -> None:
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(embed_dim, embed_dim // red, 3, 1, 1), nn.LeakyReLU(0.2, inplace=True), nn.Conv2d(embed_dim // red, embed_dim, 3, 1, 1))


class SFB(nn.Module):
    """Parameters of swinfir.py:68-81."""

    def __init__(self, embed_dim: int, red: int = 1) -> None:
        super().__init__()
        self.S = SpatialB(embed_dim, red)
        self.F = SpectralTransform(embed_dim)
        self.fusion = nn.Conv2d(embed_dim * 2, embed_dim, 1, 1, 0)


class SwinFIR(SwinIR):
    def __init__(self, scale: int = 4, n_colors: int = 3, img_range: float = 1.0, embed_dim: int = 180, depths: List[int] = [6, 6, 6, 6, 6, 6],
                 num_heads: List[int] = [6, 6, 6, 6, 6, 6], window_size: int = 8, mlp_ratio: float = 2.0, drop_rate: float = 0.0, attn_drop_rate: float = 0.0,
                 drop_path_rate: float = 0.1, upsampler: str = "pixelshuffle") -> None:
        super().__init__(scale=scale, n_colors=n_colors, img_range=img_range, embed_dim=embed_dim, depths=depths, num_heads=num_heads, window_size=window_size,
                         mlp_ratio=mlp_ratio, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate, upsampler=upsampler)
        for layer in self.layers:  # RSTB(resi_connection=SFB) (swinfir.py:112, swinir.py:241)
            layer.conv = SFB(embed_dim)
        self.conv_after_body = SFB(embed_dim)  # swinfir.py:114

    def _pack_resi(self, m, C, Cp, dt):
        """The spatial branch's two 3x3 convs run on the fused conv kernels (packed here); the spectral branch and the fusing 1x1 conv read
        the SFB's own parameters in place on the generic engine."""
        ident = packing.identity_idx(C, Cp)
        body = m.S.body
        return dict(m=m, c1=packing.pack_conv3x3(body[0].weight, body[0].bias, Cp, ident, dt), c2=packing.pack_conv3x3(body[2].weight, body[2].bias, Cp, ident, dt))

    def _run_resi(self, pk, src, dst, skip, cdt) -> None:
        """dst = SFB(src) + skip on the padded NHWC buffers of SwinIR.forward (swinfir.py:68-81, swinir.py:245-246,362)."""
        from .. import autograd as A
        from . import train

        m, C = pk["m"], self.embed_dim
        ws_ = self._ws
        # S(x) = conv(LeakyReLU_0.2(conv(x))) + x  (swinfir.py:55-65) on the fused kernels, padded layout
        mid = ws_.get("sfb.mid", tuple(src.shape), cdt)
        sp = ws_.get("sfb.s", tuple(src.shape), torch.float32)
        conv_call(src, *pk["c1"], mid, cdt, act=L.ACT_LRELU, act_slope=0.2)
        conv_call(mid, *pk["c2"], sp, cdt, skip=src)
        x = src[..., :C].float().contiguous()
        s_ = sp[..., :C].contiguous()
        with torch.no_grad(), A.autocast_state(cdt == torch.bfloat16):  # bf16 precision: bf16-operand contractions, as the Swin blocks
            y = train._sfb_tail(m, x, s_)
        if dst is skip:
            dst[..., :C] += y
        else:
            dst[..., :C] = y + skip[..., :C]
            dst[..., C:] = 0  # pad lanes stay zero end to end

    def forward_strips(self, x, comm):
        raise NotImplementedError("SwinFIR's spectral branch is a whole-image FFT: row strips with halo exchange do not apply")

    def get_training_config(self) -> Dict:  # swinfir.py:116-128
        return dict(batch_size=32, learning_rate=0.0002, beta1=0.9, beta2=0.99, weight_decay=0.0, max_iters=500000, gamma=0.5,
                    milestones=[250000, 400000, 450000, 475000], bfloat16=False)

    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        raise NotImplementedError("the reference ships no SwinFIR checkpoints (swinfir.py has no from_pretrained of its own)")
