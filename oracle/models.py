"""Oracle model forwards (TEST INFRASTRUCTURE, see oracle/__init__.py).

Each ``*_forward(sd, x, cfg, training)`` is a pure function of a ``state_dict``
whose keys/shapes are exactly the reference's (so a reference ``state_dict`` and
a ``studiosr_amd`` ``state_dict`` both plug in), an NCHW fp32 input and the dict
the reference's ``get_model_config()`` returns.  Eval-mode semantics unless
``training=True`` (which only changes SwinIR's padding; DropPath is treated as
identity, i.e. ``drop_path_rate == 0`` or eval).
"""
from __future__ import annotations

from typing import Callable, Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import functional as OF

Tensor = torch.Tensor
SD = Dict[str, Tensor]

RGB_MEAN = (0.4488, 0.4371, 0.4040)  # studiosr/models/common.py:112,223


def _mean(x: Tensor) -> Tensor:
    return torch.tensor(RGB_MEAN, dtype=x.dtype).reshape(1, 3, 1, 1)


# --------------------------------------------------------------------------- #
# SwinIR  (studiosr/models/swinir.py)
# --------------------------------------------------------------------------- #
def swin_block(sd: SD, key: str, x: Tensor, ws: int, shift: int, heads: int) -> Tensor:
    """SwinTransformerBlock.forward on [B,H,W,C] (studiosr/models/swinir.py:146-174).
    The mask is added for shift == 0 too (all zeros) exactly as the reference does."""
    _, h, w, _ = x.shape
    mask = OF.calculate_mask(h, w, ws, shift)
    rpi = sd[key + ".attn.relative_position_index"]
    y = OF.shifted_window_msa(sd, key + ".attn", OF.layer_norm(sd, key + ".norm1", x), ws, shift, rpi, heads, mask)
    x = x + y
    return x + OF.mlp(sd, key + ".mlp", OF.layer_norm(sd, key + ".norm2", x))


def sfb(sd: SD, key: str, x: Tensor) -> Tensor:
    """SwinFIR's SFB on NCHW (studiosr/models/swinfir.py:9-81): S = conv-LeakyReLU(0.2)-conv + x; F = conv_after_fft(FourierUnit(y) + y)
    with y = LeakyReLU(conv1x1(x)) and FourierUnit = irfftn(split(LeakyReLU(conv1x1(cat(rfftn(y).real, .imag))))), norm "ortho"."""
    s = OF.conv(sd, key + ".S.body.2", F.leaky_relu(OF.conv(sd, key + ".S.body.0", x), 0.2)) + x
    y = F.leaky_relu(OF.conv(sd, key + ".F.conv_before_fft.0", x), 0.2)
    f = torch.fft.rfftn(y, dim=(-2, -1), norm="ortho")
    z = F.leaky_relu(OF.conv(sd, key + ".F.fu.conv_layer", torch.cat((f.real, f.imag), dim=1)), 0.2)
    re, im = z.split(y.shape[1], dim=1)
    fu = torch.fft.irfftn(torch.complex(re, im), s=y.shape[-2:], dim=(-2, -1), norm="ortho")
    f_out = OF.conv(sd, key + ".F.conv_after_fft", fu + y)
    return OF.conv(sd, key + ".fusion", torch.cat([s, f_out], dim=1))


def _resi(sd: SD, key: str, x: Tensor) -> Tensor:
    """RSTB.conv / conv_after_body: nn.Conv2d for SwinIR, SFB for SwinFIR (swinir.py:241, swinfir.py:112-114)."""
    return OF.conv(sd, key, x) if key + ".weight" in sd else sfb(sd, key, x)


def swinir_forward(sd: SD, x: Tensor, cfg: Dict, training: bool = False) -> Tensor:
    """SwinIR.forward (studiosr/models/swinir.py:342-372)."""
    ws, scale, rng = cfg["window_size"], cfg["scale"], cfg["img_range"]
    h0, w0 = x.shape[2:]
    x = OF.pad_reflect(x, ws) if training else OF.pad_eval(x, ws)  # :356-357
    x = x / rng - _mean(x)  # common.py:228-230
    first = OF.conv(sd, "conv_first", x)
    t = OF.layer_norm(sd, "patch_embed.norm", first.permute(0, 2, 3, 1))  # :28-32
    for li, depth in enumerate(cfg["depths"]):
        tin = t
        for bi in range(depth):
            shift = 0 if bi % 2 == 0 else ws // 2  # :200
            t = swin_block(sd, f"layers.{li}.residual_group.blocks.{bi}", t, ws, shift, cfg["num_heads"][li])
        t = _resi(sd, f"layers.{li}.conv", t.permute(0, 3, 1, 2)).permute(0, 2, 3, 1) + tin  # :245-246
    t = OF.layer_norm(sd, "norm", t).permute(0, 3, 1, 2)
    x = _resi(sd, "conv_after_body", t) + first  # :362
    if cfg.get("upsampler", "pixelshuffle") == "pixelshuffle":
        x = F.leaky_relu(OF.conv(sd, "conv_before_upsample.0", x), 0.01)
        x = OF.conv(sd, "conv_last", OF.upsampler(sd, "upsample", x, scale))
    else:  # pixelshuffledirect :367-369
        x = OF.upsampler(sd, "upsample", x, scale, direct=True)
    x = (x + _mean(x)) * rng  # common.py:232-233
    return x[:, :, : h0 * scale, : w0 * scale]


# --------------------------------------------------------------------------- #
# EDSR / RCAN  (studiosr/models/edsr.py, rcan.py)
# --------------------------------------------------------------------------- #
def _mean_shift(x: Tensor, rng: float, sign: int) -> Tensor:
    """MeanShift with std 1: identity 1x1 conv + sign*range*mean bias
    (studiosr/models/common.py:108-121)."""
    return x + sign * rng * _mean(x)


def edsr_forward(sd: SD, x: Tensor, cfg: Dict, training: bool = False) -> Tensor:
    """EDSR.forward (studiosr/models/edsr.py:39-48); ResBlock common.py:150-153."""
    nb = cfg["n_resblocks"]
    x = OF.conv(sd, "head.0", _mean_shift(x, cfg["img_range"], -1))
    r = x
    for i in range(nb):
        y = OF.conv(sd, f"body.{i}.body.2", F.relu(OF.conv(sd, f"body.{i}.body.0", r)))
        r = y * cfg["res_scale"] + r
    r = OF.conv(sd, f"body.{nb}", r) + x
    y = OF.conv(sd, "tail.1", OF.upsampler(sd, "tail.0", r, cfg["scale"]))
    return _mean_shift(y, cfg["img_range"], +1)


def rcan_forward(sd: SD, x: Tensor, cfg: Dict, training: bool = False) -> Tensor:
    """RCAN.forward (studiosr/models/rcan.py:68-77); RCAB :21-24; group :33-36."""
    ng, nb = cfg["n_resgroups"], cfg["n_resblocks"]
    x = OF.conv(sd, "head.0", _mean_shift(x, cfg["img_range"], -1))
    g = x
    for gi in range(ng):
        r = g
        for bi in range(nb):
            k = f"body.{gi}.body.{bi}.body"
            y = OF.conv(sd, k + ".2", F.relu(OF.conv(sd, k + ".0", r)))
            y = OF.channel_attention(
                y,
                sd[k + ".3.conv_du.0.weight"],
                sd[k + ".3.conv_du.0.bias"],
                sd[k + ".3.conv_du.2.weight"],
                sd[k + ".3.conv_du.2.bias"],
            )
            r = y + r
        g = OF.conv(sd, f"body.{gi}.body.{nb}", r) + g
    r = OF.conv(sd, f"body.{ng}", g) + x
    y = OF.conv(sd, "tail.1", OF.upsampler(sd, "tail.0", r, cfg["scale"]))
    return _mean_shift(y, cfg["img_range"], +1)


# --------------------------------------------------------------------------- #
# HAT  (studiosr/models/hat.py)
# --------------------------------------------------------------------------- #
def hat_cab(sd: SD, key: str, x: Tensor) -> Tensor:
    """CAB: conv -> GELU -> conv -> channel attention (studiosr/models/hat.py:41-52)."""
    y = OF.conv(sd, key + ".cab.2", F.gelu(OF.conv(sd, key + ".cab.0", x)))
    a = key + ".cab.3.attention"
    return OF.channel_attention(y, sd[a + ".1.weight"], sd[a + ".1.bias"], sd[a + ".3.weight"], sd[a + ".3.bias"])


def hat_hab(sd: SD, key: str, x: Tensor, ws: int, shift: int, heads: int, rpi: Tensor, mask, conv_scale: float):
    """HAB.forward on [B,H,W,C] (studiosr/models/hat.py:153-195); mask only when shifted (:174)."""
    n = OF.layer_norm(sd, key + ".norm1", x)
    conv_x = hat_cab(sd, key + ".conv_block", n.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    a = OF.shifted_window_msa(sd, key + ".attn", n, ws, shift, rpi, heads, mask if shift > 0 else None)
    x = x + a + conv_x * conv_scale  # :192
    return x + OF.mlp(sd, key + ".mlp", OF.layer_norm(sd, key + ".norm2", x))


def hat_ocab(sd: SD, key: str, x: Tensor, ws: int, heads: int, rpi: Tensor, overlap_ratio: float) -> Tensor:
    """OCAB.forward on [B,H,W,C] (studiosr/models/hat.py:239-293).  k/v windows are the
    (ws+ov)x(ws+ov) neighbourhoods of each ws x ws query window, zero padded at the
    image border (nn.Unfold(kernel=wse, stride=ws, padding=(wse-ws)//2), :217-221)."""
    b, h, w, c = x.shape
    wse = int(ws * overlap_ratio) + ws
    pad = (wse - ws) // 2
    hd = c // heads
    qkv = OF.linear(sd, key + ".qkv", OF.layer_norm(sd, key + ".norm1", x)).reshape(b, h, w, 3, c)
    q = OF.window_partition(qkv[:, :, :, 0], ws).reshape(-1, ws * ws, heads, hd).permute(0, 2, 1, 3)
    kv = []
    for part in (1, 2):
        t = F.pad(qkv[:, :, :, part], (0, 0, pad, pad, pad, pad))  # zero pad H and W
        wins = []
        for wy in range(h // ws):
            for wx in range(w // ws):
                wins.append(t[:, wy * ws : wy * ws + wse, wx * ws : wx * ws + wse])
        t = torch.stack(wins, 1).reshape(-1, wse * wse, heads, hd).permute(0, 2, 1, 3)
        kv.append(t)
    k, v = kv
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    bias = sd[key + ".relative_position_bias_table"][rpi.reshape(-1)]  # negative idx wrap (python semantics)
    attn = attn + bias.reshape(ws * ws, wse * wse, heads).permute(2, 0, 1)[None]
    out = (torch.softmax(attn, -1) @ v).transpose(1, 2).reshape(-1, ws, ws, c)
    x = OF.linear(sd, key + ".proj", OF.window_reverse(out, ws, h, w)) + x
    return x + OF.mlp(sd, key + ".mlp", OF.layer_norm(sd, key + ".norm2", x))


def hat_forward(sd: SD, x: Tensor, cfg: Dict, training: bool = False) -> Tensor:
    """HAT.forward / forward_features (studiosr/models/hat.py:519-554)."""
    ws, scale, rng = cfg["window_size"], cfg["scale"], cfg["img_range"]
    h0, w0 = x.shape[2:]
    x = OF.pad_reflect(x, ws)  # :544 always reflect
    x = x / rng - _mean(x)
    first = OF.conv(sd, "conv_first", x)
    hp, wp = first.shape[2:]
    mask = OF.calculate_mask(hp, wp, ws, ws // 2)  # :524
    rpi_sa, rpi_oca = sd["relative_position_index_SA"], sd["relative_position_index_OCA"]
    t = OF.layer_norm(sd, "patch_embed.norm", first.permute(0, 2, 3, 1))
    for li, depth in enumerate(cfg["depths"]):
        tin = t
        rg = f"layers.{li}.residual_group"
        for bi in range(depth):
            shift = 0 if bi % 2 == 0 else ws // 2
            t = hat_hab(sd, f"{rg}.blocks.{bi}", t, ws, shift, cfg["num_heads"][li], rpi_sa, mask, cfg["conv_scale"])
        t = hat_ocab(sd, f"{rg}.overlap_attn", t, ws, cfg["num_heads"][li], rpi_oca, cfg["overlap_ratio"])
        t = OF.conv(sd, f"layers.{li}.conv", t.permute(0, 3, 1, 2)).permute(0, 2, 3, 1) + tin  # :385
    t = OF.layer_norm(sd, "norm", t).permute(0, 3, 1, 2)
    x = OF.conv(sd, "conv_after_body", t) + first
    x = F.leaky_relu(OF.conv(sd, "conv_before_upsample.0", x), 0.01)
    x = OF.conv(sd, "conv_last", OF.upsampler(sd, "upsample", x, scale))
    x = (x + _mean(x)) * rng
    return x[:, :, : h0 * scale, : w0 * scale]


def han_forward(sd: SD, x: Tensor, cfg: Dict, training: bool = False) -> Tensor:
    """HAN.forward (studiosr/models/han.py:92-115) with LAM (:19-33) and CSAM (:44-53)."""
    ng, nb = cfg["n_resgroups"], cfg["n_resblocks"]
    x = OF.conv(sd, "head.0", _mean_shift(x, cfg["img_range"], -1))
    res, feats = x, []
    for gi in range(ng):
        r = res
        for bi in range(nb):
            k = f"body.{gi}.body.{bi}.body"
            y = OF.conv(sd, k + ".2", F.relu(OF.conv(sd, k + ".0", r)))
            r = OF.channel_attention(y, sd[k + ".3.conv_du.0.weight"], sd[k + ".3.conv_du.0.bias"], sd[k + ".3.conv_du.2.weight"], sd[k + ".3.conv_du.2.bias"]) + r
        res = OF.conv(sd, f"body.{gi}.body.{nb}", r) + res
        feats.insert(0, res)  # res1 = cat([res.unsqueeze(1), res1], 1): newest first (:98-101)
    res = OF.conv(sd, f"body.{ng}", res)
    feats.insert(0, res)
    out1 = res
    st = torch.stack(feats, 1)  # [B, N, C, H, W]
    b, n, c, h, w = st.shape
    q = st.reshape(b, n, -1)
    energy = q @ q.transpose(1, 2)
    att = torch.softmax(energy.max(-1, keepdim=True)[0] - energy, dim=-1)
    la = (sd["la.gamma"] * (att @ q).reshape(b, n, c, h, w) + st).reshape(b, n * c, h, w)
    out2 = OF.conv(sd, "last_conv", la)
    a3 = torch.sigmoid(F.conv3d(out1.unsqueeze(1), sd["csa.conv.weight"], sd["csa.conv.bias"], padding=1))
    out1 = out1 * (sd["csa.gamma"] * a3).reshape(b, c, h, w) + out1
    res = OF.conv(sd, "last", torch.cat([out1, out2], 1)) + x
    y = OF.conv(sd, "tail.1", OF.upsampler(sd, "tail.0", res, cfg["scale"]))
    return _mean_shift(y, cfg["img_range"], +1)


FORWARDS: Dict[str, Callable] = {
    "SwinFIR": swinir_forward,
    "HAN": han_forward,
    "SwinIR": swinir_forward,
    "EDSR": edsr_forward,
    "RCAN": rcan_forward,
    "HAT": hat_forward,
}


# --------------------------------------------------------------------------- #
# Model.inference  (studiosr/models/common.py:36-67)
# --------------------------------------------------------------------------- #
@torch.inference_mode()
def inference(forward: Callable[[Tensor], Tensor], image: np.ndarray, img_range: float) -> np.ndarray:
    """uint8 HWC -> /scale -> NCHW -> forward -> *scale -> round-half-even -> clip -> uint8
    with scale = 255 iff img_range == 1.0 (studiosr/models/common.py:36-48)."""
    s = 255.0 if img_range == 1.0 else 1.0
    x = torch.from_numpy(image.astype(np.float32) / s).permute(2, 0, 1)[None]
    y = forward(x)[0].permute(1, 2, 0) * s
    return OF.to_uint8(y).numpy()


@torch.inference_mode()
def inference_with_self_ensemble(forward: Callable[[Tensor], Tensor], image: np.ndarray, img_range: float):
    """8 sequential rot/flip forwards averaged (studiosr/models/common.py:50-67)."""
    s = 255.0 if img_range == 1.0 else 1.0
    img = torch.from_numpy(image.astype(np.float32) / s)
    outs = [forward(v.permute(2, 0, 1)[None])[0].permute(1, 2, 0) for v in OF.ensemble_variants(img)]
    return OF.to_uint8(OF.ensemble_merge(outs) * s).numpy()
