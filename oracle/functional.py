"""Oracle building blocks (TEST INFRASTRUCTURE, see oracle/__init__.py).

Each function restates one reference routine in the simplest possible form and
cites the reference file:line it follows (paths relative to /root/reference).
Integer index maps are written as explicit index arithmetic so that they can be
compared bit-for-bit with the HIP kernels.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------- #
# integer index maps (bit-exact contracts)
# --------------------------------------------------------------------------- #
def pixel_shuffle_index_map(c_out: int, h: int, w: int, r: int) -> np.ndarray:
    """For every output element (c, y, x) of PixelShuffle(r) the flat index of the
    input element it copies, input laid out [c_out*r*r, h, w].

    nn.PixelShuffle as used by studiosr/models/common.py:129,133,136:
    out[c, h*r+i, w*r+j] = in[c*r*r + i*r + j, h, w].
    """
    c = np.arange(c_out).reshape(-1, 1, 1)
    y = np.arange(h * r).reshape(1, -1, 1)
    x = np.arange(w * r).reshape(1, 1, -1)
    src_c = c * r * r + (y % r) * r + (x % r)
    return (src_c * h + (y // r)) * w + (x // r)


def pixel_shuffle(x: Tensor, r: int) -> Tensor:
    """PixelShuffle on [B, C*r*r, H, W] via the explicit index map above."""
    b, c_in, h, w = x.shape
    c_out = c_in // (r * r)
    idx = torch.from_numpy(pixel_shuffle_index_map(c_out, h, w, r).reshape(-1))
    return x.reshape(b, -1)[:, idx].reshape(b, c_out, h * r, w * r)


def window_token_source(h: int, w: int, ws: int, shift: int) -> np.ndarray:
    """[nW, ws*ws] flat pixel index (y*w+x) that token t of window k reads after
    torch.roll(-shift) + window_partition (studiosr/models/swinir.py:154-158,
    studiosr/models/common.py:236-240).  window_reverse + roll(+shift) writes back
    to the same pixel (swinir.py:164-168, common.py:243-247)."""
    nh, nw = h // ws, w // ws
    out = np.empty((nh * nw, ws * ws), dtype=np.int64)
    for wy in range(nh):
        for wx in range(nw):
            for i in range(ws):
                for j in range(ws):
                    y = (wy * ws + i + shift) % h
                    x = (wx * ws + j + shift) % w
                    out[wy * nw + wx, i * ws + j] = y * w + x
    return out


def window_partition(x: Tensor, ws: int) -> Tensor:
    """[B,H,W,C] -> [B*nW, ws, ws, C] (studiosr/models/common.py:236-240)."""
    b, h, w, c = x.shape
    x = x.reshape(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, c)


def window_reverse(win: Tensor, ws: int, h: int, w: int) -> Tensor:
    """[B*nW, ws, ws, C] -> [B,H,W,C] (studiosr/models/common.py:243-247)."""
    b = win.shape[0] // ((h // ws) * (w // ws))
    x = win.reshape(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, -1)


def region_label(v: int, size: int, ws: int, shift: int) -> int:
    """Label of coordinate v under the slices (0,-ws),(-ws,-shift),(-shift,None)
    of studiosr/models/common.py:253-262.  With shift == 0 the python slices
    slice(-ws, -0) and slice(-0, None) are empty / everything, which makes the last
    assignment win everywhere -- reproduced by applying the slices in order."""
    lab = 0
    for k, (lo, hi) in enumerate(((0, -ws), (-ws, -shift), (-shift, None))):
        idx = range(size)[slice(lo, hi)]
        if v in idx:
            lab = k
    return lab


def calculate_mask(h: int, w: int, ws: int, shift: int) -> Tensor:
    """[nW, N, N] additive mask of 0 / -100 (studiosr/models/common.py:250-274)."""
    hl = np.array([region_label(v, h, ws, shift) for v in range(h)])
    wl = np.array([region_label(v, w, ws, shift) for v in range(w)])
    # cnt of the reference enumerates (h_slice, w_slice) pairs row-major.
    lab = torch.from_numpy(3 * hl[:, None] + wl[None, :]).to(torch.float32)
    mw = window_partition(lab.reshape(1, h, w, 1), ws).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def relative_position_index(ws: int) -> Tensor:
    """[N,N] int64: (dy+ws-1)*(2ws-1) + (dx+ws-1) (studiosr/models/swinir.py:56-67,
    studiosr/models/hat.py:480-492)."""
    n = ws * ws
    ys, xs = np.divmod(np.arange(n), ws)
    dy = ys[:, None] - ys[None, :] + ws - 1
    dx = xs[:, None] - xs[None, :] + ws - 1
    return torch.from_numpy(dy * (2 * ws - 1) + dx)


def relative_position_index_oca(ws: int, overlap_ratio: float) -> Tensor:
    """[ws*ws, wse*wse] int64 index of HAT's overlapping cross attention
    (studiosr/models/hat.py:494-517).  Entries may be negative; the reference
    indexes its table with them, i.e. Python wrap-around semantics."""
    wse = ws + int(overlap_ratio * ws)
    qy, qx = np.divmod(np.arange(ws * ws), ws)
    ky, kx = np.divmod(np.arange(wse * wse), wse)
    dy = ky[None, :] - qy[:, None] + ws - wse + 1
    dx = kx[None, :] - qx[:, None] + ws - wse + 1
    return torch.from_numpy(dy * (ws + wse - 1) + dx)


def pad_eval(x: Tensor, ws: int) -> Tensor:
    """SwinIR eval-mode pad: ALWAYS adds 1..ws rows/cols by edge-inclusive mirror
    (studiosr/models/swinir.py:249-255)."""
    _, _, h, w = x.shape
    hp = (h // ws + 1) * ws
    wp = (w // ws + 1) * ws
    ys = [i if i < h else 2 * h - 1 - i for i in range(hp)]
    xs = [j if j < w else 2 * w - 1 - j for j in range(wp)]
    return x[:, :, ys, :][:, :, :, xs]


def pad_reflect(x: Tensor, ws: int) -> Tensor:
    """Train-mode / HAT pad to the next multiple of ws by edge-exclusive reflection
    (studiosr/models/common.py:277-282)."""
    _, _, h, w = x.shape
    hp = h + (ws - h % ws) % ws
    wp = w + (ws - w % ws) % ws
    ys = [i if i < h else 2 * (h - 1) - i for i in range(hp)]
    xs = [j if j < w else 2 * (w - 1) - j for j in range(wp)]
    if min(ys) < 0 or min(xs) < 0:
        raise RuntimeError("reflect pad larger than the image (same error class as F.pad)")
    return x[:, :, ys, :][:, :, :, xs]


# --------------------------------------------------------------------------- #
# floating-point layers
# --------------------------------------------------------------------------- #
def conv(sd: SD, key: str, x: Tensor) -> Tensor:
    """nn.Conv2d stride 1, 'same' zero pad (studiosr/models/common.py:104-105)."""
    w = sd[key + ".weight"]
    return F.conv2d(x, w, sd.get(key + ".bias"), padding=w.shape[-1] // 2)


def linear(sd: SD, key: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


def layer_norm(sd: SD, key: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], 1e-5)


def mlp(sd: SD, key: str, x: Tensor) -> Tensor:
    """fc1 -> exact-erf GELU -> fc2 (studiosr/models/common.py:189-195)."""
    return linear(sd, key + ".fc2", F.gelu(linear(sd, key + ".fc1", x)))


def upsampler(sd: SD, key: str, x: Tensor, scale: int, direct: bool = False) -> Tensor:
    """conv->PixelShuffle chain (studiosr/models/common.py:124-137)."""
    if direct or (scale & (scale - 1)) != 0:
        return pixel_shuffle(conv(sd, key + ".0", x), scale)
    for i in range(int(math.log2(scale))):
        x = pixel_shuffle(conv(sd, f"{key}.{2 * i}", x), 2)
    return x


def channel_attention(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """x * sigmoid(W2 relu(W1 mean_hw(x))) (studiosr/models/common.py:156-170,
    studiosr/models/hat.py:25-38)."""
    y = x.mean(dim=(2, 3), keepdim=True)
    y = torch.sigmoid(F.conv2d(F.relu(F.conv2d(y, w1, b1)), w2, b2))
    return x * y


def window_attention(
    sd: SD,
    key: str,
    xw: Tensor,
    rpi: Tensor,
    num_heads: int,
    mask: Optional[Tensor],
) -> Tensor:
    """W-MSA on windows xw [B_, N, C] (studiosr/models/swinir.py:78-105,
    studiosr/models/hat.py:85-110)."""
    b_, n, c = xw.shape
    hd = c // num_heads
    qkv = linear(sd, key + ".qkv", xw).reshape(b_, n, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[key + ".relative_position_bias_table"][rpi.reshape(-1)].reshape(n, n, num_heads)
    attn = attn + bias.permute(2, 0, 1)[None]
    if mask is not None:
        nw = mask.shape[0]
        attn = (attn.reshape(b_ // nw, nw, num_heads, n, n) + mask[None, :, None]).reshape(-1, num_heads, n, n)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(b_, n, c)
    return linear(sd, key + ".proj", out)


def shifted_window_msa(
    sd: SD, key: str, x: Tensor, ws: int, shift: int, rpi: Tensor, num_heads: int, mask: Optional[Tensor]
) -> Tensor:
    """roll -> partition -> attention -> reverse -> roll back on x [B,H,W,C]
    (studiosr/models/swinir.py:153-168)."""
    b, h, w, c = x.shape
    if shift > 0:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    xw = window_partition(x, ws).reshape(-1, ws * ws, c)
    aw = window_attention(sd, key, xw, rpi, num_heads, mask)
    x = window_reverse(aw.reshape(-1, ws, ws, c), ws, h, w)
    if shift > 0:
        x = torch.roll(x, (shift, shift), (1, 2))
    return x


def to_uint8(x: Tensor) -> Tensor:
    """round-half-even, clip, uint8 (studiosr/models/common.py:45)."""
    return x.round().clip(0, 255).to(torch.uint8)


def ensemble_variants(img: Tensor) -> Tuple[Tensor, ...]:
    """8 rot/flip variants of an HWC image (studiosr/models/common.py:10-16)."""
    out = []
    for k in range(4):
        r = torch.rot90(img, k, dims=[0, 1])
        out += [r, torch.fliplr(r)]
    return tuple(out)


def ensemble_merge(outs) -> Tensor:
    """inverse transforms and mean (studiosr/models/common.py:19-26)."""
    acc = []
    for i, o in enumerate(outs):
        if i & 1:
            o = torch.fliplr(o)
        acc.append(torch.rot90(o, i // 2, dims=[1, 0]))
    return torch.stack(acc).mean(dim=0)
