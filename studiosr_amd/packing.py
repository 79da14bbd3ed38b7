"""One-time weight layout transforms (host side, at model-load time).

The HIP kernels read weights in MFMA fragment order (include/studiosr_hip.h):
    Wp[n_tile][k_chunk][lane][8],  element (n = 16*n_tile + (lane & 15), k = 32*k_chunk + 8*(lane >> 4) + j)
so a wave fetches one operand fragment with a single contiguous 1 KiB (bf16) read.  Everything here is
index shuffling of the reference's parameter tensors (state_dict layouts of studiosr/models/*.py); no
arithmetic except folding the attention scale into Wq (swinir.py:83, hat.py:90).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

Tensor = torch.Tensor


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pad_vec(v: Optional[Tensor], n_pad: int, idx: Optional[Tensor] = None) -> Tensor:
    """fp32 vector of length n_pad; element i = v[idx[i]] (idx < 0 -> 0), or v zero-extended."""
    out = torch.zeros(n_pad, dtype=torch.float32, device=v.device if v is not None else None)
    if v is None:
        return out
    v = v.detach().to(torch.float32)
    if idx is None:
        out[: v.numel()] = v.reshape(-1)
    else:
        idx = idx.to(v.device)
        m = idx >= 0
        out[m] = v.reshape(-1)[idx[m]]
    return out


def expand_matrix(w: Tensor, row_idx: Tensor, col_idx: Tensor) -> Tensor:
    """[len(row_idx), len(col_idx)] fp32 with out[i, j] = w[row_idx[i], col_idx[j]] (index < 0 -> 0)."""
    w = w.detach().to(torch.float32)
    row_idx, col_idx = row_idx.to(w.device), col_idx.to(w.device)
    out = w[row_idx.clamp(min=0)][:, col_idx.clamp(min=0)]
    out = out * (row_idx >= 0).to(out.dtype)[:, None] * (col_idx >= 0).to(out.dtype)[None, :]
    return out


def to_fragments(wm: Tensor, dtype: torch.dtype) -> Tensor:
    """[N_p, K_p] (N_p % 16 == 0, K_p % 32 == 0) -> flat fragment-ordered tensor of `dtype`."""
    n_p, k_p = wm.shape
    assert n_p % 16 == 0 and k_p % 32 == 0, (n_p, k_p)
    t = wm.reshape(n_p // 16, 16, k_p // 32, 4, 8).permute(0, 2, 3, 1, 4)  # [ntile, kchunk, g, n16, j]
    if isinstance(dtype, str):  # "bf16x3" (SR_BF16X3): per lane 8 hi then 8 lo bf16, lo = bf16(w - hi)
        assert dtype == "bf16x3", dtype
        t = t.contiguous().to(torch.float32)
        hi = t.to(torch.bfloat16)
        lo = (t - hi.to(torch.float32)).to(torch.bfloat16)
        return torch.cat([hi, lo], dim=-1).contiguous().reshape(-1)
    return t.contiguous().to(dtype).reshape(-1)


def identity_idx(n: int, n_pad: int) -> Tensor:
    idx = torch.full((n_pad,), -1, dtype=torch.long)
    idx[:n] = torch.arange(n)
    return idx


def head_idx(heads: int, hd: int, hd_p: int) -> Tensor:
    """padded [head][hd_p] channel -> real head*hd + d (or -1)."""
    idx = torch.full((heads, hd_p), -1, dtype=torch.long)
    idx[:, :hd] = torch.arange(heads * hd).reshape(heads, hd)
    return idx.reshape(-1)


def pack_linear(w: Tensor, b: Optional[Tensor], row_idx: Tensor, col_idx: Tensor, dtype, row_scale: Optional[Tensor] = None):
    """nn.Linear weight [out, in] -> (fragments, padded fp32 bias)."""
    wm = expand_matrix(w, row_idx, col_idx)
    bias = pad_vec(b, len(row_idx), row_idx) if b is not None else torch.zeros(len(row_idx), device=w.device)
    if row_scale is not None:
        row_scale = row_scale.to(wm.device)
        wm = wm * row_scale[:, None]
        bias = bias * row_scale
    return to_fragments(wm, dtype), bias.contiguous()


def fold_layernorm(w: Tensor, b: Optional[Tensor], gamma: Tensor, beta: Tensor):
    """LN(x) @ W^T + b  ==  ((x - mean) * rstd) @ (W * gamma)^T + (b + W @ beta): returns (W', b') in fp32."""
    w32 = w.detach().to(torch.float32)
    g32, be32 = gamma.detach().to(torch.float32), beta.detach().to(torch.float32)
    b32 = torch.zeros(w32.shape[0], device=w32.device) if b is None else b.detach().to(torch.float32)
    return w32 * g32[None, :], b32 + w32 @ be32


def pack_qkv(w: Tensor, b: Tensor, C: int, C_p: int, heads: int, hd_p: int, dtype):
    """qkv Linear [3C, C]: padded row n' = part*heads*hd_p + head*hd_p + d; q rows pre-multiplied by hd**-0.5."""
    hd = C // heads
    h_idx = head_idx(heads, hd, hd_p)
    rows = torch.cat([torch.where(h_idx >= 0, h_idx + p * C, h_idx) for p in range(3)])
    scale = torch.ones(rows.numel())
    scale[: heads * hd_p] = hd ** -0.5
    return pack_linear(w, b, rows, identity_idx(C, C_p), dtype, scale)


def pack_conv3x3(w: Tensor, b: Optional[Tensor], cin_p: int, row_idx: Tensor, dtype):
    """nn.Conv2d weight [Cout, Cin, 3, 3] -> implicit-GEMM matrix [len(row_idx), 9*cin_p] with
    k = (ky*3 + kx)*cin_p + c, then fragments."""
    cout, cin = w.shape[:2]
    w = w.detach().to(torch.float32)
    wk = torch.zeros(cout, 3, 3, cin_p, dtype=torch.float32, device=w.device)
    wk[:, :, :, :cin] = w.permute(0, 2, 3, 1)
    wm = wk.reshape(cout, 9 * cin_p)
    row_idx = row_idx.to(w.device)
    wm = wm[row_idx.clamp(min=0)] * (row_idx >= 0).to(wm.dtype)[:, None]
    bias = pad_vec(b, len(row_idx), row_idx) if b is not None else torch.zeros(len(row_idx), device=w.device)
    return to_fragments(wm, dtype), bias.contiguous()


def pixel_shuffle_rows(c_ps: int, cps_p: int, r: int) -> Tensor:
    """Row order of a conv feeding nn.PixelShuffle(r): packed row (i*r + j)*cps_p + c takes the
    reference's output channel c*r*r + i*r + j (studiosr/models/common.py:129,133,136)."""
    idx = torch.full((r * r, cps_p), -1, dtype=torch.long)
    c = torch.arange(c_ps)
    for i in range(r):
        for j in range(r):
            idx[i * r + j, :c_ps] = c * r * r + i * r + j
    return idx.reshape(-1)


def gather_bias(table: Tensor, rpi: Tensor, n_q: int, n_k: int) -> Tensor:
    """[heads, n_q, n_k] fp32 = table[rpi] (negative indices wrap, as the reference's python indexing
    does for HAT's OCA index; swinir.py:86-91, hat.py:93-96,276-279)."""
    t = table.detach().to(torch.float32)
    idx = rpi.reshape(-1).to(t.device)
    idx = torch.where(idx < 0, idx + t.shape[0], idx)
    return t[idx].reshape(n_q, n_k, -1).permute(2, 0, 1).contiguous()


def bias_fragments(bias: Tensor) -> Tensor:
    """[heads, Nq, Nk] fp32 -> accumulator-fragment order [heads][qt][kt][lane][4] with
    element = bias[h][16*qt + (lane & 15)][16*kt + 4*(lane >> 4) + r]: the fused attention kernel then
    initialises each S^T accumulator tile with ONE coalesced 1 KiB load."""
    heads, nq, nk = bias.shape
    b = bias.reshape(heads, nq // 16, 16, nk // 16, 4, 4)  # [h, qt, q16, kt, g, r]
    return b.permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)  # [h, qt, kt, g, q16, r] -> lane = g*16 + q16
