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


def bias_distinct_tiles(bias: Tensor) -> Optional[Tensor]:
    """[heads, 256, 256] fp32 bias of 16 x 16 windows -> its 31 distinct 16 x 16 tiles [heads][31][lane][4] in the accumulator-fragment
    order of bias_fragments (SrWindowAttn.bias_tiles, ABI v8), or None when the bias does not have the structure.  A relative-position bias
    table[(qy - ky + 15) * 31 + (qx - kx + 15)] (hat.py:85-110, rpi of common.py:276-290) has: tile (qt, kt) = rows 16 qt.., columns 16 kt.. is one
    (query window row, key window row) pair and depends on qt - kt only.  Checked element for element on the gathered bias."""
    heads, nq, nk = bias.shape
    if nq != 256 or nk != 256:
        return None
    b = bias.reshape(heads, 16, 16, 16, 16).permute(0, 1, 3, 2, 4)  # [h, qt, kt, i, j]
    qt = torch.arange(16, device=bias.device)
    d = qt[:, None] - qt[None, :] + 15  # [qt, kt] -> tile index
    first = torch.stack([b[:, max(t - 15, 0), max(15 - t, 0)] for t in range(31)], 1)  # [h, 31, i, j]
    if not torch.equal(first[:, d], b):
        return None
    return first.reshape(heads, 31, 16, 4, 4).permute(0, 1, 3, 2, 4).contiguous().reshape(-1)  # [h, d, g, i, r] -> lane = g * 16 + i


def oca_rel_index(ws: int = 16, wse: int = 24) -> Tensor:
    """[ws*ws, wse*wse] int64: position of bias[q][k] in the rotated relative-position table of SrOcaAttn.bias_rel,
    (ky - qy + ws - 1) * (ws + wse - 1) + (kx - qx + ws - 1)."""
    n = ws + wse - 1
    qy, qx = torch.div(torch.arange(ws * ws), ws, rounding_mode="floor"), torch.arange(ws * ws) % ws
    ky, kx = torch.div(torch.arange(wse * wse), wse, rounding_mode="floor"), torch.arange(wse * wse) % wse
    return (ky[None, :] - qy[:, None] + ws - 1) * n + (kx[None, :] - qx[:, None] + ws - 1)


def oca_bias_rel(bias: Tensor) -> Optional[Tensor]:
    """[heads, 256, >= 576] fp32 gathered bias of HAT's overlapping cross attention (16 x 16 queries, 24 x 24 keys; hat.py:494-517, 276-279) ->
    its relative-position table [heads][1521] in the order of SrOcaAttn.bias_rel (ABI v8), or None when the bias is not a function of the (row, column)
    differences.  Every entry of the gathered bias is checked against the table."""
    if bias.shape[1] != 256 or bias.shape[2] < 576:
        return None
    J = oca_rel_index().to(bias.device)                      # [256, 576]
    rel = torch.zeros(bias.shape[0], 39 * 39, dtype=torch.float32, device=bias.device)
    rel[:, J.reshape(-1)] = bias[:, :, :576].reshape(bias.shape[0], -1)  # any representative (duplicates carry the same value when the structure holds)
    if not torch.equal(rel[:, J], bias[:, :, :576]):
        return None
    return rel.contiguous()


# --------------------------------------------------------------------------- Swin block weight stream (C ABI v5, sr_swin_block)
LOG2E = 1.4426950408889634
SWIN_STREAM_SLOTS = 48


def _bf16_hi_lo(v: Tensor):
    """v (fp32) as a pair of bf16 numbers hi + lo (returned in fp32): 16 mantissa bits when both ride on constant-one channels."""
    hi = v.to(torch.bfloat16).to(torch.float32)
    lo = (v - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def gelu_bf16_value(x: float) -> float:
    """sr_common.h gelu_bf16 (x * sigmoid form of tanh-GELU) evaluated in fp32, rounded to bf16: what the kernel stores for a
    hidden pad column whose pre-activation is x.  Asserts that the value is far from a bf16 rounding boundary."""
    t = torch.tensor(float(x), dtype=torch.float32)
    sq = t * t
    pp = sq * torch.tensor(-0.1029432, dtype=torch.float32) + torch.tensor(-2.3022082, dtype=torch.float32)
    e = torch.exp2(t * pp)
    g = t / (1.0 + e)
    gb = g.to(torch.bfloat16).to(torch.float32)
    ulp = abs(float(gb)) * 2.0 ** -8
    assert abs(abs(float(g - gb)) - ulp / 2) > 1e-4 * max(abs(float(gb)), 1e-3), "gelu pad value sits on a bf16 rounding boundary"
    return float(gb)


def _tile_fragment(t: Tensor) -> Tensor:
    """[16 rows, 32 k] -> [64 lanes, 8]: lane l holds row l & 15, k = 8 (l >> 4) + j."""
    return t.reshape(16, 4, 8).permute(1, 0, 2).reshape(64, 8)


def _split16(v: Tensor):
    """v (fp32) -> (h, v - h) with h the value's leading 16 mantissa bits (what a split-operand hi | lo fragment can hold)."""
    hi = v.to(torch.bfloat16).to(torch.float32)
    h = hi + (v - hi).to(torch.bfloat16).to(torch.float32)
    return h, v - h


def pack_swin_block_stream(qkv_w: Tensor, qkv_b: Optional[Tensor], proj_w: Tensor, proj_b: Optional[Tensor], fc1_w: Tensor, fc1_b: Optional[Tensor],
                           fc2_w: Tensor, fc2_b: Optional[Tensor], C: int, heads: int, hidden: int, x3: bool = False) -> Tensor:
    """The ONE weight stream of a SwinTransformerBlock for sr_swin_block (include/studiosr_hip.h SrSwinBlock; swinir.py:78-105,146-174,
    common.py:173-195): 48 slots x 12 fragments x [64 lanes][8] bf16 in the order the kernel consumes them --
    per pass p (heads 2p, 2p+1): 6 QKV slots (K-chunks), 2 proj slots; then 6 fc1 + 6 fc2 slots per hidden half.
    Wave w of the 4-wave workgroup reads fragments 3w .. 3w+2 of every slot.  All weights are the LayerNorm-folded fp32 matrices
    (packing.fold_layernorm).  Folded in here: attention scale and log2(e) into the q rows; every bias as a hi + lo bf16 pair in the
    columns of the constant-one channels (LayerNorm image channels C, C+1; O feature hd of heads 0 / 1; hidden columns `hidden`, +1);
    v's pad feature hd := 1 (softmax denominator); the v bias into the proj bias (softmax rows sum to one); the k bias is dropped
    (it shifts every logit of a row equally).
    x3 = True: the split-operand stream of compute type SR_BF16X3 (precision "fp32x3"): every element as hi | lo (per lane 8 hi then 8 lo),
    the two bias channels carry the bias's leading 16 bits and the rest, and the hidden pad columns are set to 1 by the kernel itself."""
    assert C == 180 and heads == 6 and hidden == 360, "sr_swin_block geometry"
    hd, hdp, Cp, Hp = C // heads, 32, 192, 384
    dev = qkv_w.device
    f32 = torch.float32
    qkv_w, proj_w, fc1_w, fc2_w = (t.detach().to(f32) for t in (qkv_w, proj_w, fc1_w, fc2_w))
    zeros = lambda n: torch.zeros(n, dtype=f32, device=dev)
    qkv_b = zeros(3 * C) if qkv_b is None else qkv_b.detach().to(f32)
    proj_b = zeros(C) if proj_b is None else proj_b.detach().to(f32)
    fc1_b = zeros(hidden) if fc1_b is None else fc1_b.detach().to(f32)
    fc2_b = zeros(C) if fc2_b is None else fc2_b.detach().to(f32)

    # ---- padded matrices
    qs = hd ** -0.5 * LOG2E
    M_qkv = torch.zeros(3, heads, hdp, Cp, dtype=f32, device=dev)
    M_qkv[:, :, :hd, :C] = qkv_w.reshape(3, heads, hd, C)
    M_qkv[0] *= qs
    split = _split16 if x3 else _bf16_hi_lo
    bq_hi, bq_lo = split(qkv_b[:C].reshape(heads, hd) * qs)
    M_qkv[0, :, :hd, C] = bq_hi
    M_qkv[0, :, :hd, C + 1] = bq_lo
    M_qkv[2, :, hd, C] = 1.0  # v[:, hd] = 1: row hd of O^T is the softmax denominator

    bpf = proj_b + proj_w @ qkv_b[2 * C:]  # proj(o + b_v) = proj(o) + W_proj b_v
    M_proj = torch.zeros(Cp, heads, hdp, dtype=f32, device=dev)
    M_proj[:C, :, :hd] = proj_w.reshape(C, heads, hd)
    bp_hi, bp_lo = split(bpf)
    M_proj[:C, 0, hd] = bp_hi  # O[:, head, hd] = 1 after the softmax normalisation
    M_proj[:C, 1, hd] = bp_lo

    M_fc1 = torch.zeros(Hp, Cp, dtype=f32, device=dev)
    M_fc1[:hidden, :C] = fc1_w
    b1_hi, b1_lo = split(fc1_b)
    M_fc1[:hidden, C] = b1_hi
    M_fc1[:hidden, C + 1] = b1_lo
    if x3:
        c0 = 1.0  # the split-operand kernel writes 1.0 into hidden columns `hidden`, `hidden`+1 itself
    else:
        v0 = 1.0
        c0 = gelu_bf16_value(v0)  # hidden columns `hidden`, `hidden`+1 hold the constant c0 = gelu(v0)
        M_fc1[hidden, C] = v0
        M_fc1[hidden + 1, C] = v0

    M_fc2 = torch.zeros(Cp, Hp, dtype=f32, device=dev)
    M_fc2[:C, :hidden] = fc2_w
    b2_hi, b2_lo = split(fc2_b / c0)
    M_fc2[:C, hidden] = b2_hi
    M_fc2[:C, hidden + 1] = b2_lo

    # ---- slots: [slot][fragment 3 w + t][lane = 16 g + i][j] with element (row 16 tile + i, k = 32 chunk + 8 g + j)
    out = torch.empty(SWIN_STREAM_SLOTS, 12, 64, 8, dtype=f32, device=dev)
    att = out[:24].reshape(3, 8, 12, 64, 8)
    # QKV: M_qkv[t, head = 2p + hh, d = 16 half + i, k = 32 c + 8 g + j] -> att[p, c, 3 (2 hh + half) + t, 16 g + i, j]
    att[:, :6] = M_qkv.reshape(3, 3, 2, 2, 16, 6, 4, 8).permute(1, 5, 2, 3, 0, 6, 4, 7).reshape(3, 6, 12, 64, 8)
    # proj: M_proj[ch = 48 w + 16 n + i, head = 2p + c2, d = 8 g + j] -> att[p, 6 + c2, 3 w + n, 16 g + i, j]
    att[:, 6:] = M_proj.reshape(4, 3, 16, 3, 2, 4, 8).permute(3, 4, 0, 1, 5, 2, 6).reshape(3, 2, 12, 64, 8)
    mlp = out[24:].reshape(2, 12, 12, 64, 8)
    # fc1: M_fc1[row = 192 hf + 48 w + 16 n + i, k = 32 c + 8 g + j] -> mlp[hf, c, 3 w + n, 16 g + i, j]
    mlp[:, :6] = M_fc1.reshape(2, 4, 3, 16, 6, 4, 8).permute(0, 4, 1, 2, 5, 3, 6).reshape(2, 6, 12, 64, 8)
    # fc2: M_fc2[ch = 48 w + 16 n + i, k = 192 hf + 32 c + 8 g + j] -> mlp[hf, 6 + c, 3 w + n, 16 g + i, j]
    mlp[:, 6:] = M_fc2.reshape(4, 3, 16, 2, 6, 4, 8).permute(3, 4, 0, 1, 5, 2, 6).reshape(2, 6, 12, 64, 8)
    if x3:
        hi = out.to(torch.bfloat16)
        lo = (out - hi.to(f32)).to(torch.bfloat16)
        return torch.cat([hi, lo], dim=-1).reshape(-1).contiguous()  # [slot][fragment][lane][8 hi | 8 lo]
    return out.to(torch.bfloat16).reshape(-1).contiguous()


SWIN_TAIL_SLOTS = 30


def _hi_lo_cat(out: Tensor) -> Tensor:
    """fp32 [..., 8] fragments -> split-operand layout [..., 8 hi | 8 lo] bf16 (compute type SR_BF16X3)."""
    hi = out.to(torch.bfloat16)
    lo = (out - hi.to(torch.float32)).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=-1)


def pack_swin_tail_stream(proj_w: Tensor, fc1_w: Tensor, fc1_b: Optional[Tensor], fc2_w: Tensor, fc2_b: Optional[Tensor], C: int, heads: int,
                          hidden: int, x3: bool = False) -> Tensor:
    """The weight stream of sr_swin_tail (include/studiosr_hip.h SrSwinTail; hat.py:172-194): 30 slots x 12 fragments x [64 lanes][8] bf16 --
    6 projection slots (slot = head: K = the head's 32 padded features of the attention output, no bias: the kernel adds bproj itself),
    then the 24 MLP slots of pack_swin_block_stream (fc1 with LayerNorm2 folded by the caller; fc1 / fc2 biases on the constant-one pad
    channels)."""
    assert C == 180 and heads == 6 and hidden == 360, "sr_swin_tail geometry"
    hd, hdp, Cp = C // heads, 32, 192
    dev = proj_w.device
    f32 = torch.float32
    M_proj = torch.zeros(Cp, heads, hdp, dtype=f32, device=dev)
    M_proj[:C, :, :hd] = proj_w.detach().to(f32).reshape(C, heads, hd)
    # M_proj[ch = 48 w + 16 n + i, head = c, d = 8 g + j] -> [c, 3 w + n, 16 g + i, j]
    proj = M_proj.reshape(4, 3, 16, heads, 4, 8).permute(3, 0, 1, 4, 2, 5).reshape(heads, 12, 64, 8)
    proj = (_hi_lo_cat(proj) if x3 else proj.to(torch.bfloat16)).reshape(-1)  # x3: [slot][fragment][lane][8 hi | 8 lo]
    zero = torch.zeros(3 * C, C, dtype=f32, device=dev)
    full = pack_swin_block_stream(zero, None, proj_w, None, fc1_w, fc1_b, fc2_w, fc2_b, C, heads, hidden, x3=x3)
    return torch.cat([proj, full.reshape(SWIN_STREAM_SLOTS, -1)[24:].reshape(-1)]).contiguous()


SWIN_QKV_SLOTS = 18


def pack_swin_qkv_stream(qkv_w: Tensor, qkv_b: Optional[Tensor], C: int, heads: int, x3: bool = False) -> Tensor:
    """The weight stream of sr_swin_qkv (include/studiosr_hip.h SrSwinQkv; hat.py:164-176, 55-83): 18 slots x 12 fragments x [64 lanes][8] bf16
    -- per pass p (heads 2p, 2p+1) six K-chunk slots; wave w = (head 2p + (w >> 1), d-half w & 1) reads fragments 3w .. 3w+2 = its q, k, v
    tiles.  qkv_w / qkv_b are the LayerNorm-folded fp32 matrices; the attention scale hd^-0.5 goes into the q rows (plain exp softmax in the
    attention kernel: no log2(e) here), all three biases as hi + lo bf16 pairs into the columns of the constant-one channels C, C+1."""
    assert C == 180 and heads == 6, "sr_swin_qkv geometry"
    hd, hdp, Cp = C // heads, 32, 192
    dev = qkv_w.device
    f32 = torch.float32
    qkv_w = qkv_w.detach().to(f32)
    qkv_b = torch.zeros(3 * C, dtype=f32, device=dev) if qkv_b is None else qkv_b.detach().to(f32)
    M = torch.zeros(3, heads, hdp, Cp, dtype=f32, device=dev)
    M[:, :, :hd, :C] = qkv_w.reshape(3, heads, hd, C)
    b = qkv_b.reshape(3, heads, hd).clone()
    M[0] *= hd ** -0.5
    b[0] *= hd ** -0.5
    b_hi, b_lo = (_split16 if x3 else _bf16_hi_lo)(b)
    M[:, :, :hd, C] = b_hi
    M[:, :, :hd, C + 1] = b_lo
    # M[t, head = 2p + hh, d = 16 half + i, k = 32 c + 8 g + j] -> [p, c, 3 (2 hh + half) + t, 16 g + i, j]
    out = M.reshape(3, 3, 2, 2, 16, 6, 4, 8).permute(1, 5, 2, 3, 0, 6, 4, 7).reshape(3, 6, 12, 64, 8)
    return (_hi_lo_cat(out) if x3 else out.to(torch.bfloat16)).reshape(-1).contiguous()
