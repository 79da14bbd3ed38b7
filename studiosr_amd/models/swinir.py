"""SwinIR on the MI355X HIP hot path (reference: studiosr/models/swinir.py).

Same constructor kwargs, attributes, `state_dict` keys/shapes (official checkpoints load unchanged) and
`get_model_config` / `get_training_config` / `from_pretrained` surface as the reference class.  `forward`
is not a torch graph: it is a fixed sequence of C-ABI launches over NHWC buffers
(2 + 1 per block (fused) + 1 per RSTB + 6 kernels):

  ingest (pad + normalise, swinir.py:249-255,356-359)  -> conv_first (:361)  -> LayerNorm (:28-32)
  per block (:146-174):  [LN1 + QKV GEMM, rows gathered through roll+partition] -> window attention
                         -> [proj GEMM + residual, rows scattered through reverse+roll]
                         -> [LN2 + fc1 GEMM + GELU] -> [fc2 GEMM + residual]
  per RSTB (:245-246):   conv3x3 + residual
  LayerNorm (:349) -> conv_after_body + long skip (:362) -> conv_before_upsample + LeakyReLU (:365)
  -> Upsampler convs storing through PixelShuffle (:366) -> conv_last + unnormalise + crop to NCHW (:366-372)
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops, packing
from ..runtime import capturing_or_warming_up, compute_dtype, knob, sr_dtype
from .common import RGB_MEAN, Model, Upsampler, conv_call, pack_upsampler, run_upsampler

Tensor = torch.Tensor


# --------------------------------------------------------------------------- parameter containers
class WindowAttention(nn.Module):
    """Parameters of swinir.py:35-76 (qkv, proj, bias table, index buffer)."""

    def __init__(self, dim: int, window_size: int, num_heads: int) -> None:
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        ws = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        ys, xs = torch.div(torch.arange(ws * ws), ws, rounding_mode="floor"), torch.arange(ws * ws) % ws
        rpi = (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)
        self.register_buffer("relative_position_index", rpi)  # swinir.py:56-67
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class MlpParams(nn.Module):
    """common.py:173-187."""

    def __init__(self, dim: int, hidden: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class SwinTransformerBlock(nn.Module):
    """Parameters of swinir.py:108-144."""

    def __init__(self, dim: int, num_heads: int, window_size: int, shift_size: int, mlp_ratio: float) -> None:
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.shift_size = shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MlpParams(dim, int(dim * mlp_ratio))


class BasicLayer(nn.Module):
    def __init__(self, dim: int, depth: int, num_heads: int, window_size: int, mlp_ratio: float) -> None:
        super().__init__()
        self.blocks = nn.ModuleList(
            [SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio) for i in range(depth)]
        )


class RSTB(nn.Module):
    """swinir.py:216-246."""

    def __init__(self, dim: int, depth: int, num_heads: int, window_size: int, mlp_ratio: float) -> None:
        super().__init__()
        self.residual_group = BasicLayer(dim, depth, num_heads, window_size, mlp_ratio)
        self.conv = nn.Conv2d(dim, dim, 3, 1, 1)


class PatchEmbed(nn.Module):
    def __init__(self, embed_dim: int) -> None:
        super().__init__()
        self.norm = nn.LayerNorm(embed_dim)


# --------------------------------------------------------------------------- shared transformer pieces
class SwinGeometry:
    """Padded sizes shared by SwinIR and HAT."""

    def __init__(self, C: int, heads: int, ws: int, hidden: int) -> None:
        self.C, self.heads, self.ws, self.hidden = C, heads, ws, hidden
        self.Cp = packing.round_up(C, 64)
        self.hd = C // heads
        self.hd_p = packing.round_up(self.hd, 32)
        if (heads * self.hd_p) % 64:
            self.hd_p = packing.round_up(self.hd, 64)
        self.HP = heads * self.hd_p
        self.hid_p = packing.round_up(hidden, 64)
        self.ntok = ws * ws


def pack_ln(norm: nn.LayerNorm, n_pad: int):
    return packing.pad_vec(norm.weight, n_pad).contiguous(), packing.pad_vec(norm.bias, n_pad).contiguous()


def fold_ln(dt: torch.dtype) -> bool:
    """bf16 kernels take LayerNorm's gamma/beta folded into the following Linear (one less dependent load
    and 2 fewer VALU ops per element in the prologue); the exact-fp32 path keeps the reference's op order."""
    return dt == torch.bfloat16


def pack_attention(attn: nn.Module, geo: SwinGeometry, dt: torch.dtype, rpi: Optional[Tensor] = None, norm: Optional[nn.LayerNorm] = None) -> Dict:
    C, Cp, heads, hd_p = geo.C, geo.Cp, geo.heads, geo.hd_p
    qw, qb = attn.qkv.weight, attn.qkv.bias
    if norm is not None and fold_ln(dt):
        qw, qb = packing.fold_layernorm(qw, qb, norm.weight, norm.bias)
    qkv_w, qkv_b = packing.pack_qkv(qw, qb, C, Cp, heads, hd_p, dt)
    proj_w, proj_b = packing.pack_linear(attn.proj.weight, attn.proj.bias, packing.identity_idx(C, Cp), packing.head_idx(heads, geo.hd, hd_p), dt)
    rpi = attn.relative_position_index if rpi is None else rpi
    bias = packing.gather_bias(attn.relative_position_bias_table, rpi, geo.ntok, geo.ntok)
    # fused kernel: softmax rows sum to 1, so P (v + b_v) = P v + b_v and proj(o + b_v) = proj(o) + W_proj b_v (exact algebra);
    # the k bias shifts every logit of a row equally and cancels in the softmax.
    bv = qb.detach().to(torch.float32)[2 * C :] if qb is not None else torch.zeros(C, device=qw.device)
    pb = attn.proj.bias.detach().to(torch.float32) if attn.proj.bias is not None else torch.zeros(C, device=qw.device)
    proj_b_fused = packing.pad_vec(pb + attn.proj.weight.detach().to(torch.float32) @ bv, Cp).contiguous()
    out = dict(qkv_w=qkv_w, qkv_b=qkv_b, proj_w=proj_w, proj_b=proj_b, proj_b_fused=proj_b_fused, bias=bias, bias_frag=packing.bias_fragments(bias))
    if geo.ntok == 256 and hd_p == 32 and dt == torch.bfloat16:  # 16 x 16 windows: the 31 distinct bias tiles select the LDS form of the attention (ABI v8)
        tiles = packing.bias_distinct_tiles(bias)
        if tiles is not None:
            out["bias_tiles"] = tiles
    return out


def pack_mlp(mlp: nn.Module, geo: SwinGeometry, dt: torch.dtype, norm: Optional[nn.LayerNorm] = None) -> Dict:
    w1, b1 = mlp.fc1.weight, mlp.fc1.bias
    if norm is not None and fold_ln(dt):
        w1, b1 = packing.fold_layernorm(w1, b1, norm.weight, norm.bias)
    fc1_w, fc1_b = packing.pack_linear(w1, b1, packing.identity_idx(geo.hidden, geo.hid_p), packing.identity_idx(geo.C, geo.Cp), dt)
    fc2_w, fc2_b = packing.pack_linear(mlp.fc2.weight, mlp.fc2.bias, packing.identity_idx(geo.C, geo.Cp), packing.identity_idx(geo.hidden, geo.hid_p), dt)
    return dict(fc1_w=fc1_w, fc1_b=fc1_b, fc2_w=fc2_w, fc2_b=fc2_b)


def pack_block_stream(blk: nn.Module, geo: SwinGeometry, dt: torch.dtype) -> Dict:
    """The single weight stream of sr_swin_block (ABI v5) for one SwinTransformerBlock, when the kernel covers the geometry: bf16 operands
    for the bf16 path, split operands (hi | lo) for precision "fp32x3"; the exact-fp32 parity path has no fused block kernel."""
    from ..runtime import X3_KEY

    x3 = dt == X3_KEY
    if not ((x3 or fold_ln(dt)) and geo.C == 180 and geo.heads == 6 and geo.ws == 8 and geo.hidden == 360):
        return {}
    attn, mlp = blk.attn, blk.mlp
    qw, qb = packing.fold_layernorm(attn.qkv.weight, attn.qkv.bias, blk.norm1.weight, blk.norm1.bias)
    w1, b1 = packing.fold_layernorm(mlp.fc1.weight, mlp.fc1.bias, blk.norm2.weight, blk.norm2.bias)
    stream = packing.pack_swin_block_stream(qw, qb, attn.proj.weight, attn.proj.bias, w1, b1, mlp.fc2.weight, mlp.fc2.bias, geo.C, geo.heads, geo.hidden, x3=x3)
    bias = packing.gather_bias(attn.relative_position_bias_table, attn.relative_position_index, geo.ntok, geo.ntok)
    return dict(stream=stream, bias_frag_l2=packing.bias_fragments(bias * packing.LOG2E), stream_dtype=L.SR_BF16X3 if x3 else L.SR_BF16)


def pack_light_block(blk: nn.Module, geo: SwinGeometry, dt: torch.dtype) -> Dict:
    """Operands of sr_swin_light (ABI v7: the whole block of the lightweight geometry, swinir.py:418-427, in one launch) when it covers the
    geometry (bf16 path): heads padded to 16 features, LayerNorm affines folded, attention scale in the q rows."""
    from ..runtime import X3_KEY

    x3 = dt == X3_KEY  # split operands (hi | lo) for precision "fp32x3": what inference() runs by default
    if not ((x3 or fold_ln(dt)) and geo.Cp == 64 and geo.heads == 6 and geo.hd <= 16 and geo.ws == 8 and 64 < geo.hidden <= 128 and 48 < geo.C):
        return {}
    attn, mlp = blk.attn, blk.mlp
    C, hd = geo.C, geo.hd
    qw, qb = packing.fold_layernorm(attn.qkv.weight, attn.qkv.bias, blk.norm1.weight, blk.norm1.bias)
    w1, b1 = packing.fold_layernorm(mlp.fc1.weight, mlp.fc1.bias, blk.norm2.weight, blk.norm2.bias)
    wqkv, bqkv = packing.pack_qkv(qw, qb, C, 64, geo.heads, 16, dt)
    wproj, bproj = packing.pack_linear(attn.proj.weight, attn.proj.bias, packing.identity_idx(C, 64), packing.head_idx(geo.heads, hd, 16), dt)
    fc1, fb1 = packing.pack_linear(w1, b1, packing.identity_idx(geo.hidden, 128), packing.identity_idx(C, 64), dt)
    fc2, fb2 = packing.pack_linear(mlp.fc2.weight, mlp.fc2.bias, packing.identity_idx(C, 64), packing.identity_idx(geo.hidden, 128), dt)
    return dict(light=(wqkv, bqkv, wproj, bproj, fc1, fb1, fc2, fb2), light_dtype=L.SR_BF16X3 if x3 else L.SR_BF16)


def pack_tail_stream(proj: nn.Module, mlp: nn.Module, norm2: nn.Module, geo: SwinGeometry, dt: torch.dtype) -> Dict:
    """Weight stream of sr_swin_tail (ABI v6: projection + shortcut + LayerNorm2 + MLP behind a separate attention kernel; hat.py:172-194,
    286-293) when the kernel covers the geometry: bf16 operands, or split operands (hi | lo) for precision "fp32x3"."""
    from ..runtime import X3_KEY

    x3 = dt == X3_KEY
    if not ((x3 or fold_ln(dt)) and geo.C == 180 and geo.heads == 6 and geo.hidden == 360 and geo.ws in (8, 16)):
        return {}
    w1, b1 = packing.fold_layernorm(mlp.fc1.weight, mlp.fc1.bias, norm2.weight, norm2.bias)
    return dict(tail_stream=packing.pack_swin_tail_stream(proj.weight, w1, b1, mlp.fc2.weight, mlp.fc2.bias, geo.C, geo.heads, geo.hidden, x3=x3),
                tail_dtype=L.SR_BF16X3 if x3 else L.SR_BF16)


def pack_qkv_stream(attn: nn.Module, norm1: nn.Module, geo: SwinGeometry, dt: torch.dtype) -> Dict:
    """Weight stream of sr_swin_qkv (ABI v6: LayerNorm1 + QKV projection in front of sr_window_attention; hat.py:164-176): bf16 operands, or split
    operands (hi | lo) for precision "fp32x3"."""
    from ..runtime import X3_KEY

    x3 = dt == X3_KEY
    if not ((x3 or fold_ln(dt)) and geo.C == 180 and geo.heads == 6 and geo.ws in (8, 16)):
        return {}
    qw, qb = packing.fold_layernorm(attn.qkv.weight, attn.qkv.bias, norm1.weight, norm1.bias)
    return dict(qkv_stream=packing.pack_swin_qkv_stream(qw, qb, geo.C, geo.heads, x3=x3), qkv_dtype=L.SR_BF16X3 if x3 else L.SR_BF16)


def stream_compute_dtype(cdt: torch.dtype) -> int:
    """Compute type of the stream-form kernels for this forward: SR_BF16 (bf16 path), SR_BF16X3 (fp32 tensors inside a precision="fp32x3" forward), -1 (exact fp32)."""
    from ..runtime import x3_active

    return L.SR_BF16 if cdt == torch.bfloat16 else (L.SR_BF16X3 if x3_active() else -1)


def swin_qkv_usable(p: Dict, geo: SwinGeometry, Cp: int, cdt: torch.dtype) -> bool:
    """SR_SWIN_QKV=0 keeps the QKV GEMM (A/B switch, read per call)."""
    want = stream_compute_dtype(cdt)
    return (p.get("qkv_dtype", -2) == want and knob("SR_SWIN_QKV", "1") != "0" and ops.swin_qkv_supported(geo.C, Cp, geo.heads, geo.hd_p, geo.ws, want))


def qkv_frag_order(p: Dict, geo: SwinGeometry, Cp: int, cdt: torch.dtype) -> bool:
    """q / k / v^T between sr_swin_qkv (or sr_swin_tail's fused QKV stage) and sr_window_attention in FRAGMENT order (every operand fragment of the attention
    kernel = one coalesced 1-KiB load): 16 x 16 windows, bf16, stream-form producer.  SR_QKV_FRAG=0 keeps the row-major layouts."""
    return geo.ntok == 256 and geo.hd_p == 32 and cdt == torch.bfloat16 and swin_qkv_usable(p, geo, Cp, cdt) and knob("SR_QKV_FRAG", "1") != "0"


def swin_tail_usable(p: Dict, geo: SwinGeometry, Cp: int, cdt: torch.dtype) -> bool:
    """SR_SWIN_TAIL=0 keeps the projection GEMM + MLP kernel (A/B switch, read per call)."""
    want = stream_compute_dtype(cdt)
    return (p.get("tail_dtype", -2) == want and knob("SR_SWIN_TAIL", "1") != "0"
            and ops.swin_tail_supported(geo.C, Cp, geo.heads, geo.hd_p, geo.ws, geo.hid_p, want))


def run_swin_tail(p: Dict, geo: SwinGeometry, o: Tensor, skip: Tensor, t_out: Tensor, shift: int, y_mode: int = L.Y_ROLL, extra: Optional[Dict] = None) -> None:
    """t_out = x1 + MLP(LayerNorm2(x1)),  x1 = skip + proj(o) (+ gated second residual from `extra`, the sr_gemm skip2 fields;
    + the LayerNorm side output n1 = LN(t_out) * gamma + beta when extra carries n1 / n1_ln)."""
    B, H, W, Cp = skip.shape
    kw = {}
    extra = dict(extra or {})
    n1, n1_ln = extra.pop("n1", None), extra.pop("n1_ln", None)
    kw.update(extra.pop("ca", None) or {})  # in-kernel channel-attention gate (pool partials + squeeze weights)
    wg_tokens = int(extra.pop("wg_tokens", 0))  # 32 / 64 / 0 = the launcher's own rule (the caller may know that other part batches share the chip)
    qkv_next = extra.pop("qkv_next", None)    # the next block's LayerNorm1 + QKV as the kernel's last stage (stream: tail + that block's QKV slots)
    wstream = p["tail_stream"]
    if qkv_next is not None:
        wstream = qkv_next.get("stream", None)
        if wstream is None:
            wstream = p["tail_qkv_stream"]
        kw.update(q2=qkv_next["q"].data_ptr(), k2=qkv_next["k"].data_ptr(), vt2=qkv_next["vt"].data_ptr(), shift2=int(qkv_next["shift"]),
                  frag_order=int(bool(qkv_next.get("frag"))), oca_pad2=int(qkv_next.get("oca_pad", 0)))
    x3 = p["tail_dtype"] == L.SR_BF16X3  # split operands: o, y and the LayerNorm side output are fp32 tensors
    if n1 is not None:
        assert n1.dtype == (torch.float32 if x3 else torch.bfloat16) and n1.shape == skip.shape
        kw.update(n1=n1.data_ptr(), n1_gamma=n1_ln[0].data_ptr(), n1_beta=n1_ln[1].data_ptr(), ldn=Cp)
    if extra:
        assert extra["skip2_dtype"] == (L.SR_F32 if x3 else L.SR_BF16) and extra["gate_rows"] == H * W
        kw.update(y=extra["skip2"], gate=extra["skip2_gate"], ldy=extra["ldskip2"], ld_gate=extra["ld_gate"])
    ops.swin_tail(
        x=skip.data_ptr(), out=t_out.data_ptr(), o=o.data_ptr(), wstream=wstream.data_ptr(), bproj=p["proj_b"].data_ptr(), B=B, H=H, W=W,
        C=geo.C, Cp=Cp, ldx=Cp, heads=geo.heads, hd_p=geo.hd_p, ws=geo.ws, shift=shift, Hp=geo.hid_p, eps=1e-5, y_mode=y_mode,
        compute_dtype=p["tail_dtype"], wg_tokens=0 if x3 else wg_tokens, **kw,
    )


def run_swin_block(p: Dict, geo: SwinGeometry, t_in: Tensor, t_out: Tensor, ws_, cdt: torch.dtype, shift: int, y_mode: int = L.Y_ROLL) -> None:
    """t_out = SwinTransformerBlock(t_in) (swinir.py:146-174): ONE launch when a fused kernel covers the geometry
    (attention half + MLP half on the same window), otherwise attention and MLP as separate launches."""
    B, H, W, Cp = t_in.shape
    sdt = sr_dtype(cdt)
    from ..runtime import x3_active

    # the stream was packed for bf16 (cdt bf16) or for split operands (cdt fp32 inside a precision="fp32x3" forward)
    want = L.SR_BF16 if cdt == torch.bfloat16 else (L.SR_BF16X3 if x3_active() else -1)
    if p.get("stream_dtype", -2) == want and ops.swin_block_supported(geo.C, Cp, geo.heads, geo.hd_p, geo.ws, geo.hid_p, want):
        ops.swin_block(
            x=t_in.data_ptr(), out=t_out.data_ptr(), wstream=p["stream"].data_ptr(), bias=p["bias_frag_l2"].data_ptr(), B=B, H=H, W=W, C=geo.C,
            Cp=Cp, ldx=Cp, heads=geo.heads, hd_p=geo.hd_p, ws=geo.ws, shift=shift, Hp=geo.hid_p, eps=1e-5, y_mode=y_mode, compute_dtype=want,
            max_workgroups=int(knob("SR_BLOCK_WGS", "0")),  # 0: one workgroup per window; N / -2: persistent workgroups (SrSwinBlock.max_workgroups)
        )
        return
    if p.get("light_dtype", -2) == want and knob("SR_SWIN_LIGHT", "1") != "0" and ops.swin_light_supported(geo.C, Cp, geo.heads, geo.hd, geo.ws, geo.hidden, want):
        wqkv, bqkv, wproj, bproj, fc1, fb1, fc2, fb2 = p["light"]
        ops.swin_light(x=t_in.data_ptr(), out=t_out.data_ptr(), wqkv=wqkv.data_ptr(), bqkv=bqkv.data_ptr(), wproj=wproj.data_ptr(), bproj=bproj.data_ptr(), w1=fc1.data_ptr(),
                       b1=fb1.data_ptr(), w2=fc2.data_ptr(), b2=fb2.data_ptr(), bias=p["bias_frag"].data_ptr(), B=B, H=H, W=W, C=geo.C, ldx=Cp, shift=shift, eps=1e-5, y_mode=y_mode, compute_dtype=want)
        return
    run_window_msa(p, p["ln1"], geo, t_in, t_out, t_in, ws_, cdt, shift, y_mode=y_mode)
    run_mlp(p, p["ln2"], geo, t_out, ws_, cdt)


def run_window_msa(p: Dict, ln, geo: SwinGeometry, t_in: Tensor, t_out: Tensor, skip: Tensor, ws_, cdt: torch.dtype, shift: int, name: str = "msa",
                   y_mode: int = L.Y_ROLL, before_proj=None, with_mlp: bool = False, qkv_ready: bool = False, attn_launch=None, qkv_n1=None, qkv_in_attn: bool = False):
    """t_out = skip + proj(attention(qkv(LN(t_in))))  with window partition / shift folded into addressing.
    t_in, t_out, skip: fp32 [B, H, W, Cp] (t_out may alias skip).
    before_proj (HAT): called right before the projection GEMM of the un-fused path; returns extra sr_gemm fields for it (the gated
    second residual).  Returns True iff it was used (the one-kernel attention half has no hook).
    with_mlp: the caller's next step is run_mlp(p, ...) on t_out; when sr_swin_tail covers the geometry the projection AND that MLP run as
    one launch and the function returns "tail" (the caller must then skip run_mlp).
    qkv_ready: q / k / v^T of this block are already in the workspace (written by the previous block's sr_swin_tail): no QKV launch.
    attn_launch (HAT): called with the fields of ops.window_attention INSTEAD of that launch (sr_hab_mid: attention + CAB as one launch).
    qkv_n1 (HAT): (n1, gamma, beta): sr_swin_qkv also writes LayerNorm(t_in) * gamma + beta to n1 (the caller guarantees that sr_swin_qkv runs).
    qkv_in_attn (HAT): no QKV launch at all -- the attention workgroups project their own head from t_in (SrWindowAttn.x / wqkv, csrc/sr_wattn_qkv_body.h)."""
    B, H, W, Cp = t_in.shape
    M = B * H * W
    nb = M // geo.ntok
    sdt = sr_dtype(cdt)
    q = ws_.get(name + ".q", (nb, geo.heads, geo.ntok, geo.hd_p), cdt)
    k = ws_.get(name + ".k", (nb, geo.heads, geo.ntok, geo.hd_p), cdt)
    vt = ws_.get(name + ".vt", (nb, geo.heads, geo.hd_p, geo.ntok), cdt)
    o = ws_.get(name + ".o", (M, geo.HP), cdt)
    frag = qkv_frag_order(p, geo, Cp, cdt)  # (a qkv_ready producer decides with the same predicate)
    assert qkv_n1 is None or (not qkv_ready and swin_qkv_usable(p, geo, Cp, cdt))
    if qkv_ready or qkv_in_attn:
        pass
    elif swin_qkv_usable(p, geo, Cp, cdt):
        ops.swin_qkv(x=t_in.data_ptr(), q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), wstream=p["qkv_stream"].data_ptr(), B=B, H=H, W=W, C=geo.C,
                     Cp=Cp, ldx=Cp, heads=geo.heads, hd_p=geo.hd_p, ws=geo.ws, shift=shift, eps=1e-5, y_mode=y_mode, compute_dtype=p["qkv_dtype"],
                     frag_order=int(frag), **({} if qkv_n1 is None else dict(n1=qkv_n1[0].data_ptr(), n1_gamma=qkv_n1[1].data_ptr(), n1_beta=qkv_n1[2].data_ptr(), ldn=Cp)))
    else:
        ops.gemm(
            A=t_in.data_ptr(), Wp=p["qkv_w"].data_ptr(), bias=p["qkv_b"].data_ptr(), ln_gamma=None if fold_ln(cdt) else ln[0].data_ptr(),
            ln_beta=None if fold_ln(cdt) else ln[1].data_ptr(), ln_norm_only=int(fold_ln(cdt)), out=q.data_ptr(), out_k=k.data_ptr(), out_vt=vt.data_ptr(), M=M, K=Cp, N=3 * geo.HP, k_real=geo.C, lda=Cp,
            a_dtype=L.SR_F32, out_dtype=sdt, compute_dtype=sdt, act=L.ACT_NONE, out_scale=1.0, a_map=L.MAP_WINDOW, o_map=L.MAP_IDENTITY,
            H=H, W=W, ws=geo.ws, shift=shift, epi=L.EPI_QKV, heads=geo.heads, hd_p=geo.hd_p, ntok=geo.ntok, ln_eps=1e-5, y_mode=y_mode,
        )
    akw = dict(q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), bias=p["bias"].data_ptr(), out=o.data_ptr(), n_bwin=nb, heads=geo.heads,
               hd_p=geo.hd_p, ntok=geo.ntok, H=H, W=W, ws=geo.ws, shift=shift, dtype=sdt, y_mode=y_mode, bias_frag=p["bias_frag"].data_ptr(),
               qkv_frag=int(frag))
    if "bias_tiles" in p and sdt == L.SR_BF16 and knob("SR_ATTN_LDS", "1") != "0":  # K / V^T / distinct bias tiles staged in LDS once per (window, head)
        akw["bias_tiles"] = p["bias_tiles"].data_ptr()
    if qkv_in_attn:
        akw.update(x=t_in.data_ptr(), wqkv=p["qkv_stream"].data_ptr(), ldx=Cp, C=geo.C, eps=1e-5, qkv_frag=0)
    if attn_launch is not None:
        attn_launch(akw)
    else:
        ops.window_attention(**akw)
    extra = before_proj() if before_proj is not None else {}
    if with_mlp and swin_tail_usable(p, geo, Cp, cdt):
        run_swin_tail(p, geo, o, skip, t_out, shift, y_mode, extra)
        return "tail"
    for k_ in ("n1", "n1_ln", "ca", "qkv_next", "wg_tokens"):  # the LayerNorm side output / in-kernel gate / fused next QKV exist only in sr_swin_tail
        extra.pop(k_, None)
    ops.gemm(
        A=o.data_ptr(), Wp=p["proj_w"].data_ptr(), bias=p["proj_b"].data_ptr(), out=t_out.data_ptr(), skip=skip.data_ptr(),
        M=M, K=geo.HP, N=Cp, lda=geo.HP, ldo=Cp, ldskip=Cp, a_dtype=sdt, out_dtype=L.SR_F32, compute_dtype=sdt, act=L.ACT_NONE,
        out_scale=1.0, a_map=L.MAP_IDENTITY, o_map=L.MAP_WINDOW, H=H, W=W, ws=geo.ws, shift=shift, epi=L.EPI_STD, y_mode=y_mode, **extra,
    )
    return before_proj is not None


def run_mlp(p: Dict, ln, geo: SwinGeometry, t: Tensor, ws_, cdt: torch.dtype, name: str = "mlp") -> None:
    """t += fc2(GELU(fc1(LN(t))))  in place on the fp32 stream [.., Cp]."""
    Cp = t.shape[-1]
    M = t.numel() // Cp
    sdt = sr_dtype(cdt)
    if ops.mlp_fused_supported(Cp, geo.hid_p, sdt):  # one kernel, hidden activations stay on the CU
        ops.mlp_fused(
            x=t.data_ptr(), out=t.data_ptr(), ln_gamma=None, ln_beta=None, w1p=p["fc1_w"].data_ptr(),
            b1=p["fc1_b"].data_ptr(), w2p=p["fc2_w"].data_ptr(), b2=p["fc2_b"].data_ptr(), M=M, C=geo.C, Cp=Cp, Hp=geo.hid_p, ldx=Cp, eps=1e-5,
        )
        return
    h = ws_.get(name + ".h", (M, geo.hid_p), cdt)
    ops.gemm(
        A=t.data_ptr(), Wp=p["fc1_w"].data_ptr(), bias=p["fc1_b"].data_ptr(), ln_gamma=None if fold_ln(cdt) else ln[0].data_ptr(),
        ln_beta=None if fold_ln(cdt) else ln[1].data_ptr(), ln_norm_only=int(fold_ln(cdt)), out=h.data_ptr(), M=M, K=Cp, N=geo.hid_p, k_real=geo.C, lda=Cp, ldo=geo.hid_p, a_dtype=L.SR_F32, out_dtype=sdt,
        compute_dtype=sdt, act=L.ACT_GELU, out_scale=1.0, epi=L.EPI_STD, ln_eps=1e-5,
    )
    ops.gemm(
        A=h.data_ptr(), Wp=p["fc2_w"].data_ptr(), bias=p["fc2_b"].data_ptr(), out=t.data_ptr(), skip=t.data_ptr(), M=M, K=geo.hid_p,
        N=Cp, lda=geo.hid_p, ldo=Cp, ldskip=Cp, a_dtype=sdt, out_dtype=L.SR_F32, compute_dtype=sdt, act=L.ACT_NONE, out_scale=1.0,
        epi=L.EPI_STD,
    )


def final_affine(img_range: float, n_colors: int, device) -> tuple:
    """Normalizer.unnormalize (common.py:232-233): (x + mean) * range = x*range + mean*range."""
    mean = torch.tensor(RGB_MEAN[:n_colors], dtype=torch.float32, device=device)
    return torch.full((n_colors,), float(img_range), dtype=torch.float32, device=device), (mean * img_range).contiguous()


def ingest_affine(img_range: float, n_colors: int, device) -> tuple:
    """Normalizer.normalize (common.py:228-230): x / range - mean."""
    mean = torch.tensor(RGB_MEAN[:n_colors], dtype=torch.float32, device=device)
    return torch.full((n_colors,), 1.0 / float(img_range), dtype=torch.float32, device=device), (-mean).contiguous()


def direct_cps_p(c_ps: int, r: int) -> int:
    cps = packing.round_up(c_ps, 4)
    while (r * r * cps) % 16:
        cps += 4
    return cps


# --------------------------------------------------------------------------- the model
class SwinIR(Model):
    part_batches = 0  # inside a HIP-graph capture, run a batch as this many part batches on the model's own streams (0: two from 16 tiles on; see forward)

    def __init__(
        self,
        scale: int = 4,
        n_colors: int = 3,
        img_range: float = 1.0,
        embed_dim: int = 180,
        depths: List[int] = [6, 6, 6, 6, 6, 6],
        num_heads: List[int] = [6, 6, 6, 6, 6, 6],
        window_size: int = 8,
        mlp_ratio: float = 2.0,
        drop_rate: float = 0.0,
        attn_drop_rate: float = 0.0,
        drop_path_rate: float = 0.1,
        upsampler: str = "pixelshuffle",
        resi_connection: Optional[nn.Module] = None,
    ) -> None:
        super().__init__(scale, n_colors, img_range)
        if resi_connection is not None:
            raise NotImplementedError("custom resi_connection modules are outside the HIP hot path")
        assert n_colors == 3, "Normalizer mean has 3 channels (common.py:223)"
        self.embed_dim = embed_dim
        self.depths = depths
        self.num_heads = num_heads
        self.window_size = window_size
        self.mlp_ratio = mlp_ratio
        self.drop_rate = drop_rate
        self.attn_drop_rate = attn_drop_rate
        self.drop_path_rate = drop_path_rate
        self.upsampler = upsampler

        self.conv_first = nn.Conv2d(n_colors, embed_dim, 3, 1, 1)
        self.patch_embed = PatchEmbed(embed_dim)
        self.layers = nn.ModuleList([RSTB(embed_dim, depths[i], num_heads[i], window_size, mlp_ratio) for i in range(len(depths))])
        self.norm = nn.LayerNorm(embed_dim)
        self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1)
        if upsampler == "pixelshuffle":
            num_feat = 64
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.upsample = Upsampler(scale, num_feat)
            self.conv_last = nn.Conv2d(num_feat, n_colors, 3, 1, 1)
        elif upsampler == "pixelshuffledirect":
            self.upsample = Upsampler(scale, embed_dim, n_colors)
        self.apply(self._init_weights)

    def _init_weights(self, m: nn.Module) -> None:  # swinir.py:333-340
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ packing
    def _geo(self, li: int) -> SwinGeometry:
        return SwinGeometry(self.embed_dim, self.num_heads[li], self.window_size, int(self.embed_dim * self.mlp_ratio))

    def _pack(self, dt: torch.dtype) -> Dict:
        C = self.embed_dim
        geo0 = self._geo(0)
        Cp = geo0.Cp
        dev = self.conv_first.weight.device
        P: Dict = {}
        P["first"] = packing.pack_conv3x3(self.conv_first.weight, self.conv_first.bias, 32, packing.identity_idx(C, Cp), dt)
        P["pe_norm"] = pack_ln(self.patch_embed.norm, Cp)
        P["layers"] = []
        for li, layer in enumerate(self.layers):
            geo = self._geo(li)
            blocks = []
            for blk in layer.residual_group.blocks:
                e = dict(shift=blk.shift_size, ln1=pack_ln(blk.norm1, Cp), ln2=pack_ln(blk.norm2, Cp))
                e.update(pack_attention(blk.attn, geo, dt, norm=blk.norm1))
                e.update(pack_mlp(blk.mlp, geo, dt, norm=blk.norm2))
                e.update(pack_block_stream(blk, geo, dt))
                e.update(pack_light_block(blk, geo, dt))
                blocks.append(e)
            P["layers"].append(dict(blocks=blocks, conv=self._pack_resi(layer.conv, C, Cp, dt), geo=geo))
        P["norm"] = pack_ln(self.norm, Cp)
        P["after_body"] = self._pack_resi(self.conv_after_body, C, Cp, dt)
        P["fin"] = final_affine(self.img_range, self.n_colors, dev)
        P["ing"] = ingest_affine(self.img_range, self.n_colors, dev)
        if self.upsampler == "pixelshuffle":
            cbu = self.conv_before_upsample[0]
            P["before_up"] = packing.pack_conv3x3(cbu.weight, cbu.bias, Cp, packing.identity_idx(64, 64), dt)
            P["up"] = pack_upsampler(self.upsample, 64, dt)
            P["last"] = packing.pack_conv3x3(self.conv_last.weight, self.conv_last.bias, 64, packing.identity_idx(self.n_colors, 16), dt)
        elif self.upsampler == "pixelshuffledirect":
            P["up"] = pack_upsampler(self.upsample, Cp, dt, last_cps_p=direct_cps_p(self.n_colors, self.scale))
        return P

    # ------------------------------------------------------------------ residual connection of an RSTB / conv_after_body (swinir.py:241,316)
    def _pack_resi(self, m: nn.Module, C: int, Cp: int, dt: torch.dtype):
        return packing.pack_conv3x3(m.weight, m.bias, Cp, packing.identity_idx(C, Cp), dt)

    def _run_resi(self, packed, src: Tensor, dst: Tensor, skip: Tensor, cdt: torch.dtype) -> None:
        """dst = resi(src) + skip on padded NHWC buffers (dst may be skip).  SwinIR: one 3x3 conv launch; SwinFIR overrides it with its SFB."""
        conv_call(src, *packed, dst, cdt, skip=skip)

    # ------------------------------------------------------------------ forward
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
        # Inside a HIP-graph capture a batch can run as part batches on several streams: the parts' launches are out of phase, so one part's
        # x-fetch / store / convolution phases run under the other's MFMAs (what bench.py's two batches in flight do across steps).  Eager
        # forwards stay one launch sequence (they are launch-bound).  `part_batches` (attribute; SR_SWIN_PARTS overrides): 1 = off (default).  It is a LATENCY knob:
        # one batch of 8 alone on the GPU 1.77 -> 1.70 (2 parts) / 1.65 ms (4); with a second batch in flight (bench.py's throughput leg) 1.44 -> 1.67 / 1.60 ms.
        # part_batches = 0 (default): two half batches from 16 tiles on (a lone batch: b16 3.30 -> 2.90 ms, b32 6.10 -> 5.66; lightweight geometry b32 1.85 -> 1.65)
        parts = int(os.environ.get("SR_SWIN_PARTS", "0")) or int(getattr(self, "part_batches", 0) or 0) or (2 if B >= 16 else 1)
        if parts > 1 and B % parts == 0 and B // parts >= 2 and x.is_cuda and capturing_or_warming_up():
            from ..runtime import WorkspaceView

            main = torch.cuda.current_stream(x.device)
            h = B // parts
            sides = [self._part_stream(x.device, i) for i in range(parts - 1)]
            for i, side in enumerate(sides):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._forward_into(P, x[(i + 1) * h:(i + 2) * h], out[(i + 1) * h:(i + 2) * h], WorkspaceView(ws_, f"p{i + 1}."), cdt)
            self._forward_into(P, x[:h], out[:h], WorkspaceView(ws_, "p0."), cdt)
            for side in sides:
                main.wait_stream(side)
        else:
            self._forward_into(P, x, out, ws_, cdt)
        return out

    def _part_stream(self, device, i: int) -> "torch.cuda.Stream":
        sts = getattr(self, "_part_streams", None)
        if not isinstance(sts, dict) or sts.get("device") != torch.device(device):
            sts = {"device": torch.device(device)}
            object.__setattr__(self, "_part_streams", sts)
        if i not in sts:
            sts[i] = torch.cuda.Stream(device=device)
        return sts[i]

    def _forward_into(self, P: Dict, x: Tensor, out: Tensor, ws_, cdt) -> None:
        B, _, H, W = x.shape
        w8 = self.window_size
        if self.training:  # check_image_size: reflect pad to the next multiple (swinir.py:356, common.py:277-282)
            Hp, Wp, pad_mode = H + (w8 - H % w8) % w8, W + (w8 - W % w8) % w8, L.PAD_REFLECT
            if Hp - H >= H or Wp - W >= W:
                raise RuntimeError("Padding size should be less than the corresponding input dimension (reflect pad)")
        else:  # check_image_size_for_eval: always adds 1..ws rows/cols (swinir.py:249-255)
            Hp, Wp, pad_mode = (H // w8 + 1) * w8, (W // w8 + 1) * w8, L.PAD_EVAL_MIRROR
        Cp = P["layers"][0]["geo"].Cp if P["layers"] else packing.round_up(self.embed_dim, 64)

        xin = ws_.get("xin", (B, Hp, Wp, 32), cdt)
        ops.ingest_nchw(x, xin, pad_mode, *P["ing"])
        first = ws_.get("first", (B, Hp, Wp, Cp), torch.float32)
        conv_call(xin, *P["first"], first, cdt)
        ta = ws_.get("ta", (B, Hp, Wp, Cp), torch.float32)
        tb = ws_.get("tb", (B, Hp, Wp, Cp), torch.float32)
        ops.layernorm(first, ta, *P["pe_norm"], self.embed_dim)

        for lp in P["layers"]:
            geo = lp["geo"]
            cur = ta  # RSTB input stays in ta until the closing conv has consumed it as the skip
            for bp in lp["blocks"]:
                run_swin_block(bp, geo, cur, tb, ws_, cdt, bp["shift"])
                cur = tb
            # ta = conv(cur) + ta   (swinir.py:245-246).  Within an RSTB the first block reads `ta` and
            # writes `tb`; later blocks work in place on `tb`; a zero-depth RSTB convolves `ta` itself.
            if cur is ta:
                cur = ws_.get("tc", (B, Hp, Wp, Cp), torch.float32)
                cur.copy_(ta)
            self._run_resi(lp["conv"], cur, ta, ta, cdt)
        normed = ws_.get("normed", (B, Hp, Wp, Cp), cdt)  # read only by the conv, which rounds to the compute dtype anyway
        ops.layernorm(ta, normed, *P["norm"], self.embed_dim)
        body = ws_.get("body", (B, Hp, Wp, Cp), cdt)
        self._run_resi(P["after_body"], normed, body, first, cdt)  # conv_after_body(features) + x  (swinir.py:362)

        s = self.scale
        fin = (*P["fin"], self.n_colors, H * s, W * s)
        if self.upsampler == "pixelshuffle":
            feat = ws_.get("feat", (B, Hp, Wp, 64), cdt)
            conv_call(body, *P["before_up"], feat, cdt, act=L.ACT_LRELU)
            up = run_upsampler(P["up"], feat, ws_, cdt, "swin")
            conv_call(up, *P["last"], out, cdt, out_mode=L.OUT_FINAL_NCHW, fin=fin, cout_p=16)
        else:
            wp, b, r, cps_p = P["up"][0]
            conv_call(body, wp, b, out, cdt, out_mode=L.OUT_FINAL_NCHW, ps_r=r, cps_p=cps_p, fin=fin, cout_p=r * r * cps_p)

    def forward_strips(self, x: Tensor, comm) -> Tensor:
        """forward() of ONE image, row-strip sharded with per-layer halo exchange (studiosr_amd/strips.py; SURVEY.md
        section 8e, config 4).  `comm` is a strips.DistStripComm (one strip per process / GPU) or a strips.LocalStripComm."""
        from ..runtime import x3_mode
        from ..strips import swinir_forward_strips

        with x3_mode(self.precision == "fp32x3"):
            return swinir_forward_strips(self, x, comm)

    # ------------------------------------------------------------------ reference API
    def get_model_config(self) -> Dict:
        config = super().get_model_config()
        config.update(
            dict(
                embed_dim=self.embed_dim,
                depths=self.depths,
                num_heads=self.num_heads,
                window_size=self.window_size,
                mlp_ratio=self.mlp_ratio,
                drop_rate=self.drop_rate,
                attn_drop_rate=self.attn_drop_rate,
                drop_path_rate=self.drop_path_rate,
                upsampler=self.upsampler,
            )
        )
        return config

    def get_training_config(self) -> Dict:  # swinir.py:391-402
        return dict(
            batch_size=32,
            learning_rate=0.0002,
            beta1=0.9,
            beta2=0.99,
            weight_decay=0.0,
            max_iters=500000,
            gamma=0.5,
            milestones=[250000, 400000, 450000, 475000],
        )

    @classmethod
    def from_pretrained(cls, scale: int = 4, light: bool = False, dataset: str = "DF2K", pretrained: bool = True) -> "SwinIR":
        """Same configurations and checkpoint file names as swinir.py:404-445; weights are read from
        ./pretrained/<file> (the GPU box has no network, so a missing file is an error, not a download)."""
        assert scale in [2, 3, 4, 8]
        assert dataset in ["DIV2K", "DF2K"]
        config: Dict = {"scale": scale}
        img_size = 64 if dataset == "DF2K" else 48
        task, label = "001_classicalSR", "M"
        if light:
            config.update(depths=[6, 6, 6, 6], embed_dim=60, num_heads=[6, 6, 6, 6], upsampler="pixelshuffledirect")
            task, dataset, img_size, label = "002_lightweightSR", "DIV2K", 64, "S"
        model = cls(**config)
        if pretrained:
            path = os.path.join("pretrained", f"{task}_{dataset}_s{img_size}w8_SwinIR-{label}_x{scale}.pth")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found (no network access here; place the official checkpoint there)")
            ckpt = torch.load(path, map_location="cpu")
            model.load_state_dict(ckpt["params"] if "params" in ckpt else ckpt, strict=False)
        return model
