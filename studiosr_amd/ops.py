"""Python faces of the C-ABI entry points (tensor -> raw pointers).  No math happens here."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

Tensor = torch.Tensor


def _dt(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return L.SR_F32
    if t.dtype == torch.bfloat16:
        return L.SR_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t: Optional[Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.HipLibraryError("studiosr_amd ops need ROCm device tensors (there is no CPU path)")
    assert t.is_contiguous()
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def ingest_nchw(x: Tensor, out: Tensor, pad_mode: int, scale: Tensor, bias: Tensor) -> Tensor:
    B, Cc, H, W = x.shape
    _, Hp, Wp, Cp = out.shape
    assert x.dtype == torch.float32
    L.check(L.lib().sr_ingest_nchw(_p(x), _p(out), _dt(out), B, Cc, H, W, Hp, Wp, Cp, pad_mode, _p(scale), _p(bias), _stream()), "sr_ingest_nchw")
    return out


def layernorm(x: Tensor, out: Tensor, gamma: Tensor, beta: Tensor, C_real: int, eps: float = 1e-5) -> Tensor:
    Cp = x.shape[-1]
    M = x.numel() // Cp
    a = L.SrLayernorm()  # (the argument-block form: recordable in a launch plan)
    a.x, a.y, a.gamma, a.beta, a.y_dtype, a.M, a.C, a.Cp, a.eps = _p(x), _p(out), _p(gamma), _p(beta), _dt(out), M, C_real, Cp, eps
    L.check(L.lib().sr_layernorm_to_args(C.byref(a), _stream()), "sr_layernorm")
    return out


def _x3(kw) -> None:
    from .runtime import x3_active

    if kw.get("compute_dtype", L.SR_F32) == L.SR_F32 and x3_active():
        kw["compute_dtype"] = L.SR_BF16X3


def gemm(**kw) -> None:
    _x3(kw)
    g = L.SrGemm()
    for k, v in kw.items():
        setattr(g, k, v)
    L.check(L.lib().sr_gemm(C.byref(g), _stream()), "sr_gemm")


def swin_light_supported(C_: int, Cp: int, heads: int, hd: int, ws: int, hidden: int, compute_dtype: int) -> bool:
    return bool(L.lib().sr_swin_light_supported(C_, Cp, heads, hd, ws, hidden, compute_dtype))


def swin_light(**kw) -> None:
    """The whole SwinTransformerBlock of the lightweight geometry (embed 60, 6 heads, window 8, hidden 120) in one launch (ABI v7)."""
    a = L.SrSwinLight()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_swin_light(C.byref(a), _stream()), "sr_swin_light")


def swin_block_supported(C_: int, Cp: int, heads: int, hd_p: int, ws: int, Hp: int, compute_dtype: int) -> bool:
    return bool(L.lib().sr_swin_block_supported(C_, Cp, heads, hd_p, ws, Hp, compute_dtype))


def swin_block(**kw) -> None:
    """The whole SwinTransformerBlock in one launch from one packed weight stream (ABI v5; swinir.py:146-174)."""
    a = L.SrSwinBlock()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_swin_block(C.byref(a), _stream()), "sr_swin_block")


def cab_supported(cin_p: int, cmid_p: int, cout_p: int, dtype: int) -> bool:
    return bool(L.lib().sr_cab_supported(cin_p, cmid_p, cout_p, dtype))


def cab_pool_tiles(H: int, W: int) -> int:
    return int(L.lib().sr_cab_pool_tiles(H, W))


def cab_pool_tiles_rows(H: int, W: int, tile_rows: int) -> int:
    """Pool slots per image of sr_hab_mid's CAB role for SrCab.tile_rows (0 / 6 / 8; ABI v9)."""
    return int(L.lib().sr_cab_pool_tiles_rows(H, W, tile_rows))


def cab_fused(**kw) -> None:
    """conv -> GELU -> conv of HAT's CAB in one launch, with the pool partials of the channel-attention squeeze (ABI v6; hat.py:41-49)."""
    a = L.SrCab()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_cab_fused(C.byref(a), _stream()), "sr_cab_fused")


def swin_qkv_supported(C_: int, Cp: int, heads: int, hd_p: int, ws: int, compute_dtype: int) -> bool:
    return bool(L.lib().sr_swin_qkv_supported(C_, Cp, heads, hd_p, ws, compute_dtype))


def swin_qkv(**kw) -> None:
    """LayerNorm1 + QKV projection in front of sr_window_attention from one packed weight stream (ABI v6; hat.py:164-176)."""
    a = L.SrSwinQkv()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_swin_qkv(C.byref(a), _stream()), "sr_swin_qkv")


def swin_tail_supported(C_: int, Cp: int, heads: int, hd_p: int, ws: int, Hp: int, compute_dtype: int) -> bool:
    return bool(L.lib().sr_swin_tail_supported(C_, Cp, heads, hd_p, ws, Hp, compute_dtype))


def swin_tail(**kw) -> None:
    """Projection + shortcut (+ gated second residual) + LayerNorm2 + MLP behind an attention kernel in one launch (ABI v6; hat.py:172-194)."""
    a = L.SrSwinTail()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_swin_tail(C.byref(a), _stream()), "sr_swin_tail")


def mlp_fused_supported(Cp: int, Hp: int, compute_dtype: int) -> bool:
    return bool(L.lib().sr_mlp_fused_supported(Cp, Hp, compute_dtype))


def mlp_fused(**kw) -> None:
    a = L.SrMlp()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_mlp_fused(C.byref(a), _stream()), "sr_mlp_fused")


def conv3x3(**kw) -> None:
    _x3(kw)
    c = L.SrConv3x3()
    for k, v in kw.items():
        setattr(c, k, v)
    L.check(L.lib().sr_conv3x3(C.byref(c), _stream()), "sr_conv3x3")


def rcab_conv_pair(**kw) -> None:
    from .runtime import x3_active

    if x3_active():  # precision "fp32x3": split-operand form (ABI v11; fp32 tensors, weights packed hi | lo)
        kw.setdefault("compute_dtype", L.SR_BF16X3)
    a = L.SrRcab()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_rcab_conv_pair(C.byref(a), _stream()), "sr_rcab_conv_pair")


def rcab_pool_tiles(H: int, W: int) -> int:
    return int(L.lib().sr_rcab_pool_tiles(H, W))


def window_attention(**kw) -> None:
    from .runtime import knob, x3_active

    # precision "fp32x3": the flash form has a split-operand instantiation (ABI v11) -- fp32 tensors, three bf16 MFMAs per product instead of eight fp32 ones
    if (kw.get("dtype") == L.SR_F32 and x3_active() and kw.get("bias_frag") and kw.get("hd_p") == 32 and kw.get("ntok") in (64, 256) and kw.get("ws", 0) % 4 == 0
            and not kw.get("qkv_frag") and not kw.get("bias_tiles") and not kw.get("x") and knob("SR_ATTN_X3", "1") != "0"):
        kw["dtype"] = L.SR_BF16X3
    a = L.SrWindowAttn()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_window_attention(C.byref(a), _stream()), "sr_window_attention")


def hab_mid_supported(ntok: int, hd_p: int, ws: int, attn_dtype: int, cin_p: int, cmid_p: int, cout_p: int, cab_dtype: int) -> bool:
    return bool(L.lib().sr_hab_mid_supported(ntok, hd_p, ws, attn_dtype, cin_p, cmid_p, cout_p, cab_dtype))


def hab_mid(attn: dict, cab: dict) -> None:
    """Window attention and CAB body of one HAB as ONE launch (ABI v8; hat.py:165-176): `attn` = the fields of window_attention, `cab` = those of cab_fused."""
    a, c = L.SrWindowAttn(), L.SrCab()
    for k, v in attn.items():
        setattr(a, k, v)
    for k, v in cab.items():
        setattr(c, k, v)
    L.check(L.lib().sr_hab_mid(C.byref(a), C.byref(c), _stream()), "sr_hab_mid")


def oca_attention(**kw) -> None:
    from .runtime import knob, x3_active

    # precision "fp32x3": split-operand instantiation of the flash form (ABI v11), as window_attention
    if (kw.get("dtype") == L.SR_F32 and x3_active() and kw.get("bias_frag") and kw.get("nk_frag", 0) % 64 == 0 and kw.get("nk_frag", 0) > 0 and kw.get("hd_p") == 32
            and (kw.get("ws", 0) ** 2) % 64 == 0 and knob("SR_ATTN_X3", "1") != "0"):
        kw["dtype"] = L.SR_BF16X3
    a = L.SrOcaAttn()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_oca_attention(C.byref(a), _stream()), "sr_oca_attention")


def channel_attention(**kw) -> None:
    a = L.SrChannelAttn()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_channel_attention(C.byref(a), _stream()), "sr_channel_attention")


def channel_gate(gate: Tensor, **kw) -> None:
    a = L.SrChannelAttn()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.lib().sr_channel_gate(C.byref(a), gate.data_ptr(), _stream()), "sr_channel_gate")


def pixel_shuffle(x: Tensor, r: int) -> Tensor:
    """nn.PixelShuffle(r) on an NCHW device tensor (bit-exact copy kernel)."""
    B, Cin, H, W = x.shape
    assert Cin % (r * r) == 0 and x.element_size() in (2, 4)
    x = x.contiguous()
    out = torch.empty(B, Cin // (r * r), H * r, W * r, dtype=x.dtype, device=x.device)
    L.check(L.lib().sr_pixel_shuffle_nchw(_p(x), _p(out), x.element_size(), B, Cin // (r * r), H, W, r, _stream()), "sr_pixel_shuffle_nchw")
    return out


def u8_to_nchw(img: Tensor, divisor: float) -> Tensor:
    """uint8 [B,H,W,C] -> fp32 [B,C,H,W] = u8 / divisor (Model.inference front end, common.py:42-43)."""
    assert img.dtype == torch.uint8 and img.dim() == 4 and img.is_contiguous()
    B, H, W, Cc = img.shape
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=img.device)
    L.check(L.lib().sr_u8_to_nchw(_p(img), _p(out), B, Cc, H, W, float(divisor), _stream()), "sr_u8_to_nchw")
    return out


def nchw_to_u8(x: Tensor, mult: float) -> Tensor:
    """fp32 [B,C,H,W] -> uint8 [B,H,W,C] = clip(round_half_even(x * mult), 0, 255) (common.py:44-45)."""
    assert x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty(B, H, W, Cc, dtype=torch.uint8, device=x.device)
    L.check(L.lib().sr_nchw_to_u8(_p(x), _p(out), B, Cc, H, W, float(mult), _stream()), "sr_nchw_to_u8")
    return out


def conv_pool_tiles(H: int, W: int, cout_p: int, compute_dtype: int, tile_rows: int = 0) -> int:
    if tile_rows:
        return int(L.lib().sr_conv3x3_pool_tiles_rows(H, W, cout_p, tile_rows))
    return int(L.lib().sr_conv3x3_pool_tiles(H, W, cout_p, compute_dtype))


# --------------------------------------------------------------------------- device-side weight packing (sr_pack_*)
def _ip(t: Optional[Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()
    return t.data_ptr()


def pack_matrix(w: Tensor, n_p: int, k_p: int, dtype: torch.dtype, row_idx: Optional[Tensor] = None, col_idx: Optional[Tensor] = None,
                row_scale: Optional[Tensor] = None, col_scale: Optional[Tensor] = None) -> Tensor:
    """nn.Linear weight [rows, cols] fp32 -> MFMA fragments (the device twin of packing.pack_linear's matrix half)."""
    assert w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1
    out = torch.empty(n_p * k_p, dtype=dtype, device=w.device)
    L.check(L.lib().sr_pack_matrix(_p(w) if w.is_contiguous() else w.data_ptr(), w.stride(0), _ip(row_idx), _ip(col_idx), _p(row_scale), _p(col_scale), out.data_ptr(), _dt(out),
                                   n_p, k_p, w.shape[0], w.shape[1], _stream()), "sr_pack_matrix")
    return out


def pack_conv3x3(w: Tensor, cin_p: int, n_p: int, dtype: torch.dtype, row_idx: Optional[Tensor] = None) -> Tensor:
    assert w.dtype == torch.float32 and w.dim() == 4 and w.shape[2:] == (3, 3) and w.is_contiguous()
    out = torch.empty(n_p * 9 * cin_p, dtype=dtype, device=w.device)
    L.check(L.lib().sr_pack_conv3x3(_p(w), _ip(row_idx), out.data_ptr(), _dt(out), n_p, w.shape[0], w.shape[1], cin_p, _stream()), "sr_pack_conv3x3")
    return out


def pack_vector(b: Optional[Tensor], n_p: int, idx: Optional[Tensor] = None, scale: Optional[Tensor] = None, device=None) -> Tensor:
    out = torch.empty(n_p, dtype=torch.float32, device=b.device if b is not None else device)
    L.check(L.lib().sr_pack_vector(_p(b), _ip(idx), _p(scale), out.data_ptr(), n_p, 0 if b is None else b.numel(), _stream()), "sr_pack_vector")
    return out


def pack_bias_fragments(table: Tensor, rpi: Tensor, n_q: int, n_k: int) -> Tensor:
    assert table.dtype == torch.float32 and table.is_contiguous() and rpi.dtype == torch.int64 and rpi.is_contiguous() and rpi.numel() == n_q * n_k
    heads = table.shape[1]
    out = torch.empty(heads * n_q * n_k, dtype=torch.float32, device=table.device)
    L.check(L.lib().sr_pack_bias_fragments(_p(table), rpi.data_ptr(), out.data_ptr(), table.shape[0], heads, n_q, n_k, _stream()), "sr_pack_bias_fragments")
    return out
