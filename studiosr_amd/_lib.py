"""ctypes binding of libstudiosr_hip.so (the C ABI declared in include/studiosr_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails the error is
raised.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SR_LIB_PATH selects an experimental build (tools/variants.sh) without touching the shipped library.
LIB_PATH = os.environ.get("SR_LIB_PATH") or os.path.join(_HERE, "lib", "libstudiosr_hip.so")
CSRC = os.path.join(_HERE, "csrc")

SR_F32, SR_BF16 = 0, 1
SR_BF16X3 = 2  # compute_dtype of sr_gemm / sr_conv3x3 only: split-operand bf16 (hi + lo), fp32 tensors
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_GELU = 0, 1, 2, 3
PAD_NONE, PAD_EVAL_MIRROR, PAD_REFLECT = 0, 1, 2
MAP_IDENTITY, MAP_WINDOW = 0, 1
EPI_STD, EPI_QKV, EPI_QKV_OCA = 0, 1, 2
OUT_NHWC, OUT_PIXEL_SHUFFLE, OUT_FINAL_NCHW = 0, 1, 2
Y_ROLL, Y_STRIP, Y_STRIP_LAST = 0, 1, 2

ABI_VERSION = 11
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float


class SrGemm(C.Structure):
    _fields_ = [
        ("A", _vp), ("Wp", _vp), ("bias", _vp), ("ln_gamma", _vp), ("ln_beta", _vp),
        ("out", _vp), ("out_k", _vp), ("out_vt", _vp), ("skip", _vp),
        ("M", _i), ("K", _i), ("N", _i), ("k_real", _i),
        ("lda", _i), ("ldo", _i), ("ldskip", _i),
        ("a_dtype", _i), ("out_dtype", _i), ("compute_dtype", _i),
        ("act", _i), ("out_scale", _f),
        ("a_map", _i), ("o_map", _i),
        ("H", _i), ("W", _i), ("ws", _i), ("shift", _i),
        ("epi", _i), ("heads", _i), ("hd_p", _i), ("ntok", _i),
        ("ln_eps", _f), ("ln_norm_only", _i), ("oca_pad", _i), ("y_mode", _i),
        ("skip2", _vp), ("skip2_gate", _vp), ("skip2_dtype", _i), ("ldskip2", _i), ("gate_rows", _i), ("ld_gate", _i),
    ]


class SrSwinBlock(C.Structure):
    _fields_ = [
        ("x", _vp), ("out", _vp), ("wstream", _vp), ("bias", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("Cp", _i), ("ldx", _i), ("heads", _i), ("hd_p", _i), ("ws", _i), ("shift", _i), ("Hp", _i),
        ("eps", _f), ("y_mode", _i), ("compute_dtype", _i), ("max_workgroups", _i),
    ]


class SrSwinLight(C.Structure):
    _fields_ = [
        ("x", _vp), ("out", _vp), ("wqkv", _vp), ("bqkv", _vp), ("wproj", _vp), ("bproj", _vp), ("w1", _vp), ("b1", _vp), ("w2", _vp), ("b2", _vp), ("bias", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("ldx", _i), ("shift", _i), ("eps", _f), ("y_mode", _i), ("compute_dtype", _i),
    ]


class SrCab(C.Structure):  # (mid_pre: ABI v8)
    _fields_ = [
        ("x", _vp), ("w1p", _vp), ("b1", _vp), ("w2p", _vp), ("b2", _vp), ("y", _vp), ("pool_partial", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("Cin_p", _i), ("Cmid_p", _i), ("Cout_p", _i), ("dtype", _i),
        ("mid_pre", _vp), ("tile_rows", _i), ("bwd_pre", _vp), ("bwd_dmid", _vp), ("bwd_g", _vp),
    ]


class SrSwinQkv(C.Structure):
    _fields_ = [
        ("x", _vp), ("q", _vp), ("k", _vp), ("vt", _vp), ("wstream", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("Cp", _i), ("ldx", _i), ("heads", _i), ("hd_p", _i), ("ws", _i), ("shift", _i),
        ("eps", _f), ("y_mode", _i), ("compute_dtype", _i), ("frag_order", _i), ("oca_pad", _i),
        ("n1", _vp), ("n1_gamma", _vp), ("n1_beta", _vp), ("ldn", _i),
    ]


class SrSwinTail(C.Structure):
    _fields_ = [
        ("x", _vp), ("out", _vp), ("o", _vp), ("wstream", _vp), ("bproj", _vp), ("y", _vp), ("gate", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("Cp", _i), ("ldx", _i), ("ldy", _i), ("ld_gate", _i), ("heads", _i), ("hd_p", _i), ("ws", _i),
        ("shift", _i), ("Hp", _i), ("eps", _f), ("y_mode", _i), ("compute_dtype", _i),
        ("n1", _vp), ("n1_gamma", _vp), ("n1_beta", _vp), ("ldn", _i),
        ("pool_partial", _vp), ("ca_w1", _vp), ("ca_b1", _vp), ("ca_w2", _vp), ("ca_b2", _vp), ("ca_Cr", _i), ("ca_n_tiles", _i), ("y_scale", _f),
        ("q2", _vp), ("k2", _vp), ("vt2", _vp), ("shift2", _i), ("frag_order", _i), ("oca_pad2", _i), ("wg_tokens", _i),
    ]


class SrMlp(C.Structure):
    _fields_ = [
        ("x", _vp), ("out", _vp), ("ln_gamma", _vp), ("ln_beta", _vp), ("w1p", _vp), ("b1", _vp), ("w2p", _vp), ("b2", _vp),
        ("M", _i), ("C", _i), ("Cp", _i), ("Hp", _i), ("ldx", _i), ("eps", _f),
    ]


class SrConv3x3(C.Structure):
    _fields_ = [
        ("x", _vp), ("Wp", _vp), ("bias", _vp), ("out", _vp), ("skip", _vp),
        ("pool_partial", _vp), ("fin_scale", _vp), ("fin_bias", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("Cin_p", _i), ("Cout_p", _i),
        ("x_dtype", _i), ("out_dtype", _i), ("skip_dtype", _i), ("compute_dtype", _i),
        ("act", _i), ("out_scale", _f), ("out_mode", _i), ("ps_r", _i), ("cps_p", _i),
        ("fin_c", _i), ("fin_h", _i), ("fin_w", _i), ("act_slope", _f), ("tile_rows", _i),
    ]


class SrRcab(C.Structure):
    _fields_ = [
        ("x", _vp), ("w1p", _vp), ("b1", _vp), ("w2p", _vp), ("b2", _vp), ("y", _vp), ("pool_partial", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C_p", _i), ("x_dtype", _i), ("y_dtype", _i),
        ("gate_y", _vp), ("gate_pool", _vp), ("gate_w1", _vp), ("gate_b1", _vp), ("gate_w2", _vp), ("gate_b2", _vp), ("x_out", _vp),
        ("gate_C", _i), ("gate_Cr", _i), ("compute_dtype", _i),
    ]


class SrWindowAttn(C.Structure):
    _fields_ = [
        ("q", _vp), ("k", _vp), ("vt", _vp), ("bias", _vp), ("out", _vp),
        ("n_bwin", _i), ("heads", _i), ("hd_p", _i), ("ntok", _i),
        ("H", _i), ("W", _i), ("ws", _i), ("shift", _i), ("dtype", _i), ("y_mode", _i), ("bias_frag", _vp), ("qkv_frag", _i),
        ("bias_tiles", _vp), ("x", _vp), ("wqkv", _vp), ("ldx", _i), ("C", _i), ("eps", _f),
    ]


class SrOcaAttn(C.Structure):
    _fields_ = [
        ("q", _vp), ("k", _vp), ("vt", _vp), ("bias", _vp), ("out", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("heads", _i), ("hd_p", _i), ("ws", _i), ("pad", _i), ("border", _i), ("nk_pad", _i), ("dtype", _i),
        ("bias_frag", _vp), ("nk_frag", _i), ("bias_rel", _vp),
    ]


class SrChannelAttn(C.Structure):
    _fields_ = [
        ("y", _vp), ("pool_partial", _vp), ("w1", _vp), ("b1", _vp), ("w2", _vp), ("b2", _vp),
        ("skip", _vp), ("out", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("C_p", _i), ("Cr", _i), ("n_tiles", _i),
        ("y_dtype", _i), ("skip_dtype", _i), ("out_dtype", _i), ("y_scale", _f),
        ("skip2", _vp), ("skip2_dtype", _i),
    ]


_ll = C.c_longlong


class SrBgemm(C.Structure):
    _fields_ = [
        ("A", _vp), ("B", _vp), ("C", _vp), ("bias", _vp), ("M", _i), ("N", _i), ("K", _i),
        ("sa_m", _ll), ("sa_k", _ll), ("sb_k", _ll), ("sb_n", _ll), ("sc_m", _ll), ("sc_n", _ll),
        ("nb1", _i), ("nb2", _i), ("sa_b1", _ll), ("sa_b2", _ll), ("sb_b1", _ll), ("sb_b2", _ll), ("sc_b1", _ll), ("sc_b2", _ll),
        ("alpha", _f), ("accumulate", _i), ("ksplit", _i), ("compute_dtype", _i),
    ]



# ---- fast training path (ABI v7)
class SrTrWgradJob(C.Structure):
    _fields_ = [("A", _vp), ("B", _vp), ("out", _vp), ("lda", _i), ("ldb", _i), ("Np", _i), ("Kp", _i), ("T", _i), ("taps", _i), ("H", _i), ("W", _i),
                ("ones_col", _i), ("ks", _i), ("a_f32", _i), ("b_f32", _i), ("halo", _i)]


class SrTrAttnBwd(C.Structure):
    _fields_ = [
        ("q", _vp), ("qT", _vp), ("k", _vp), ("kT", _vp), ("v", _vp), ("o", _vp), ("dO", _vp), ("dOT", _vp), ("bias", _vp), ("biasT", _vp),
        ("dq", _vp), ("dk", _vp), ("dv", _vp), ("lse", _vp), ("delta", _vp), ("dtab_part", _vp), ("rpi", _vp),
        ("n_bwin", _i), ("heads", _i), ("hd_p", _i), ("Nq", _i), ("Nk", _i), ("ldo", _i), ("groups", _i), ("T", _i), ("Tpad", _i), ("toeplitz16", _i), ("H", _i), ("W", _i), ("ws", _i), ("shift", _i),
        ("oca_rel", _i), ("lse_given", _i),
    ]


class SrTrAttnFwd(C.Structure):
    _fields_ = [("q", _vp), ("k", _vp), ("vT", _vp), ("bias", _vp), ("out", _vp), ("n_bwin", _i), ("heads", _i), ("hd_p", _i), ("Nq", _i), ("Nk", _i), ("ldo", _i),
                ("bias_rel", _vp), ("lse", _vp)]


class SrTrOcaFold(C.Structure):
    _fields_ = [("k", _vp), ("v", _vp), ("kwin", _vp), ("vwin", _vp), ("kwinT", _vp), ("vwinT", _vp), ("B", _i), ("nwy", _i), ("nwx", _i), ("heads", _i), ("wse", _i), ("pad", _i)]


class SrTrQkvFwd(C.Structure):
    _fields_ = [
        ("x", _vp), ("gamma", _vp), ("beta", _vp), ("wstream", _vp), ("q", _vp), ("qT", _vp), ("k", _vp), ("kT", _vp), ("v", _vp), ("vT", _vp), ("n1", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("Cp", _i), ("ldx", _i), ("ldn", _i), ("heads", _i), ("hd_p", _i), ("ws", _i), ("shift", _i), ("eps", _f),
    ]


class SrTrTailFwd(C.Structure):
    _fields_ = [
        ("x", _vp), ("out", _vp), ("x1", _vp), ("o", _vp), ("wstream", _vp), ("bproj", _vp), ("gamma", _vp), ("beta", _vp),
        ("y", _vp), ("pool_partial", _vp), ("ca_w1", _vp), ("ca_b1", _vp), ("ca_w2", _vp), ("ca_b2", _vp), ("gate_out", _vp), ("s_a", _vp), ("s_m", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("Cp", _i), ("ldx", _i), ("ldy", _i), ("heads", _i), ("hd_p", _i), ("ws", _i), ("shift", _i), ("Hp", _i),
        ("ca_Cr", _i), ("ca_n_tiles", _i), ("eps", _f), ("y_scale", _f),
    ]


class SrTrTailBwd(C.Structure):
    _fields_ = [
        ("dout", _vp), ("x1", _vp), ("y", _vp), ("gate", _vp), ("gamma", _vp), ("beta", _vp), ("wstream", _vp), ("s_a", _vp), ("s_m", _vp),
        ("dx1", _vp), ("n2w", _vp), ("doutw", _vp), ("gw", _vp), ("dhw", _vp), ("dOw", _vp), ("dOT", _vp), ("dx1sw", _vp), ("dyc", _vp), ("dgate_part", _vp), ("ln_part", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("Cp", _i), ("ldx", _i), ("ldy", _i), ("heads", _i), ("hd_p", _i), ("ws", _i), ("shift", _i), ("Hp", _i), ("eps", _f),
    ]


class SrTrQkvBwd(C.Structure):
    _fields_ = [
        ("dx1", _vp), ("x", _vp), ("dq", _vp), ("dk", _vp), ("dv", _vp), ("dn1c", _vp), ("gamma", _vp), ("beta", _vp), ("wstream", _vp),
        ("dx", _vp), ("n1w", _vp), ("dqkvw", _vp), ("ln_part", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("Cp", _i), ("ldx", _i), ("ldn", _i), ("heads", _i), ("hd_p", _i), ("ws", _i), ("shift", _i), ("eps", _f),
    ]


class SrTrLnBwd(C.Structure):
    _fields_ = [("x", _vp), ("dy", _vp), ("gamma", _vp), ("dskip", _vp), ("dx", _vp), ("ln_part", _vp), ("M", _ll), ("C", _i), ("Cp", _i), ("ld", _i),
                ("dy_bf16", _i), ("dskip_bf16", _i), ("eps", _f)]


class SrTrCaBwd(C.Structure):
    _fields_ = [
        ("dgate_part", _vp), ("pool_partial", _vp), ("w1", _vp), ("b1", _vp), ("w2", _vp), ("b2", _vp), ("dy", _vp), ("dparam_part", _vp),
        ("B", _i), ("H", _i), ("W", _i), ("C", _i), ("Cp", _i), ("Cr", _i), ("n_tiles", _i), ("parts", _i), ("ld", _i), ("dparam_stride", _i), ("y_scale", _f),
    ]


EW_GELU_FWD, EW_GELU_BWD, EW_RELU_FWD, EW_RELU_BWD, EW_LRELU_FWD, EW_LRELU_BWD, EW_AXPBY, EW_MUL = range(8)
EW_SIGMOID_FWD, EW_SIGMOID_BWD, EW_SCALE_SAMPLE, EW_MUL_BC, EW_BCAST_BC, EW_AFFINE_C = range(8, 14)

class SrTrGelu(C.Structure):  # ABI v10
    _fields_ = [("x", _vp), ("dg", _vp), ("g", _vp), ("dx", _vp), ("n", _ll)]


class SrTrAdd(C.Structure):
    _fields_ = [("a", _vp), ("b", _vp), ("out", _vp), ("n", _ll), ("b_dtype", _i)]


class SrTrFinalize(C.Structure):
    _fields_ = [("arena", _vp), ("src", _vp), ("dst", _vp), ("stride", _vp), ("ns", _vp), ("scale", _vp), ("grad", _vp), ("n", _ll), ("lanes", _i)]


class SrTrUnshuffle(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("B", _i), ("H", _i), ("W", _i), ("cps", _i), ("r", _i)]


class SrTrLreluBwd(C.Structure):
    _fields_ = [("dy", _vp), ("y", _vp), ("dx", _vp), ("n", _ll), ("slope", _f)]


class SrLayernorm(C.Structure):
    _fields_ = [("x", _vp), ("y", _vp), ("gamma", _vp), ("beta", _vp), ("y_dtype", _i), ("M", _i), ("C", _i), ("Cp", _i), ("eps", _f)]


class SrPlanOp(C.Structure):  # ABI v10 (csrc/sr_plan.cpp)
    _fields_ = [("kind", _i), ("stream", _i), ("ival", _i), ("arg_bytes", _i), ("arg2_bytes", _i), ("reserved", _i), ("fn", _vp), ("arg", _vp), ("arg2", _vp)]


PLAN_CALL1, PLAN_CALL2, PLAN_CALLI, PLAN_EVENT_RECORD, PLAN_STREAM_WAIT = range(5)

# every symbol include/studiosr_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sr_abi_version": (_i, []),
    "sr_last_error": (C.c_char_p, []),
    "sr_ingest_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sr_layernorm": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "sr_layernorm_to": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _f, _vp]),
    "sr_gemm": (_i, [C.POINTER(SrGemm), _vp]),
    "sr_swin_block_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "sr_swin_block": (_i, [C.POINTER(SrSwinBlock), _vp]),
    "sr_swin_light_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "sr_swin_light": (_i, [C.POINTER(SrSwinLight), _vp]),
    "sr_cab_supported": (_i, [_i, _i, _i, _i]),
    "sr_cab_pool_tiles": (_i, [_i, _i]),
    "sr_cab_pool_tiles_rows": (_i, [_i, _i, _i]),
    "sr_cab_fused": (_i, [C.POINTER(SrCab), _vp]),
    "sr_swin_qkv_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "sr_swin_qkv": (_i, [C.POINTER(SrSwinQkv), _vp]),
    "sr_swin_tail_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "sr_swin_tail": (_i, [C.POINTER(SrSwinTail), _vp]),
    "sr_mlp_fused_supported": (_i, [_i, _i, _i]),
    "sr_mlp_fused": (_i, [C.POINTER(SrMlp), _vp]),
    "sr_conv3x3": (_i, [C.POINTER(SrConv3x3), _vp]),
    "sr_conv3x3_pool_tiles": (_i, [_i, _i, _i, _i]),
    "sr_conv3x3_pool_tiles_rows": (_i, [_i, _i, _i, _i]),
    "sr_window_attention": (_i, [C.POINTER(SrWindowAttn), _vp]),
    "sr_hab_mid_supported": (_i, [_i] * 8),
    "sr_hab_mid": (_i, [C.POINTER(SrWindowAttn), C.POINTER(SrCab), _vp]),  # ABI v8
    "sr_oca_attention": (_i, [C.POINTER(SrOcaAttn), _vp]),
    "sr_pixel_shuffle_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sr_rcab_conv_pair": (_i, [C.POINTER(SrRcab), _vp]),
    "sr_rcab_pool_tiles": (_i, [_i, _i]),
    "sr_u8_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "sr_nchw_to_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "sr_channel_attention": (_i, [C.POINTER(SrChannelAttn), _vp]),
    "sr_channel_gate": (_i, [C.POINTER(SrChannelAttn), _vp, _vp]),
    # training engine (ABI v3)
    "sr_bgemm": (_i, [C.POINTER(SrBgemm), _vp]),
    "sr_im2col3x3": (_i, [_vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _vp]),
    "sr_col2im3x3": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sr_softmax_fwd": (_i, [_vp, _vp, _vp, _ll, _i, _i, _i, _i, _vp]),
    "sr_softmax_bwd": (_i, [_vp, _vp, _ll, _i, _vp]),
    "sr_layernorm_fwd_train": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _f, _vp]),
    "sr_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp]),
    "sr_colsum": (_i, [_vp, _vp, _i, _ll, _i, _f, _vp]),
    "sr_batch_sum": (_i, [_vp, _vp, _ll, _ll, _ll, _vp]),
    "sr_eltwise": (_i, [_i, _vp, _vp, _vp, _vp, _ll, _ll, _i, _f, _f, _vp]),
    "sr_window_copy": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sr_oca_unfold": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sr_pixel_shuffle_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sr_bias_gather": (_i, [_vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp]),
    "sr_nhwc_out": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sr_copy_cols": (_i, [_vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp]),
    "sr_conv3d27": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sr_conv3d27_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    # device-side weight packing
    "sr_pack_matrix": (_i, [_vp, _ll, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sr_pack_conv3x3": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sr_pack_vector": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sr_pack_bias_fragments": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    # fast training path (ABI v7)
    "sr_tr_gather": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _vp]),
    "sr_tr_finalize": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "sr_tr_finalize_to": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "sr_tr_finalize_to8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "sr_tr_wgrad": (_i, [C.POINTER(SrTrWgradJob), _i, _vp]),
    "sr_tr_wgrad_out_floats": (_ll, [C.POINTER(SrTrWgradJob)]),
    "sr_tr_attn_bwd": (_i, [C.POINTER(SrTrAttnBwd), _vp]),
    "sr_tr_attn_fwd": (_i, [C.POINTER(SrTrAttnFwd), _vp]),
    "sr_tr_oca_fold": (_i, [C.POINTER(SrTrOcaFold), _i, _vp]),
    "sr_tr_block_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "sr_tr_qkv_fwd": (_i, [C.POINTER(SrTrQkvFwd), _vp]),
    "sr_tr_tail_fwd": (_i, [C.POINTER(SrTrTailFwd), _vp]),
    "sr_tr_tail_bwd": (_i, [C.POINTER(SrTrTailBwd), _vp]),
    "sr_tr_qkv_bwd": (_i, [C.POINTER(SrTrQkvBwd), _vp]),
    "sr_tr_ca_bwd": (_i, [C.POINTER(SrTrCaBwd), _vp]),
    "sr_tr_gelu": (_i, [_vp, _vp, _vp, _vp, _ll, _vp]),
    "sr_tr_gelu_args": (_i, [C.POINTER(SrTrGelu), _vp]),
    "sr_tr_add_args": (_i, [C.POINTER(SrTrAdd), _vp]),
    "sr_tr_finalize_to_args": (_i, [C.POINTER(SrTrFinalize), _vp]),
    "sr_tr_unshuffle_args": (_i, [C.POINTER(SrTrUnshuffle), _vp]),
    "sr_tr_lrelu_bwd_args": (_i, [C.POINTER(SrTrLreluBwd), _vp]),
    "sr_layernorm_to_args": (_i, [C.POINTER(SrLayernorm), _vp]),
    # launch plans (ABI v10)
    "sr_plan_create": (_vp, [C.POINTER(SrPlanOp), _i, _i]),
    "sr_plan_streams": (_i, [_vp]),
    "sr_plan_ops": (_i, [_vp]),
    "sr_plan_run": (_i, [_vp, C.POINTER(_vp), _i]),
    "sr_plan_destroy": (None, [_vp]),
    "sr_tr_ln_bwd": (_i, [C.POINTER(SrTrLnBwd), _vp]),
    "sr_tr_unshuffle": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sr_tr_lrelu_bwd": (_i, [_vp, _vp, _vp, _f, _ll, _vp]),
    "sr_tr_add": (_i, [_vp, _vp, _i, _vp, _ll, _vp]),
    "sr_tr_adam": (_i, [_vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _ll, _vp]),
}

_lib = None
_lock = threading.Lock()


class HipLibraryError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile libstudiosr_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    jobs = str(min(8, os.cpu_count() or 1))
    proc = subprocess.run(["make", "-C", CSRC, "-j", jobs], capture_output=True, text=True)
    if proc.returncode != 0:
        raise HipLibraryError("building libstudiosr_hip.so failed:\n" + proc.stdout[-4000:] + proc.stderr[-4000:])
    if verbose:
        print(proc.stdout)
    return LIB_PATH


def lib():
    """The loaded library; raises HipLibraryError when it is absent (no CPU fallback exists).  While a launch plan is being recorded (`recording()`)
    the returned object records every launch entry point instead of calling it."""
    if _REC is not None:
        return _REC.proxy
    return _handle()


def _handle() -> C.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise HipLibraryError(
                        f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C studiosr_amd/csrc`). studiosr_amd has no CPU fallback."
                    )
                handle = C.CDLL(LIB_PATH)
                for name, (res, args) in SYMBOLS.items():
                    fn = getattr(handle, name)  # AttributeError if the ABI drifted
                    fn.restype, fn.argtypes = res, args
                if handle.sr_abi_version() != ABI_VERSION:
                    raise HipLibraryError("libstudiosr_hip.so ABI version mismatch")
                _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipLibraryError(f"{what} failed ({rc}): {_handle().sr_last_error().decode()}")


# --------------------------------------------------------------------------- launch plans (ABI v10, csrc/sr_plan.cpp)
_REC = None  # the PlanRecorder that is recording, if any (one thread drives a model: no lock)


def _is_launch(name: str) -> bool:
    res, args = SYMBOLS.get(name, (None, []))
    return res is _i and len(args) >= 2 and args[-1] is _vp and not name.startswith(("sr_plan_", "sr_pack_"))


class _RecordingLib:
    """What `lib()` returns while a plan is recorded: launch entry points are recorded (and report success), everything else passes through."""

    def __init__(self, rec: "PlanRecorder") -> None:
        self._rec, self._h = rec, _handle()

    def __getattr__(self, name: str):
        fn = getattr(self._h, name)
        if not _is_launch(name):
            return fn
        rec = self._rec

        def record(*args):
            rec.add_call(name, fn, args)
            return 0

        setattr(self, name, record)
        return record


class LaunchPlan:
    """A recorded launch sequence: C segments (sr_plan_run: one call enqueues the segment) with the few launches that have no argument-block form
    (positional scalars: sr_tr_add, sr_tr_finalize_to, ...) kept as Python closures between them."""

    def __init__(self, segments, side_streams, n_launches: int) -> None:
        self.segments, self.side_streams, self.n_launches = segments, side_streams, n_launches
        self._tab = (_vp * (1 + len(side_streams)))()
        for i, s in enumerate(side_streams):
            self._tab[1 + i] = s

    def run(self, stream: int) -> None:
        h = _handle()
        tab = self._tab
        tab[0] = stream
        n = len(tab)
        for kind, seg in self.segments:
            if kind == "c":
                rc = h.sr_plan_run(seg, tab, n)
                if rc != 0:
                    check(rc, "sr_plan_run")
            else:
                seg(tab)

    def __del__(self):
        try:
            h = _handle()
            for kind, seg in self.segments:
                if kind == "c":
                    h.sr_plan_destroy(seg)
        except Exception:
            pass


class PlanRecorder:
    """Records the launches made through `lib()` between `with recording(rec):` ... (nothing is enqueued meanwhile).  Streams are recorded by handle:
    the stream that is current when recording starts becomes slot 0 and is replaced by the caller's current stream at every run; other handles (the
    side streams of two-branch blocks: persistent torch streams) are replayed as they are."""

    def __init__(self, main_stream: int) -> None:
        self.main = int(main_stream or 0)
        self.side: list = []
        self.items: list = []  # ("op", SrPlanOp, keep-alive) | ("py", closure)
        self.n_events = 0
        self.n_launches = 0
        self.proxy = _RecordingLib(self)

    def slot(self, stream) -> int:
        s = int(stream or 0)
        if s == self.main:
            return 0
        if s not in self.side:
            self.side.append(s)
        return 1 + self.side.index(s)

    @staticmethod
    def _block(a):
        """(address, bytes, keep-alive) of an argument block given as byref(struct), a ctypes array or a ctypes structure; None for anything else."""
        obj = getattr(a, "_obj", a)  # C.byref(x) keeps x in ._obj
        if isinstance(obj, (C.Structure, C.Array)):
            return C.addressof(obj), C.sizeof(obj), obj
        return None

    def add_call(self, name: str, fn, args) -> None:
        *a, st = args
        slot = self.slot(st)
        self.n_launches += 1
        blocks = [self._block(x) for x in a]
        op = SrPlanOp()
        op.stream, op.fn = slot, C.cast(fn, _vp).value
        if len(a) == 1 and blocks[0]:
            op.kind = PLAN_CALL1
        elif len(a) == 2 and blocks[0] and blocks[1]:
            op.kind = PLAN_CALL2
            op.arg2, op.arg2_bytes = blocks[1][0], blocks[1][1]
        elif len(a) == 2 and blocks[0] and isinstance(a[1], int):
            op.kind, op.ival = PLAN_CALLI, a[1]
        else:  # positional scalars / raw pointers: replayed from Python with the run's stream
            frozen = tuple(a)
            self.items.append(("py", lambda tab, fn=fn, frozen=frozen, slot=slot, name=name: check(fn(*frozen, tab[slot]), name)))
            return
        op.arg, op.arg_bytes = blocks[0][0], blocks[0][1]
        self.items.append(("op", op, [b[2] for b in blocks if b]))

    def event(self) -> int:
        self.n_events += 1
        return self.n_events - 1

    def add_event_record(self, stream, ev: int) -> None:
        op = SrPlanOp()
        op.kind, op.stream, op.ival = PLAN_EVENT_RECORD, self.slot(stream), ev
        self.items.append(("op", op, None))

    def add_stream_wait(self, stream, ev: int) -> None:
        op = SrPlanOp()
        op.kind, op.stream, op.ival = PLAN_STREAM_WAIT, self.slot(stream), ev
        self.items.append(("op", op, None))

    def finish(self) -> LaunchPlan:
        h = _handle()
        segments, run = [], []

        def flush():
            if not run:
                return
            arr = (SrPlanOp * len(run))(*run)
            # events are per C segment: an edge may not cross a Python-replayed launch
            recorded = set()
            for o in run:
                if o.kind == PLAN_EVENT_RECORD:
                    recorded.add(o.ival)
                elif o.kind == PLAN_STREAM_WAIT and o.ival not in recorded:
                    raise HipLibraryError("launch plan: a cross-stream edge spans a launch that is replayed from Python")
            seg = h.sr_plan_create(arr, len(run), self.n_events)
            if not seg:
                raise HipLibraryError("sr_plan_create failed: " + h.sr_last_error().decode())
            segments.append(("c", seg))
            run.clear()

        for it in self.items:
            if it[0] == "op":
                run.append(it[1])
            else:
                flush()
                segments.append(("py", it[1]))
        flush()
        return LaunchPlan(segments, list(self.side), self.n_launches)


class recording:
    """with recording(rec): ...   -- launches made through lib() inside the block are recorded into rec, not enqueued."""

    def __init__(self, rec: PlanRecorder) -> None:
        self.rec = rec

    def __enter__(self):
        global _REC
        assert _REC is None, "a launch plan is already being recorded"
        _REC = self.rec
        return self.rec

    def __exit__(self, *exc):
        global _REC
        _REC = None
        return False


def recorder() -> Optional["PlanRecorder"]:
    return _REC
