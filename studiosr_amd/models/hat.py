"""HAT on the MI355X HIP hot path (reference: studiosr/models/hat.py).

Same constructor kwargs / attributes / state_dict keys (incl. the model-level buffers
relative_position_index_SA / _OCA, hat.py:428-431) as the reference.  forward is a sequence of C-ABI launches:

  ingest (reflect pad + normalise, hat.py:544-546) -> conv_first -> LayerNorm
  per HAB (hat.py:153-195):
      LayerNorm kernel -> CAB: conv3x3+GELU -> conv3x3 (+ per-tile channel sums)              (:41-52)
      (shifted) window attention on x: fused kernel when the geometry allows, else
          [LN + QKV GEMM] -> window attention -> [proj GEMM + shortcut]                        (:55-110)
      channel-attention gate: x = (shortcut + attn) + conv_scale * cab * sigmoid(..)           (:25-38,192)
      MLP: fused LN + fc1 + GELU + fc2 + residual                                              (:193)
  per OCAB (hat.py:239-293):
      [LN + QKV GEMM] with q -> window order, k -> zero-bordered image, v -> transposed zero-bordered planes
      overlapping cross attention (neighbourhood gathered by addressing, never unfolded)
      [proj GEMM + shortcut] -> MLP
  per RHAG: conv3x3 + residual (:385);  tail as SwinIR (:549-554)
"""
from __future__ import annotations

import os
from typing import Dict, List

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops, packing
from ..runtime import capturing_or_warming_up, compute_dtype, knob, sr_dtype
from .common import Model, Upsampler, conv_call, pack_upsampler, run_upsampler
from .rcan import pack_ca, run_channel_attention
from .swinir import (
    MlpParams,
    PatchEmbed,
    SwinGeometry,
    final_affine,
    fold_ln,
    ingest_affine,
    pack_attention,
    pack_ln,
    pack_mlp,
    pack_qkv_stream,
    pack_tail_stream,
    qkv_frag_order,
    run_mlp,
    run_swin_tail,
    run_window_msa,
    swin_qkv_usable,
    swin_tail_usable,
)

Tensor = torch.Tensor


# --------------------------------------------------------------------------- parameter containers
class ChannelAttention(nn.Module):
    """hat.py:25-34 (keys attention.1 / attention.3)."""

    def __init__(self, num_feat: int, squeeze_factor: int = 16) -> None:
        super().__init__()
        self.attention = nn.Sequential(
            nn.AdaptiveAvgPool2d(1),
            nn.Conv2d(num_feat, num_feat // squeeze_factor, 1, padding=0),
            nn.ReLU(inplace=True),
            nn.Conv2d(num_feat // squeeze_factor, num_feat, 1, padding=0),
            nn.Sigmoid(),
        )


class CAB(nn.Module):
    """hat.py:41-49 (keys cab.0 / cab.2 / cab.3)."""

    def __init__(self, num_feat: int, compress_ratio: int = 3, squeeze_factor: int = 30) -> None:
        super().__init__()
        self.cab = nn.Sequential(
            nn.Conv2d(num_feat, num_feat // compress_ratio, 3, 1, 1),
            nn.GELU(),
            nn.Conv2d(num_feat // compress_ratio, num_feat, 3, 1, 1),
            ChannelAttention(num_feat, squeeze_factor),
        )


class WindowAttention(nn.Module):
    """hat.py:55-83 (no index buffer: HAT keeps it at model level)."""

    def __init__(self, dim: int, window_size: int, num_heads: int) -> None:
        super().__init__()
        ws = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class HAB(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio, compress_ratio, squeeze_factor, conv_scale) -> None:
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.shift_size, self.conv_scale = shift_size, conv_scale
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.conv_block = CAB(dim, compress_ratio, squeeze_factor)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MlpParams(dim, int(dim * mlp_ratio))


class OCAB(nn.Module):
    def __init__(self, dim, num_heads, window_size, mlp_ratio, overlap_ratio) -> None:
        super().__init__()
        self.overlap_win_size = int(window_size * overlap_ratio) + window_size
        self.norm1 = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, dim * 3)
        n = window_size + self.overlap_win_size - 1
        self.relative_position_bias_table = nn.Parameter(torch.zeros(n * n, num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MlpParams(dim, int(dim * mlp_ratio))


class AttenBlocks(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, compress_ratio, squeeze_factor, conv_scale, overlap_ratio) -> None:
        super().__init__()
        self.blocks = nn.ModuleList(
            [HAB(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio, compress_ratio, squeeze_factor, conv_scale) for i in range(depth)]
        )
        self.overlap_attn = OCAB(dim, num_heads, window_size, mlp_ratio, overlap_ratio)


class RHAG(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, compress_ratio, squeeze_factor, conv_scale, overlap_ratio) -> None:
        super().__init__()
        self.residual_group = AttenBlocks(dim, depth, num_heads, window_size, mlp_ratio, compress_ratio, squeeze_factor, conv_scale, overlap_ratio)
        self.conv = nn.Conv2d(dim, dim, 3, 1, 1)


def rpi_sa(ws: int) -> Tensor:
    """hat.py:480-492."""
    ys, xs = torch.div(torch.arange(ws * ws), ws, rounding_mode="floor"), torch.arange(ws * ws) % ws
    return (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)


def rpi_oca(ws: int, overlap_ratio: float) -> Tensor:
    """hat.py:494-517 (entries may be negative: the table is indexed python-style, i.e. they wrap)."""
    wse = ws + int(overlap_ratio * ws)
    qy, qx = torch.div(torch.arange(ws * ws), ws, rounding_mode="floor"), torch.arange(ws * ws) % ws
    ky, kx = torch.div(torch.arange(wse * wse), wse, rounding_mode="floor"), torch.arange(wse * wse) % wse
    dy = ky[None, :] - qy[:, None] + ws - wse + 1
    dx = kx[None, :] - qx[:, None] + ws - wse + 1
    return dy * (ws + wse - 1) + dx


class HAT(Model):
    def __init__(
        self,
        scale: int = 4,
        n_colors: int = 3,
        img_range: float = 1.0,
        embed_dim: int = 180,
        depths: List[int] = [6, 6, 6, 6, 6, 6],
        num_heads: List[int] = [6, 6, 6, 6, 6, 6],
        window_size: int = 16,
        mlp_ratio: float = 2.0,
        drop_rate: float = 0.0,
        attn_drop_rate: float = 0.0,
        drop_path_rate: float = 0.1,
        compress_ratio: int = 3,
        squeeze_factor: int = 30,
        conv_scale: float = 0.01,
        overlap_ratio: float = 0.5,
    ) -> None:
        super().__init__(scale, n_colors, img_range)
        assert n_colors == 3, "Normalizer mean has 3 channels (common.py:223)"
        self.embed_dim, self.depths, self.num_heads, self.window_size, self.mlp_ratio = embed_dim, depths, num_heads, window_size, mlp_ratio
        self.drop_rate, self.attn_drop_rate, self.drop_path_rate = drop_rate, attn_drop_rate, drop_path_rate
        self.compress_ratio, self.squeeze_factor, self.conv_scale, self.overlap_ratio = compress_ratio, squeeze_factor, conv_scale, overlap_ratio
        self.shift_size = window_size // 2
        self.register_buffer("relative_position_index_SA", rpi_sa(window_size))
        self.register_buffer("relative_position_index_OCA", rpi_oca(window_size, overlap_ratio))
        self.conv_first = nn.Conv2d(n_colors, embed_dim, 3, 1, 1)
        self.patch_embed = PatchEmbed(embed_dim)
        self.layers = nn.ModuleList(
            [RHAG(embed_dim, depths[i], num_heads[i], window_size, mlp_ratio, compress_ratio, squeeze_factor, conv_scale, overlap_ratio) for i in range(len(depths))]
        )
        self.norm = nn.LayerNorm(embed_dim)
        self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1)
        num_feat = 64
        self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
        self.upsample = Upsampler(scale, num_feat)
        self.conv_last = nn.Conv2d(num_feat, n_colors, 3, 1, 1)
        self.apply(self._init_weights)

    def _init_weights(self, m: nn.Module) -> None:  # hat.py:471-478
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
        Cp = self._geo(0).Cp
        dev = self.conv_first.weight.device
        ws = self.window_size
        wse = ws + int(self.overlap_ratio * ws)
        nk = wse * wse
        nk_pad = packing.round_up(nk, 32)
        c3 = C // self.compress_ratio
        c3p = packing.round_up(c3, 32)
        ident = packing.identity_idx(C, Cp)
        P: Dict = dict(Cp=Cp, c3p=c3p, wse=wse, nk_pad=nk_pad, pad=(wse - ws) // 2)
        P["border"] = packing.round_up(P["pad"], 4)
        P["first"] = packing.pack_conv3x3(self.conv_first.weight, self.conv_first.bias, 32, ident, dt)
        P["pe_norm"] = pack_ln(self.patch_embed.norm, Cp)
        P["layers"] = []
        for li, layer in enumerate(self.layers):
            geo = self._geo(li)
            if geo.hd_p != 32:
                raise NotImplementedError("HAT HIP path supports head_dim <= 32")
            blocks = []
            for blk in layer.residual_group.blocks:
                e = dict(shift=blk.shift_size, ln1=pack_ln(blk.norm1, Cp), ln2=pack_ln(blk.norm2, Cp))
                e.update(pack_attention(blk.attn, geo, dt, rpi=self.relative_position_index_SA, norm=blk.norm1))
                e.update(pack_mlp(blk.mlp, geo, dt, norm=blk.norm2))
                e.update(pack_tail_stream(blk.attn.proj, blk.mlp, blk.norm2, geo, dt))
                e.update(pack_qkv_stream(blk.attn, blk.norm1, geo, dt))
                cab = blk.conv_block.cab
                e["cab1"] = packing.pack_conv3x3(cab[0].weight, cab[0].bias, Cp, packing.identity_idx(c3, c3p), dt)
                e["cab2"] = packing.pack_conv3x3(cab[2].weight, cab[2].bias, c3p, ident, dt)
                att = cab[3].attention
                e["ca"] = pack_ca(att[1].weight, att[1].bias, att[3].weight, att[3].bias)
                blocks.append(e)
            for i in range(len(blocks) - 1):  # sr_swin_tail of block i continues with block i + 1's LayerNorm1 + QKV: one stream
                if blocks[i].get("tail_dtype") == L.SR_BF16 and "qkv_stream" in blocks[i + 1]:
                    blocks[i]["tail_qkv_stream"] = torch.cat([blocks[i]["tail_stream"], blocks[i + 1]["qkv_stream"]]).contiguous()
            oc = layer.residual_group.overlap_attn
            o = dict(ln1=pack_ln(oc.norm1, Cp), ln2=pack_ln(oc.norm2, Cp))
            o.update(pack_attention(oc, geo, dt, rpi=self.relative_position_index_SA, norm=oc.norm1))  # qkv / proj packing (bias replaced below)
            ob = packing.gather_bias(oc.relative_position_bias_table, self.relative_position_index_OCA, geo.ntok, nk)  # [heads, nq, nk], negative idx wrap
            obp = torch.zeros(ob.shape[0], ob.shape[1], nk_pad, dtype=torch.float32, device=ob.device)
            obp[:, :, :nk] = ob
            o["oca_bias"] = obp.contiguous()
            nk_frag = packing.round_up(nk, 64)  # flash-form kernel: key blocks of 64, pad columns vanish in the softmax
            obf = torch.full((ob.shape[0], ob.shape[1], nk_frag), -1.0e30, dtype=torch.float32, device=ob.device)
            obf[:, :, :nk] = ob
            o["oca_bias_frag"], o["oca_nk_frag"] = packing.bias_fragments(obf), nk_frag
            if ws == 16 and wse == 24 and dt == torch.bfloat16:  # the bias as its relative-position table: selects the LDS form of the attention (ABI v8)
                rel = packing.oca_bias_rel(ob)
                if rel is not None:
                    o["oca_bias_rel"] = rel
            o.update(pack_mlp(oc.mlp, geo, dt, norm=oc.norm2))
            o.update(pack_tail_stream(oc.proj, oc.mlp, oc.norm2, geo, dt))
            o.update(pack_qkv_stream(oc, oc.norm1, geo, dt))
            if blocks and blocks[-1].get("tail_dtype") == L.SR_BF16 and o.get("qkv_dtype") == L.SR_BF16:  # the last HAB's sr_swin_tail continues with the OCAB's LayerNorm1 + QKV
                blocks[-1]["tail_oca_stream"] = torch.cat([blocks[-1]["tail_stream"], o["qkv_stream"]]).contiguous()
            conv = packing.pack_conv3x3(layer.conv.weight, layer.conv.bias, Cp, ident, dt)
            P["layers"].append(dict(blocks=blocks, ocab=o, conv=conv, geo=geo))
        P["norm"] = pack_ln(self.norm, Cp)
        P["after_body"] = packing.pack_conv3x3(self.conv_after_body.weight, self.conv_after_body.bias, Cp, ident, dt)
        P["fin"] = final_affine(self.img_range, self.n_colors, dev)
        P["ing"] = ingest_affine(self.img_range, self.n_colors, dev)
        cbu = self.conv_before_upsample[0]
        P["before_up"] = packing.pack_conv3x3(cbu.weight, cbu.bias, Cp, packing.identity_idx(64, 64), dt)
        P["up"] = pack_upsampler(self.upsample, 64, dt)
        P["last"] = packing.pack_conv3x3(self.conv_last.weight, self.conv_last.bias, 64, packing.identity_idx(self.n_colors, 16), dt)
        return P

    # ------------------------------------------------------------------ blocks
    dual_stream = True  # HAB: conv branch beside the attention branch on a second HIP stream

    def _side_stream(self, device) -> "torch.cuda.Stream":
        st = getattr(self, "_side", None)
        if st is None or st.device != torch.device(device):
            st = torch.cuda.Stream(device=device)
            object.__setattr__(self, "_side", st)
        return st

    def _run_hab(self, bp: Dict, geo: SwinGeometry, P: Dict, t_in: Tensor, t: Tensor, ws_, cdt, n1_ready: bool = False, next_ln=None,
                 qkv_ready: bool = False, next_bp=None, next_oca=None):
        """t = HAB(t_in); t_in may be t (in place).
        n1_ready / qkv_ready: "hab.n1" already holds LayerNorm1(t_in) / "hab.q", "hab.k", "hab.vt" already hold this block's q, k, v^T (written by
        the previous block's sr_swin_tail).  next_ln: norm1 (gamma, beta) of the block that follows, next_bp: its packed entry (next_oca: the group's OCAB, for the last block).  Returns
        (n1 written for the next block, q / k / v^T written for the next block)."""
        B, H, W, Cp = t_in.shape
        f32 = torch.float32
        # conv branch on LayerNorm1(x)  (hat.py:165-170)
        n1 = ws_.get("hab.n1", (B, H, W, Cp), cdt)  # consumed only by the conv, which rounds to the compute dtype anyway
        unfused = True  # (the attention half is always QKV -> attention -> tail launches; the round-1 one-kernel attention half left with ABI v10)
        mid = ws_.get("hab.mid", (B, H, W, P["c3p"]), cdt)
        y = ws_.get("hab.y", (B, H, W, Cp), cdt)  # enters the block scaled by conv_scale = 0.01
        # small batches: 4-row conv tiles (twice the workgroups) also for the conv with the pool side output
        th = 4 if (cdt == torch.bfloat16 and ((W + 15) // 16) * ((H + 7) // 8) * B < 256) else 0
        # conv -> GELU -> conv as ONE launch (sr_cab_fused; SR_CAB_FUSED=0: two sr_conv3x3 launches)
        # (round 5, ABI v11: sr_cab_fused also has a split-operand form for precision "fp32x3"; it is correct and tested but one workgroup per CU -- 103 KB of images -- and at
        # HAT's sizes SLOWER than the two split-operand sr_conv3x3 launches: x4 b4 6.71 vs 6.26 ms, b16 21.8 vs 20.7: SR_CAB_X3=1 selects it)
        from ..runtime import x3_active

        cab_x3 = cdt == torch.float32 and x3_active() and knob("SR_CAB_X3", "0") != "0"
        cab_code = L.SR_BF16X3 if cab_x3 else L.SR_BF16
        cab_fused = (cdt == torch.bfloat16 or cab_x3) and knob("SR_CAB_FUSED", "1") != "0" and ops.cab_supported(Cp, P["c3p"], Cp, cab_code)
        # The conv branch (LayerNorm1, 2 convs, gate) and the attention branch (QKV GEMM, attention) only share their input, and at the
        # tile sizes of this model every launch is a fraction of the chip: the conv branch runs on a side stream beside the attention
        # branch and joins before the projection GEMM -- the first writer of t (which may be t_in) and the consumer of the conv branch.
        # Also valid under HIP-graph capture.
        main = torch.cuda.current_stream(t_in.device)
        side = self._side_stream(t_in.device) if (self.dual_stream and knob("SR_HAT_DUAL", "1") != "0") else main
        gate = ws_.get("hab.gate", (B, Cp), f32)
        w1, b1, w2, b2 = bp["ca"]
        gate_in_tail = unfused and swin_tail_usable(bp, geo, Cp, cdt) and knob("SR_TAIL_GATE", "1") != "0" and w1.shape[0] <= 8
        # the tail kernel goes on with the next block's LayerNorm1 + QKV (its attention kernel is the next launch of the chain)
        fuse_next_qkv = (unfused and cdt == torch.bfloat16 and next_bp is not None and "tail_qkv_stream" in bp and swin_tail_usable(bp, geo, Cp, cdt) and swin_qkv_usable(next_bp, geo, Cp, cdt)
                         and knob("SR_TAIL_QKV", "1") != "0" and bp["shift"] % 4 == 0 and next_bp["shift"] % 4 == 0)

        # the group's last block: the tail goes on with the OCAB's LayerNorm1 + QKV instead (q in window order, k / v^T in the zero-bordered layouts; ABI v9, SR_TAIL_OCA=0)
        fuse_oca = (unfused and cdt == torch.bfloat16 and next_bp is None and next_oca is not None and "tail_oca_stream" in bp and swin_tail_usable(bp, geo, Cp, cdt)
                    and swin_qkv_usable(next_oca, geo, Cp, cdt) and P["border"] % 4 == 0 and knob("SR_TAIL_OCA", "1") != "0" and bp["shift"] % 4 == 0)

        # attention + CAB as ONE launch (sr_hab_mid, ABI v8): no second stream, no fork / join edges in the captured graph (SR_HAB_MID=0: two launches)
        mid_fused = (unfused and cab_fused and not cab_x3 and gate_in_tail and knob("SR_HAB_MID", "1") != "0" and (qkv_ready or swin_qkv_usable(bp, geo, Cp, cdt))
                     and ops.hab_mid_supported(geo.ntok, geo.hd_p, geo.ws, L.SR_BF16, Cp, P["c3p"], Cp, L.SR_BF16))
        # the attention workgroups of sr_hab_mid project their own head from the stream (LayerNorm1 + QKV inside the attention role: no QKV stage on the tail's chain)
        qkv_in_attn = (mid_fused and not qkv_ready and "bias_tiles" in bp and bp.get("qkv_dtype") == L.SR_BF16 and geo.heads == 6 and geo.C == 180
                       and knob("SR_ATTN_LDS", "1") != "0"
                       # every head's workgroup re-reads and re-normalises its window's rows (6 x the stream through L2: 75 MB per launch at 4 x 64 x 64): it pays while the
                       # launch is a latency chain (a single tile: 2.10 -> 1.97 ms) and costs above (b4 2.54 -> 2.92 ms, b16 7.00 -> 7.52); SR_ATTN_QKV=1 / 0 forces it
                       and (knob("SR_ATTN_QKV", "auto") == "1" or (knob("SR_ATTN_QKV", "auto") == "auto" and (getattr(self, "_total_B", B) * H * W // geo.ntok) * geo.heads <= 128)))
        if qkv_in_attn:
            fuse_next_qkv = False
        # CAB tiles of 14 x 8 outputs instead of 14 x 6 in sr_hab_mid from 4 x 64 x 64 pixels on (SrCab.tile_rows, ABI v9; SR_CAB_ROWS8_FROM pixels): b16 6.52 -> 6.28 ms,
        # 64 tiles 27.7 -> 26.7, b4 2.44 -> 2.41; a single tile 1.64 -> 1.74 the other way (there the launch is one short chain per CU)
        cab_rows = 8 if (mid_fused and not qkv_in_attn and "bias_tiles" in bp and knob("SR_ATTN_LDS", "1") != "0"
                         and getattr(self, "_total_B", B) * H * W >= int(knob("SR_CAB_ROWS8_FROM", "16384"))) else 0
        n_tiles = (ops.cab_pool_tiles_rows(H, W, cab_rows) if cab_fused else ops.conv_pool_tiles(H, W, Cp, sr_dtype(cdt), th))
        pool = ws_.get("hab.pool", (B, n_tiles, Cp), f32)
        qkv_n1 = None
        if mid_fused:
            side = main
            if not n1_ready:
                if not qkv_ready and not qkv_in_attn and True:  # a group's first block: LayerNorm1 leaves sr_swin_qkv as a side output (one launch less)
                    qkv_n1 = (n1, *bp["ln1"])
                else:
                    ops.layernorm(t_in, n1, *bp["ln1"], self.embed_dim)
        cab_kw = dict(x=n1.data_ptr(), w1p=bp["cab1"][0].data_ptr(), b1=bp["cab1"][1].data_ptr(), w2p=bp["cab2"][0].data_ptr(), b2=bp["cab2"][1].data_ptr(),
                      y=y.data_ptr(), pool_partial=pool.data_ptr(), B=B, H=H, W=W, Cin_p=Cp, Cmid_p=P["c3p"], Cout_p=Cp, dtype=cab_code, tile_rows=cab_rows)

        def conv_branch():
            if mid_fused:
                return
            with torch.cuda.stream(side):
                if unfused and not n1_ready:
                    ops.layernorm(t_in, n1, *bp["ln1"], self.embed_dim)
                if cab_fused:
                    ops.cab_fused(**cab_kw)
                else:
                    conv_call(n1, *bp["cab1"], mid, cdt, act=L.ACT_GELU)
                    conv_call(mid, *bp["cab2"], y, cdt, pool=pool, tile_rows=th)
                # conv_scale * sigmoid(squeeze MLP(mean(y))) per (image, channel): consumed by the projection GEMM's gated second residual
                # (sr_swin_tail recomputes it per workgroup from the pool partials instead: one launch less at the end of this branch)
                if not gate_in_tail:
                    ops.channel_gate(gate, pool_partial=pool.data_ptr(), w1=w1.data_ptr(), b1=b1.data_ptr(), w2=w2.data_ptr(), b2=b2.data_ptr(), B=B, H=H, W=W,
                                     C=self.embed_dim, C_p=Cp, Cr=w1.shape[0], n_tiles=n_tiles, y_scale=float(self.conv_scale))

        # Launch ORDER: a captured HIP graph keeps the FIRST-created successor of a node on that node's queue and moves the others to another
        # queue behind an ~8-13 us cross-queue signal.  With the next block's QKV fused into sr_swin_tail the conv branch (one 30-35 us launch
        # beside the attention kernel) is the longer one and is created first (default); SR_HAT_SIDE_FIRST=0 creates the attention branch first
        # and enqueues the conv branch from the projection hook.  Measured on two boxes (HAT x4 b4, ms): fused QKV + conv first 3.42 / 3.43,
        # fused + attention first 3.59 / 3.62, separate QKV + conv first 3.83 / 3.36, separate + attention first 3.85 / 3.42.
        late = unfused and side is not main and False
        fork = None
        if late:
            fork = torch.cuda.Event()
            fork.record(main)
        else:
            if side is not main:
                side.wait_stream(main)
            conv_branch()

        def join():  # the projection is the first launch that needs the conv branch
            if late:
                side.wait_event(fork)
                conv_branch()
            if side is not main:
                main.wait_stream(side)
            d = dict(skip2=y.data_ptr(), skip2_gate=gate.data_ptr(), skip2_dtype=sr_dtype(y.dtype), ldskip2=Cp, gate_rows=H * W, ld_gate=Cp)
            if fuse_next_qkv:
                nb_ = B * H * W // geo.ntok
                d.update(qkv_next=dict(q=ws_.get("hab.q", (nb_, geo.heads, geo.ntok, geo.hd_p), cdt), k=ws_.get("hab.k", (nb_, geo.heads, geo.ntok, geo.hd_p), cdt),
                                       vt=ws_.get("hab.vt", (nb_, geo.heads, geo.hd_p, geo.ntok), cdt), shift=next_bp["shift"],
                                       frag=qkv_frag_order(next_bp, geo, Cp, cdt)))
            if fuse_oca:
                nb_, e_ = B * H * W // geo.ntok, P["border"]
                d.update(qkv_next=dict(q=ws_.get("oca.q", (nb_, geo.heads, geo.ntok, geo.hd_p), cdt), k=ws_.get("oca.k", (B, H + 2 * e_, W + 2 * e_, geo.heads, geo.hd_p), cdt),
                                       vt=ws_.get("oca.vt", (B, geo.heads, geo.hd_p, H + 2 * e_, W + 2 * e_), cdt), shift=0, frag=False, oca_pad=e_,
                                       stream=bp["tail_oca_stream"]))
            if gate_in_tail:
                d.update(ca=dict(pool_partial=pool.data_ptr(), ca_w1=w1.data_ptr(), ca_b1=b1.data_ptr(), ca_w2=w2.data_ptr(), ca_b2=b2.data_ptr(),
                                 ca_Cr=w1.shape[0], ca_n_tiles=n_tiles, y_scale=float(self.conv_scale)))
            # 32-token workgroups while the WHOLE forward (all part batches) is at most one 64-token workgroup per CU: single tile 1.89 -> 1.67 ms, b2 2.22 -> 1.96,
            # b4 2.59 -> 2.55; two parts of four images each (b8) would put 1,024 of them on 512 places: 3.90 -> 4.12 ms
            if getattr(self, "_total_B", B) * H * W // 64 <= int(knob("SR_TAIL_WG32_UPTO", "256")):
                d.update(wg_tokens=32)
            else:
                d.update(wg_tokens=64)
            if next_ln is not None:  # n1's last reader (this block's first conv) has joined: the tail may overwrite it (bf16; fp32 on the split-operand path)
                d.update(n1=n1, n1_ln=next_ln)
            return d

        # attention branch + shortcut (+ conv_scale * CA(cab) in the projection's epilogue) -> t   (hat.py:172-192)
        used = run_window_msa(bp, bp["ln1"], geo, t_in, t, t_in, ws_, cdt, bp["shift"], name="hab", before_proj=join, with_mlp=True, qkv_ready=qkv_ready,
                              attn_launch=(lambda akw: ops.hab_mid(akw, cab_kw)) if mid_fused else None, qkv_n1=qkv_n1, qkv_in_attn=qkv_in_attn)
        if used == "tail":  # projection + both residuals + LayerNorm2 + MLP (+ the next block's LayerNorm1 and QKV) ran as one launch (sr_swin_tail)
            return next_ln is not None, fuse_next_qkv or fuse_oca
        if not used:
            # one-kernel attention half (ws 8 geometries): the combine stays a separate pass over the stream
            if side is not main:
                main.wait_stream(side)
            run_channel_attention(bp["ca"], y, pool, n_tiles, self.embed_dim, t, skip=t, y_scale=float(self.conv_scale))
        run_mlp(bp, bp["ln2"], geo, t, ws_, cdt, name="hab")
        return False, False

    def _run_ocab(self, op: Dict, geo: SwinGeometry, P: Dict, t: Tensor, ws_, cdt, qkv_ready: bool = False) -> None:
        """t = OCAB(t) in place (hat.py:239-293).  qkv_ready: "oca.q" / "oca.k" / "oca.vt" were written by the last HAB's sr_swin_tail."""
        B, H, W, Cp = t.shape
        M = B * H * W
        nb = M // geo.ntok
        sdt = sr_dtype(cdt)
        e = P["border"]
        q = ws_.get("oca.q", (nb, geo.heads, geo.ntok, geo.hd_p), cdt)
        k = ws_.get("oca.k", (B, H + 2 * e, W + 2 * e, geo.heads, geo.hd_p), cdt)  # zero border: allocated zeroed, border never written
        vt = ws_.get("oca.vt", (B, geo.heads, geo.hd_p, H + 2 * e, W + 2 * e), cdt)  # 5-D key: the zero border must never alias another geometry
        o = ws_.get("oca.o", (M, geo.HP), cdt)
        fold = fold_ln(cdt)
        if qkv_ready:
            pass
        elif swin_qkv_usable(op, geo, Cp, cdt) and e % 4 == 0:  # stream form: LayerNorm1 + QKV with k / v^T in the zero-bordered layouts
            ops.swin_qkv(x=t.data_ptr(), q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), wstream=op["qkv_stream"].data_ptr(), B=B, H=H, W=W, C=geo.C, Cp=Cp,
                         ldx=Cp, heads=geo.heads, hd_p=geo.hd_p, ws=geo.ws, shift=0, eps=1e-5, y_mode=L.Y_ROLL, compute_dtype=op["qkv_dtype"], oca_pad=e)
        else:
            ops.gemm(
                A=t.data_ptr(), Wp=op["qkv_w"].data_ptr(), bias=op["qkv_b"].data_ptr(), ln_gamma=None if fold else op["ln1"][0].data_ptr(),
                ln_beta=None if fold else op["ln1"][1].data_ptr(), ln_norm_only=int(fold), out=q.data_ptr(), out_k=k.data_ptr(), out_vt=vt.data_ptr(),
                M=M, K=Cp, N=3 * geo.HP, k_real=geo.C, lda=Cp, a_dtype=L.SR_F32, out_dtype=sdt, compute_dtype=sdt, act=L.ACT_NONE, out_scale=1.0,
                a_map=L.MAP_WINDOW, o_map=L.MAP_IDENTITY, H=H, W=W, ws=geo.ws, shift=0, epi=L.EPI_QKV_OCA, heads=geo.heads, hd_p=geo.hd_p,
                ntok=geo.ntok, ln_eps=1e-5, oca_pad=e,
            )
        ops.oca_attention(
            q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), bias=op["oca_bias"].data_ptr(), out=o.data_ptr(), B=B, H=H, W=W, heads=geo.heads,
            hd_p=geo.hd_p, ws=geo.ws, pad=P["pad"], border=e, nk_pad=P["nk_pad"], dtype=sdt, bias_frag=op["oca_bias_frag"].data_ptr(),
            nk_frag=op["oca_nk_frag"], bias_rel=op["oca_bias_rel"].data_ptr() if ("oca_bias_rel" in op and sdt == L.SR_BF16) else None,
        )
        if swin_tail_usable(op, geo, Cp, cdt):
            run_swin_tail(op, geo, o, t, t, 0, extra=dict(wg_tokens=32 if getattr(self, "_total_B", B) * H * W // 64 <= int(knob("SR_TAIL_WG32_UPTO", "256")) else 64))
            return
        ops.gemm(
            A=o.data_ptr(), Wp=op["proj_w"].data_ptr(), bias=op["proj_b"].data_ptr(), out=t.data_ptr(), skip=t.data_ptr(), M=M, K=geo.HP, N=Cp,
            lda=geo.HP, ldo=Cp, ldskip=Cp, a_dtype=sdt, out_dtype=L.SR_F32, compute_dtype=sdt, act=L.ACT_NONE, out_scale=1.0,
            a_map=L.MAP_IDENTITY, o_map=L.MAP_WINDOW, H=H, W=W, ws=geo.ws, shift=0, epi=L.EPI_STD,
        )
        run_mlp(op, op["ln2"], geo, t, ws_, cdt, name="oca")

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
        object.__setattr__(self, "_total_B", B)  # (launch-form heuristics look at the whole batch, so that half batches take the forms of the one-sequence forward)
        # Inside a HIP-graph capture a large batch runs as two half batches on two streams (as SwinIR's SR_SWIN_PARTS / RCAN's quarter batches): every launch
        # of this model is one or two residency rounds of latency-chain workgroups, and two out-of-phase chains fill each other's rounds (HAT x4 b16 7.61 ->
        # 7.19 ms as two batches of 8 in flight; no gain at b4: 2.62 vs 2.56).  bf16 path only (one queue per half: the other precisions fork a side stream per block).
        # Part batches of FOUR images (each part's launches are then the b4 sizes: 256 tail / 544 mid workgroups, out of phase on up to 16 streams): b16 in 4 parts 6.80 ->
        # 6.64 ms; b32 in 4 / 8 / 16 parts 13.7 / 12.7 / 14.1 ms; 64 tiles in 4 / 8 / 16 parts 27.2 / 26.8 / 24.9 ms
        parts = int(knob("SR_HAT_PARTS", "0")) or (min(16, B // 4) if (B >= 16 and B % 4 == 0 and B % min(16, B // 4) == 0) else 2)
        if parts > 1 and cdt == torch.bfloat16 and B >= int(knob("SR_HAT_PART_MIN", "4")) * parts and B % parts == 0 and x.is_cuda and capturing_or_warming_up():
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
        w = self.window_size
        Hp, Wp = H + (w - H % w) % w, W + (w - W % w) % w  # check_image_size: reflect pad (hat.py:544)
        if Hp - H >= H or Wp - W >= W:
            raise RuntimeError("Padding size should be less than the corresponding input dimension (reflect pad)")
        Cp = P["Cp"]
        f32 = torch.float32
        xin = ws_.get("xin", (B, Hp, Wp, 32), cdt)
        ops.ingest_nchw(x, xin, L.PAD_REFLECT if (Hp != H or Wp != W) else L.PAD_NONE, *P["ing"])
        first = ws_.get("first", (B, Hp, Wp, Cp), f32)
        conv_call(xin, *P["first"], first, cdt)
        ta = ws_.get("ta", (B, Hp, Wp, Cp), f32)
        tb = ws_.get("tb", (B, Hp, Wp, Cp), f32)
        ops.layernorm(first, ta, *P["pe_norm"], self.embed_dim)
        for lp in P["layers"]:
            geo = lp["geo"]
            cur = ta
            ready = qready = False
            for i, bp in enumerate(lp["blocks"]):
                nbp = lp["blocks"][i + 1] if i + 1 < len(lp["blocks"]) else None
                ready, qready = self._run_hab(bp, geo, P, cur, tb, ws_, cdt, n1_ready=ready, next_ln=None if nbp is None else nbp["ln1"],
                                              qkv_ready=qready, next_bp=nbp, next_oca=lp["ocab"] if nbp is None else None)
                cur = tb
            if cur is ta:
                tb.copy_(ta)
                qready = False
            self._run_ocab(lp["ocab"], geo, P, tb, ws_, cdt, qkv_ready=qready and bool(lp["blocks"]))
            conv_call(tb, *lp["conv"], ta, cdt, skip=ta)  # ta = conv(group(ta)) + ta  (hat.py:385)
        normed = ws_.get("normed", (B, Hp, Wp, Cp), cdt)  # read only by the conv, which rounds to the compute dtype anyway
        ops.layernorm(ta, normed, *P["norm"], self.embed_dim)
        body = ws_.get("body", (B, Hp, Wp, Cp), cdt)
        conv_call(normed, *P["after_body"], body, cdt, skip=first)
        feat = ws_.get("feat", (B, Hp, Wp, 64), cdt)
        conv_call(body, *P["before_up"], feat, cdt, act=L.ACT_LRELU)
        up = run_upsampler(P["up"], feat, ws_, cdt, "hat")
        s = self.scale
        conv_call(up, *P["last"], out, cdt, out_mode=L.OUT_FINAL_NCHW, fin=(*P["fin"], self.n_colors, H * s, W * s), cout_p=16)

    # ------------------------------------------------------------------ reference API
    def get_model_config(self) -> Dict:
        config = super().get_model_config()
        config.update(
            dict(
                embed_dim=self.embed_dim, depths=self.depths, num_heads=self.num_heads, window_size=self.window_size, mlp_ratio=self.mlp_ratio,
                drop_rate=self.drop_rate, attn_drop_rate=self.attn_drop_rate, drop_path_rate=self.drop_path_rate,
                compress_ratio=self.compress_ratio, squeeze_factor=self.squeeze_factor, conv_scale=self.conv_scale, overlap_ratio=self.overlap_ratio,
            )
        )
        return config

    @classmethod
    def from_pretrained(cls, scale: int = 4) -> "HAT":
        """hat.py:576-593: HAT_SRx{scale}.pth (key params_ema), read from ./pretrained (no network here)."""
        path = os.path.join("pretrained", f"HAT_SRx{scale}.pth")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found (no network access here; place the official checkpoint there)")
        model = cls(scale=scale)
        model.load_state_dict(torch.load(path, map_location="cpu")["params_ema"])
        return model
