#!/usr/bin/env python3
"""Launch time of the 16 x 16 window attention (hat.py:85-110) in its forms -- register-only flash (row-major / fragment-order operands), LDS form
(SrWindowAttn.bias_tiles), and inside sr_hab_mid beside the CAB -- at HAT x4 tile batches: python tools/attn_bench.py [B ...] (default 4 16 64)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from studiosr_amd import _lib as L, ops, packing  # noqa: E402
from studiosr_amd.models.hat import rpi_sa  # noqa: E402

DEV = torch.device("cuda:0")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    heads, hd_p, ws, H, W, C, Cp, c3, c3p = 6, 32, 16, 64, 64, 180, 192, 60, 64
    ntok = ws * ws
    torch.manual_seed(0)
    table = torch.randn((2 * ws - 1) ** 2, heads, device=DEV)
    bias = packing.gather_bias(table, rpi_sa(ws), ntok, ntok)
    bias_frag, tiles = packing.bias_fragments(bias), packing.bias_distinct_tiles(bias)
    w1, b1 = torch.randn(c3, C, 3, 3, device=DEV) * 0.03, torch.randn(c3, device=DEV) * 0.1
    w2, b2 = torch.randn(C, c3, 3, 3, device=DEV) * 0.05, torch.randn(C, device=DEV) * 0.1
    p1 = packing.pack_conv3x3(w1, b1, Cp, packing.identity_idx(c3, c3p), torch.bfloat16)
    p2 = packing.pack_conv3x3(w2, b2, c3p, packing.identity_idx(C, Cp), torch.bfloat16)
    for B in [int(v) for v in sys.argv[1:]] or [4, 16, 64]:
        nb = B * (H // ws) * (W // ws)
        q = (torch.randn(nb, heads, ntok, hd_p, device=DEV) * 0.4).to(torch.bfloat16)
        k = torch.randn(nb, heads, ntok, hd_p, device=DEV).to(torch.bfloat16)
        vt = torch.randn(nb, heads, hd_p, ntok, device=DEV).to(torch.bfloat16)
        o = torch.empty(nb * ntok, heads * hd_p, device=DEV, dtype=torch.bfloat16)
        x = torch.randn(B, H, W, Cp, device=DEV).to(torch.bfloat16)
        y = torch.empty_like(x)
        pool = torch.empty(B, ops.cab_pool_tiles(H, W), Cp, device=DEV)
        ckw = dict(x=x.data_ptr(), w1p=p1[0].data_ptr(), b1=p1[1].data_ptr(), w2p=p2[0].data_ptr(), b2=p2[1].data_ptr(), y=y.data_ptr(),
                   pool_partial=pool.data_ptr(), B=B, H=H, W=W, Cin_p=Cp, Cmid_p=c3p, Cout_p=Cp, dtype=L.SR_BF16)
        row = {"tiles": B, "windows": nb}
        for shift in (0, 8):
            for name, frag, lds in (("flash", 0, 0), ("flash_frag", 1, 0), ("lds", 0, 1), ("lds_frag", 1, 1)):
                akw = dict(q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), bias=bias.data_ptr(), out=o.data_ptr(), n_bwin=nb, heads=heads, hd_p=hd_p, ntok=ntok,
                           H=H, W=W, ws=ws, shift=shift, dtype=L.SR_BF16, y_mode=L.Y_ROLL, bias_frag=bias_frag.data_ptr(), qkv_frag=frag,
                           bias_tiles=tiles.data_ptr() if lds else None)
                row[f"{name}_s{shift}_us"] = round(timed(lambda: ops.window_attention(**akw)), 1)
                if frag:
                    row[f"mid_{name}_s{shift}_us"] = round(timed(lambda: ops.hab_mid(akw, ckw)), 1)
        row["cab_us"] = round(timed(lambda: ops.cab_fused(**ckw)), 1)
        gf = 4.0 * ntok * ntok * 30 * heads * nb / 1e9
        row["attn_gflop"] = round(gf, 2)
        row["lds_frag_tflops"] = round(gf / row["lds_frag_s0_us"] * 1e3 / 1e3, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
