"""In-kernel s_memtime stamps of the round-3 Swin block kernel (needs a -DSR_STAMPS variant: SR_LIB_PATH=studiosr_amd/lib/variants/stamps.so)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S
import studiosr_amd._lib as L
from studiosr_amd.models import swinir as SW

NAMES = {0: "start", 1: "x load + LN1", 2: "bar", 27: "LN2", 28: "bar", 30: "fc1 h0", 31: "gelu h0", 32: "bar", 33: "fc2 h0", 35: "fc1 h1", 36: "gelu h1(+bar)",
         37: "bar", 38: "fc2 h1", 40: "-", 41: "store"}
for p in range(3):
    NAMES.update({3 + 8 * p: f"p{p} qkv gemm", 4 + 8 * p: f"p{p} bias+bar+write", 5 + 8 * p: f"p{p} bar", 6 + 8 * p: f"p{p} attn", 7 + 8 * p: f"p{p} bar", 8 + 8 * p: f"p{p} proj"})
dev = torch.device("cuda")
cdt = torch.bfloat16
lib = L.lib()
m = S.SwinIR(scale=4, depths=[2], num_heads=[6]).eval().to(dev).set_precision("bf16")
P = m._get_packed(cdt)
lp = P["layers"][0]
geo, bp = lp["geo"], lp["blocks"][1]
f = lib.sr_debug_sw3_stamps
f.argtypes = [ctypes.c_void_p]
for B in (1, 8):
    t = torch.randn(B, 72, 72, geo.Cp, device=dev)
    t[..., geo.C:] = 0
    o = torch.empty_like(t)
    ws_ = S.runtime.Workspace(dev)
    for _ in range(3):
        SW.run_swin_block(bp, geo, t, o, ws_, cdt, bp["shift"])
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    f(buf)
    v = {i: buf[i] for i in range(64) if buf[i]}
    ks = sorted(v, key=lambda i: v[i])
    print(f"B={B} total {v[ks[-1]] - v[ks[0]]} cycles")
    for a, b in zip(ks[:-1], ks[1:]):
        print(f"   {NAMES.get(b, b):>18s} {v[b] - v[a]:7d}")
