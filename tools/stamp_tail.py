"""In-kernel s_memtime stamps of sr_swin_tail inside a HAT forward (needs a -DSR_STAMPS variant of sr_swin_tail.hip:
SR_EXP_SRC=sr_swin_tail bash tools/exp3.sh tail_stamps -DSR_STAMPS; SR_LIB_PATH=studiosr_amd/lib/variants/tail_stamps.so)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S
import studiosr_amd._lib as L

NAMES = {15: "O gather DMAs issued", 16: "ring loads issued", 1: "x (y, gate) loads issued", 2: "wait loads + bar", 3: "proj (6 steps)", 4: "bias/y + LN2 + bar", 5: "fc1 h0", 6: "gelu h0", 7: "bar", 8: "fc2 h0",
         9: "fc1 h1", 10: "bar + gelu h1", 11: "bar", 12: "fc2 h1", 13: "-", 17: "LN stats + tile + n1 side output", 18: "bar + row store", 19: "LN1 image + dest + 2 bar", 20: "qkv p0 (6 steps)", 21: "stores p0", 22: "qkv p1", 23: "stores p1",
         24: "qkv p2", 25: "stores p2", 14: "end", 26: "x / y loads issued", 27: "ca_squeeze (global) / operand loads (LDS form)", 28: "ca_squeeze from LDS"}
dev = torch.device("cuda")
lib = L.lib()
f = lib.sr_debug_tail_stamps
f.argtypes = [ctypes.c_void_p]
for B in (4, 16):
    m = S.HAT(scale=4, depths=[2], num_heads=[6]).eval().to(dev).set_precision("bf16")
    x = torch.rand(B, 3, 64, 64, device=dev)
    with torch.no_grad():
        for _ in range(3):
            m(x)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    f(buf)
    v = {i: buf[i] for i in range(32) if buf[i]}
    ks = sorted(v, key=lambda i: v[i])
    print(f"B={B} total {v[ks[-1]] - v[ks[0]]} cycles (s_memtime)")
    for a, b in zip(ks[:-1], ks[1:]):
        print(f"   {NAMES.get(b, b):>24s} {v[b] - v[a]:7d}")
