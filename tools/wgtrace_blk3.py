"""Per-workgroup start / end times (s_memrealtime, 100 MHz) and placement of the round-3 block kernel: needs SR_LIB_PATH=.../wgtrace.so (-DSR_WGTRACE)."""
import ctypes
import os
import sys
from collections import Counter

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S
import studiosr_amd._lib as L
from studiosr_amd.models import swinir as SW

import bench  # noqa: E402  (the hash bench.py checks the profile against)

print(f"# kernel_src_sha16 = {bench.dominant_kernel_src_sha16()}")
dev = torch.device("cuda")
cdt = torch.bfloat16
m = S.SwinIR(scale=4, depths=[2], num_heads=[6]).eval().to(dev).set_precision("bf16")
lp = m._get_packed(cdt)["layers"][0]
geo, bp = lp["geo"], lp["blocks"][1]
f = L.lib().sr_debug_sw3_wgtrace
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
for B in (8, 16):
    n = B * 81
    t = torch.randn(B, 72, 72, geo.Cp, device=dev)
    t[..., geo.C:] = 0
    o = torch.empty_like(t)
    ws_ = S.runtime.Workspace(dev)
    for _ in range(5):
        SW.run_swin_block(bp, geo, t, o, ws_, cdt, bp["shift"])
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (6 * n))()
    f(buf, n)
    a = np.array(buf, dtype=np.uint64).reshape(n, 6)
    st, en = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    hwid = (a[:, 2] & 0xFFFFFFFF).astype(np.int64)
    xcc = (a[:, 2] >> 32).astype(np.int64) & 0xF
    cu = (hwid >> 8) & 0xF
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 10 + cu  # XCC / SE / SH / CU
    t0 = st.min()
    print(f"B={B}: {n} workgroups; kernel span {(en.max() - t0) / 100:.1f} us; starts within {(st.max() - t0) / 100:.1f} us; lifetime median {(np.median(en - st)) / 100:.1f} us, min {((en - st).min()) / 100:.1f}, max {((en - st).max()) / 100:.1f}")
    cyc = (a[:, 4] - a[:, 3]).astype(np.int64)
    ghz = cyc / ((en - st) * 10.0)  # s_memtime cycles per ns of s_memrealtime (100 MHz)
    print(f"   B={B} shader clock held under the kernel: clock_ghz_median = {np.median(ghz):.3f}  (min {ghz.min():.3f}, max {ghz.max():.3f}; lifetime median {np.median(cyc)} cycles)")
    per = Counter(key.tolist())
    print("   distinct CUs", len(per), " workgroups per CU histogram", sorted(Counter(per.values()).items()))
    for k in sorted(set(per.values())):
        sel = np.array([per[x] == k for x in key.tolist()])
        print(f"   CUs holding {k}: workgroup lifetime median {np.median((en - st)[sel]) / 100:.1f} us, last end {(en[sel].max() - t0) / 100:.1f} us")
    late = st > t0 + 300
    print(f"   workgroups starting later than 3 us: {late.sum()}; their lifetime median {np.median((en - st)[late]) / 100 if late.any() else 0:.1f} us")
