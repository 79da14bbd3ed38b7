"""In-kernel s_memtime stamps of the gated sr_rcab kernel inside an RCAN forward (needs a -DSR_STAMPS variant of sr_rcab.hip:
SR_EXP_SRC=sr_rcab bash tools/exp3.sh rcab_stamps -DSR_STAMPS; SR_LIB_PATH=studiosr_amd/lib/variants/rcab_stamps.so)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S
import studiosr_amd._lib as L

NAMES = {1: "weights + first halo loads issued", 2: "gate squeeze (4 barriers)", 3: "halo commit, 2nd round load + commit", 4: "barrier", 5: "conv1 (18 steps)",
         6: "bias + ReLU -> mid image, barrier", 7: "conv2 (18 steps)", 8: "y stores (7 rows through the private tile)", 9: "pool partials"}
dev = torch.device("cuda")
f = L.lib().sr_debug_rcab_stamps
f.argtypes = [ctypes.c_void_p]
for B in (8, 16):
    m = S.RCAN(scale=4, n_resgroups=1, n_resblocks=4).eval().to(dev).set_precision("bf16")
    x = torch.rand(B, 3, 64, 64, device=dev)
    with torch.no_grad():
        for _ in range(3):
            m(x)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    f(buf)
    v = {i: buf[i] for i in range(16) if buf[i]}
    ks = sorted(v, key=lambda i: v[i])
    print(f"B={B} total {v[ks[-1]] - v[ks[0]]} cycles (s_memtime)")
    for a, b in zip(ks[:-1], ks[1:]):
        print(f"   {NAMES.get(b, b):>40s} {v[b] - v[a]:7d}")
