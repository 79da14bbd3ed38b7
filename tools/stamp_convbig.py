"""In-kernel s_memtime stamps of sr_conv3x3_big on the RSTB conv shape (8 x 72 x 72, 192 -> 192, fp32 in / skip / out): needs
SR_EXP_SRC=sr_conv_big tools/exp3.sh convbig_stamps -DSR_STAMPS; SR_LIB_PATH=studiosr_amd/lib/variants/convbig_stamps.so."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd._lib as L
from studiosr_amd import packing
from studiosr_amd.models.common import conv_call

dev = torch.device("cuda")
f = L.lib().sr_debug_convbig_stamps
f.argtypes = [ctypes.c_void_p]
w = torch.randn(192, 192, 3, 3, device=dev) * 0.05
wp, bp = packing.pack_conv3x3(w, torch.randn(192, device=dev), 192, packing.identity_idx(192, 192), torch.bfloat16)
for B, dt in ((8, torch.float32), (8, torch.bfloat16), (1, torch.float32)):
    x = torch.randn(B, 72, 72, 192, device=dev).to(dt)
    skip = torch.randn(B, 72, 72, 192, device=dev)
    out = torch.empty(B, 72, 72, 192, device=dev, dtype=dt)
    for _ in range(3):
        conv_call(x, wp, bp, out, torch.bfloat16, skip=skip if dt == torch.float32 else None)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    f(buf)
    v = [buf[i] for i in range(5)]
    print(f"B={B} {str(dt)[6:]}: skip issue {v[1] - v[0]}  halo staging {v[2] - v[1]}  mfma (9 taps x 6 chunks) {v[3] - v[2]}  epilogue {v[4] - v[3]}  total {v[4] - v[0]} cycles")
