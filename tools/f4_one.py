#!/usr/bin/env python3
"""One model of tools/f4_bench.py, a few forwards (for rocprofv3): python tools/f4_one.py SwinFIR 8"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402

kind, B = sys.argv[1], int(sys.argv[2])
m = getattr(S, kind)(scale=4).eval().cuda().set_precision("bf16")
x = torch.rand(B, 3, 64, 64, device="cuda")
with torch.no_grad():
    for _ in range(5):
        m(x)
torch.cuda.synchronize()
