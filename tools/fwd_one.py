#!/usr/bin/env python3
"""A few eval forwards of one model at one precision (for rocprofv3): python tools/fwd_one.py SwinIR 8 fp32"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402

kind, B, prec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
m = getattr(S, kind)(scale=4).eval().cuda().set_precision(prec)
x = torch.rand(B, 3, 64, 64, device="cuda")
with torch.no_grad():
    for _ in range(5):
        m(x)
torch.cuda.synchronize()
