#!/usr/bin/env python3
"""RCAN x4 forward with and without the two-half-batch pipelining (same box): python tools/rcan_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402
from studiosr_amd.runtime import GraphedForward  # noqa: E402

dev = torch.device("cuda")
for B in (8, 16, 32):
    for flag in (False, True):
        torch.manual_seed(0)
        m = S.RCAN(scale=4).eval().to(dev).set_precision("bf16")
        m.pipeline_halves = flag
        x = torch.rand(B, 3, 64, 64, device=dev)
        with torch.no_grad():
            gf = GraphedForward(lambda t: m(t), x)
            for _ in range(3):
                gf.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                gf.replay()
            torch.cuda.synchronize()
        print(f"RCAN x4 b{B} halves={flag}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
