#!/usr/bin/env python3
"""Eval-forward time of the two section-8(f4) models at the SwinIR / RCAN bench shapes: SwinFIR x4 b8 and HAN x4 b16, 64x64 LR tiles."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402

dev = torch.device("cuda")
for kind, B in [("SwinFIR", 8), ("HAN", 16)]:
    for prec in ("bf16", "fp32"):
        torch.manual_seed(0)
        m = getattr(S, kind)(scale=4).eval().to(dev)
        if hasattr(m, "set_precision"):
            m.set_precision(prec)
        x = torch.rand(B, 3, 64, 64, device=dev)
        with torch.no_grad():
            for _ in range(2):
                y = m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                y = m(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{kind} x4 b{B} {prec}: {dt * 1e3:8.2f} ms  {B * 0.065536 / dt:7.1f} HR-Mpix/s  out {tuple(y.shape)}", flush=True)
        del m
