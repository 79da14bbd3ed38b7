"""Throughput of the reference-precision paths at the BASELINE shapes: 'fp32' (exact fp32 MFMA, what inference() uses outside autocast),
'fp32x3' (split-operand bf16 with fp32 accumulation) and 'bf16', with the error of each against the exact path."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S

dev = torch.device("cuda")
for kind, B in (("SwinIR", 8), ("EDSR", 16), ("HAT", 4), ("RCAN", 16)):
    torch.manual_seed(0)
    m = getattr(S, kind)(scale=4).eval().to(dev)
    x = torch.rand(B, 3, 64, 64, device=dev)
    ref = None
    for prec in ("fp32", "fp32x3", "bf16"):
        m.set_precision(prec)
        with torch.no_grad():
            for _ in range(2):
                y = m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                m(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
        if ref is None:
            ref = y.clone()
        err = float((y - ref).abs().max())
        mse = float(((y - ref) ** 2).mean())
        psnr = float("inf") if mse == 0 else 10 * torch.log10(torch.tensor(1.0 / mse)).item()
        print(f"{kind} x4 b{B} {prec:7s}: {dt * 1e3:8.2f} ms = {B * 0.065536 / dt:7.1f} HR-Mpix/s   max|d| vs exact {err:.2e}  PSNR {psnr:.1f} dB", flush=True)
