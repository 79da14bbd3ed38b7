#!/usr/bin/env python3
"""Forward throughput of every model family at the BASELINE.json single-GPU shapes (default architectures, bf16 operands,
synthetic weights / inputs, HIP-graph replay): EDSR x4 b16 (config 2), SwinIR x4 b8 (config 3), RCAN x4 b16, HAT x4 b4.
Prints one JSON line per model: ms per forward, HR-Mpix/s, achieved TFLOP/s against the 2.5 PFLOP/s bf16 MFMA peak."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402
from studiosr_amd.runtime import GraphedForward  # noqa: E402

GFLOP_PER_TILE = {"EDSR": 411.67, "SwinIR": 135.56, "RCAN": 130.40, "HAT": 207.76}  # BASELINE.md section 2 (64x64 LR tile, x4)
CASES = {"EDSR": 16, "SwinIR": 8, "RCAN": 16, "HAT": 4}


def main():
    which = sys.argv[1:] or list(CASES)
    dev = torch.device("cuda")
    for kind in which:
        kind, _, bs = kind.partition(":")  # "HAT:16" overrides the batch size
        B = int(bs) if bs else CASES[kind]
        torch.manual_seed(0)
        m = getattr(S, kind)(scale=4).eval().to(dev).set_precision("bf16")
        x = torch.rand(B, 3, 64, 64, device=dev)
        with torch.no_grad():
            g = GraphedForward(m, x)
            for _ in range(3):
                g(x)
            torch.cuda.synchronize()
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                g(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        tf = B * GFLOP_PER_TILE[kind] / dt / 1e3
        print(json.dumps({"model": f"{kind} x4", "batch": B, "ms": round(dt * 1e3, 3), "hr_mpix_per_s": round(B * 256 * 256 / 1e6 / dt, 1),
                          "tflops": round(tf, 1), "frac_bf16_mfma_peak": round(tf / 2500.0, 4)}), flush=True)
        del g, m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
