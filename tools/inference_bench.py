#!/usr/bin/env python3
"""What Evaluator(model.inference) runs (evaluator.py:45-50, common.py:36-48): model.inference(uint8 HWC image) at the DEFAULT precision, per call, eager.
python tools/inference_bench.py [KIND ...] (INF_SIZE=H,W of the LR image, default 128,128; INF_ENSEMBLE=1: inference_with_self_ensemble)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402

kinds = sys.argv[1:] or ["SwinIR", "HAT", "EDSR", "RCAN"]
H, W = (int(v) for v in os.environ.get("INF_SIZE", "128,128").split(","))
ens = os.environ.get("INF_ENSEMBLE", "0") == "1"
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
for kind in kinds:
    m = getattr(S, kind)(scale=4).eval().to(dev)
    f = m.inference_with_self_ensemble if ens else m.inference
    for _ in range(2):
        out = f(img)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        out = f(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{kind} x4 inference({H}x{W} uint8{', self-ensemble' if ens else ''}) -> {tuple(out.shape)}: {dt * 1e3:.2f} ms per call = {out.shape[0] * out.shape[1] / dt / 1e6:.1f} HR-Mpix/s", flush=True)
