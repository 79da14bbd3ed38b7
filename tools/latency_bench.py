#!/usr/bin/env python3
"""Single-image latency of Model.inference (uint8 HWC -> uint8 HWC), the Evaluator's call pattern: wall time per call against the GPU time
of the same launches (HIP events), i.e. how much of a call is host-side dispatch.  python tools/latency_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402

dev = torch.device("cuda")
img = np.random.default_rng(0).integers(0, 256, size=(64, 64, 3)).astype(np.uint8)
for kind, kw in [("EDSR", dict(scale=2, n_feats=64, n_resblocks=16, res_scale=1.0)), ("SwinIR", dict(scale=4)), ("HAT", dict(scale=4)), ("RCAN", dict(scale=4))]:
    for prec in ("auto", "bf16"):
        m = getattr(S, kind)(**kw).eval().to(dev).set_precision(prec)
        for _ in range(3):
            m.inference(img)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        n = 10
        for _ in range(n):
            out = m.inference(img)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        print(f"{kind:7s} {prec:5s}: {wall:7.2f} ms per inference() call (events {e0.elapsed_time(e1) / n:7.2f} ms)  out {out.shape}", flush=True)
