#!/usr/bin/env python3
"""Whole-image forward throughput (what Evaluator / inference() run): one LR image of size S x S, bf16 operands, eager and HIP-graph replay.
python tools/image_bench.py [KIND] [sizes...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402
from studiosr_amd.runtime import GraphedForward  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "SwinIR"
sizes = [int(v) for v in sys.argv[2:]] or [128, 256, 512, 1024]
dev = torch.device("cuda", 0)
m = getattr(S, kind)(scale=4).eval().to(dev).set_precision("bf16")
for s in sizes:
    x = torch.rand(1, 3, s, s, device=dev)
    with torch.no_grad():
        for _ in range(2):
            m(x)
        torch.cuda.synchronize()
        n = max(3, min(20, int(2e6 / (s * s))))
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n
        g = GraphedForward(lambda t: m(t), x)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / n
        del g
    m.release_workspace() if hasattr(m, "release_workspace") else None
    torch.cuda.empty_cache()
    hr = s * s * 16 / 1e6
    print(json.dumps({"model": f"{kind} x4", "lr_size": s, "eager_ms": round(eager * 1e3, 2), "graph_ms": round(graph * 1e3, 2),
                      "hr_mpix_per_s_eager": round(hr / eager, 1), "hr_mpix_per_s_graph": round(hr / graph, 1)}), flush=True)
