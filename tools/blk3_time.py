#!/usr/bin/env python3
"""Launch time of the Swin block kernel(s) at the bench shape: python tools/blk3_time.py [v2 v3]  (SR_LIB_PATH selects a variant library)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import studiosr_amd as S  # noqa: E402
from studiosr_amd.models import swinir as SW  # noqa: E402
from tools.blk3_check import timeit  # noqa: E402

dev = torch.device("cuda")
m = S.SwinIR(scale=4, depths=[2], num_heads=[6]).eval().to(dev).set_precision("bf16")
cdt = torch.bfloat16
lp = m._get_packed(cdt)["layers"][0]
geo, bp = lp["geo"], lp["blocks"][1]
ws_ = S.runtime.Workspace(dev)
kinds = ["v3"]  # (one kernel since C ABI v10; the argument is accepted for old command lines)
for B in [int(b) for b in os.environ.get("SR_BS", "1,8,16").split(",")]:
    t = torch.randn(B, 72, 72, geo.Cp, device=dev)
    t[..., geo.C:] = 0
    o = torch.empty_like(t)
    res = {}
    for k in kinds:
        res[k] = min(timeit(lambda: SW.run_swin_block(bp, geo, t, o, ws_, cdt, bp["shift"]), iters=30) for _ in range(3))
    print(f"B={B:2d} windows={B * 81:5d} " + " ".join(f"{k}={v:7.1f}us" for k, v in res.items()), flush=True)
