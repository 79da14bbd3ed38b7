#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at the bench shapes (HIP events, back-to-back launches)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import studiosr_amd as S  # noqa: E402
import studiosr_amd._lib as L  # noqa: E402
from studiosr_amd import ops  # noqa: E402
from studiosr_amd.models import swinir as SW  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    torch.manual_seed(0)
    dev = torch.device("cuda")
    m = S.SwinIR(scale=4, depths=[2], num_heads=[6]).eval().to(dev).set_precision("bf16")
    cdt = torch.bfloat16
    P = m._get_packed(cdt)
    lp = P["layers"][0]
    geo, bp = lp["geo"], lp["blocks"][1]
    which = sys.argv[1:] or ["mlp", "qkv", "attn", "proj", "msa"]
    for B in ((1, 8) if "mlpab" in which else (1, 2, 4, 8, 16)):
        H = W = 72
        t = torch.randn(B, H, W, geo.Cp, device=dev)
        t[..., geo.C:] = 0
        ws_ = S.runtime.Workspace(dev)
        M = B * H * W
        res = {}
        if "mlp" in which:
            res["mlp"] = timeit(lambda: SW.run_mlp(bp, bp["ln2"], geo, t, ws_, cdt))
        if "mlpab" in which:
            for fl in (0, 1, 2, 4, 5, 7):
                res[f"mlp_f{fl}"] = timeit(lambda: ops.mlp_fused(
                    x=t.data_ptr(), out=t.data_ptr(), ln_gamma=None, ln_beta=None, w1p=bp["fc1_w"].data_ptr(), b1=bp["fc1_b"].data_ptr(),
                    w2p=bp["fc2_w"].data_ptr(), b2=bp["fc2_b"].data_ptr(), M=M, C=geo.C, Cp=geo.Cp, Hp=geo.hid_p, ldx=geo.Cp, eps=1e-5, debug_flags=fl))
        if "msa" in which:
            res["msa"] = timeit(lambda: SW.run_window_msa(bp, bp["ln1"], geo, t, t, t, ws_, cdt, bp["shift"]))
        print(f"B={B:2d} M={M:6d} WG64={M // 64:5d} " + " ".join(f"{k}={v:8.1f}us" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
