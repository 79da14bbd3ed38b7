#!/usr/bin/env python3
"""A/B of the two Swin block kernels (SR_SWIN_BLOCK=v1: 12 waves / 2 windows, round 1; default v2: 4 waves / 1 window) in ONE
process: same inputs, max |diff| of the outputs, interleaved HIP-event timings (median / min of rounds)."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import studiosr_amd as S  # noqa: E402
from studiosr_amd.models import swinir as SW  # noqa: E402


def main():
    torch.manual_seed(0)
    dev = torch.device("cuda")
    m = S.SwinIR(scale=4, depths=[2], num_heads=[6]).eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.1)
    m = m.to(dev).set_precision("bf16")
    cdt = torch.bfloat16
    P = m._get_packed(cdt)
    lp = P["layers"][0]
    geo = lp["geo"]
    ws_ = S.runtime.Workspace(dev)
    for B in (1, 8, 16):
        H = W = 72
        x = torch.randn(B, H, W, geo.Cp, device=dev)
        x[..., geo.C:] = 0
        outs = {}
        for bi in (0, 1):
            bp = lp["blocks"][bi]
            for ver in ("v1", "v2"):
                os.environ["SR_SWIN_BLOCK"] = ver
                o = torch.empty_like(x)
                SW.run_swin_block(bp, geo, x, o, ws_, cdt, bp["shift"])
                torch.cuda.synchronize()
                outs[(bi, ver)] = o
            d = (outs[(bi, "v1")] - outs[(bi, "v2")]).abs().max().item()
            r = outs[(bi, "v1")].abs().max().item()
            print(f"B={B} shift={bp['shift']}: max|v1-v2| = {d:.3e} (range {r:.2f}) pad={outs[(bi, 'v2')][..., geo.C:].abs().max().item():.1e}", flush=True)
        bp = lp["blocks"][1]
        t = {"v1": [], "v2": []}
        o = torch.empty_like(x)
        for rnd in range(7):
            for ver in ("v1", "v2"):
                os.environ["SR_SWIN_BLOCK"] = ver
                for _ in range(3):
                    SW.run_swin_block(bp, geo, x, o, ws_, cdt, bp["shift"])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    SW.run_swin_block(bp, geo, x, o, ws_, cdt, bp["shift"])
                e1.record()
                torch.cuda.synchronize()
                t[ver].append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"B={B}: " + "  ".join(f"{v}: median {statistics.median(ts):.1f} us min {min(ts):.1f} us" for v, ts in t.items()), flush=True)


if __name__ == "__main__":
    main()
