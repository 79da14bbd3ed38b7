#!/usr/bin/env python3
"""Quick GPU diagnostic: HIP models vs the oracle on the golden fixtures (prints error tables)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import studiosr_amd as S  # noqa: E402
from oracle import models as OM  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
CT = {"SwinIR": S.SwinIR, "EDSR": S.EDSR, "RCAN": S.RCAN}
if hasattr(S, "HAT"):
    CT["HAT"] = S.HAT


def load(name):
    with np.load(os.path.join(G, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def run(name, kind):
    g = load(name)
    cfg = json.loads(str(g["cfg"]))
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    m = CT[kind](**cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    for k in sorted(g):
        if not k.startswith("y_"):
            continue
        mode, b, h, w = k[2:].split("_")
        x = torch.from_numpy(g[f"x_{b}_{h}_{w}"])
        m.train(mode == "train")
        ref = torch.from_numpy(g[k])
        for prec in ("fp32", "bf16"):
            m.set_precision(prec)
            with torch.no_grad():
                y = m(x.cuda()).cpu()
            err = (y - ref).abs().max().item()
            rel = err / max(ref.abs().max().item(), 1e-9)
            print(f"{name:24s} {k:22s} {prec}: max|d|={err:.3e} rel={rel:.3e} shape={tuple(y.shape)}", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    cases = [
        ("f11_edsr_x2", "EDSR"), ("f11_edsr_x3", "EDSR"), ("f11_edsr_x4", "EDSR"), ("f11_edsr_r255_x2", "EDSR"),
        ("f11_rcan_x4", "RCAN"), ("f11_rcan_x3", "RCAN"),
        ("f11_swinir_x2", "SwinIR"), ("f11_swinir_x3", "SwinIR"), ("f11_swinir_x4", "SwinIR"),
        ("f11_swinir_direct_x4", "SwinIR"), ("f11_swinir_c180_x4", "SwinIR"),
        ("f11_hat_w8_x4", "HAT"), ("f11_hat_w16_x2", "HAT"),
    ]
    only = sys.argv[1:]
    for name, kind in cases:
        if kind not in CT or (only and not any(o in name for o in only)):
            continue
        t = time.time()
        try:
            run(name, kind)
        except Exception as e:  # keep going: this is a diagnostic
            print(f"{name}: EXC {type(e).__name__}: {e}", flush=True)
        print(f"  ({time.time() - t:.1f}s)", flush=True)
