#!/usr/bin/env python3
"""Run model forwards with guard zones around every workspace buffer (SR_WS_GUARD=1) and report out-of-bounds stores."""
import os, sys
os.environ["SR_WS_GUARD"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S

torch.manual_seed(0)
dev = "cuda"
cases = [("SwinIR", S.SwinIR(scale=4), (8, 3, 64, 64)), ("SwinIR-small", S.SwinIR(scale=4), (1, 3, 13, 17)), ("EDSR", S.EDSR(scale=4, n_feats=64, n_resblocks=2), (2, 3, 20, 36)),
         ("EDSR-wide", S.EDSR(scale=4, n_resblocks=2), (6, 3, 50, 70)), ("HAT-w16", S.HAT(scale=4, depths=[2], num_heads=[6]), (3, 3, 64, 48)),
         ("HAT", S.HAT(scale=2, embed_dim=60, depths=[2], num_heads=[6], window_size=8), (1, 3, 24, 16)), ("RCAN", S.RCAN(scale=3, n_feats=32, n_resblocks=2, n_resgroups=2, reduction=8), (2, 3, 12, 12)),
         ("RCAN-64 (chained RCABs)", S.RCAN(scale=2, n_feats=64, n_resblocks=3, n_resgroups=2), (3, 3, 33, 45)),
         ("RCAN-64 b16 (two half batches)", S.RCAN(scale=2, n_feats=64, n_resblocks=3, n_resgroups=1), (16, 3, 20, 24)),
         ("EDSR-256 tail (narrow conv, K split)", S.EDSR(scale=2, n_resblocks=1), (2, 3, 19, 50)),
         ("HAN", S.HAN(scale=2, n_feats=64, n_resblocks=2, n_resgroups=2), (2, 3, 24, 20)),
         ("SwinFIR", S.SwinFIR(scale=2, embed_dim=60, depths=[2], num_heads=[6]), (2, 3, 20, 28))]
for name, m, shp in cases:
    m = m.to(dev).eval()
    for prec in ("bf16", "fp32"):
        m.set_precision(prec)
        x = torch.rand(*shp, device=dev)
        with torch.no_grad():
            y = m(x)
        torch.cuda.synchronize()
        bad = m._ws.check_guards()
        print(name, prec, tuple(y.shape), "finite", bool(torch.isfinite(y).all()), "GUARD VIOLATIONS:" if bad else "guards ok", bad, flush=True)
