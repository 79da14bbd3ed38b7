#!/usr/bin/env python3
"""Training-step timing of the HIP training path: HAT / SwinIR / EDSR / RCAN at BASELINE shapes (forward, backward, Adam), with a
per-phase split from HIP events.  python tools/train_bench.py [KIND[:B]] ..."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402


def main():
    dev = torch.device("cuda")
    for spec in sys.argv[1:] or ["HAT:4", "SwinIR:4", "EDSR:4", "RCAN:4"]:
        kind, _, b = spec.partition(":")
        b = int(b or 4)
        torch.manual_seed(0)
        m = getattr(S, kind)(scale=4).to(dev).train()
        from studiosr_amd.optim import Adam  # what studiosr_amd.Trainer builds (torch.optim.Adam; one flat launch on the fused path)

        opt = Adam(m.parameters(), model=m, lr=2e-4, betas=(0.9, 0.99))
        x, y = torch.rand(b, 3, 64, 64, device=dev), torch.rand(b, 3, 256, 256, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ac = os.environ.get("AUTOCAST", "1") != "0"  # the reference Trainer's bf16 autocast context (AUTOCAST=0: exact fp32 everywhere)
        for it in range(int(os.environ.get("TRAIN_STEPS", "4"))):
            ev[0].record()
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=ac):
                out = m(x)
                loss = F.l1_loss(out, y)
            ev[1].record()
            loss.backward()
            ev[2].record()
            opt.step()
            opt.zero_grad(set_to_none=True)
            ev[3].record()
            torch.cuda.synchronize()
        t = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
        print(f"{kind} x4 b{b} {'autocast-bf16' if ac else 'fp32'}: forward {t[0]:.1f} ms  backward {t[1]:.1f} ms  adam {t[2]:.1f} ms  total {sum(t):.1f} ms  "
              f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  loss {loss.item():.4f}", flush=True)
        del m, opt
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
