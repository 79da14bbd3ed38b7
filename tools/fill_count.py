#!/usr/bin/env python3
"""Where do the zero-fill launches of a training step come from?  One HAT x4 step under torch.profiler (all threads, shapes recorded):
aten::fill_ / aten::zero_ calls grouped by tensor shape."""
import collections
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402

dev = torch.device("cuda")
m = S.HAT(scale=4).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=2e-4)
x, y = torch.rand(4, 3, 64, 64, device=dev), torch.rand(4, 3, 256, 256, device=dev)


def step():
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        loss = F.l1_loss(m(x), y)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::clone", "aten::copy_", "aten::contiguous"):
        c[(e.name, str(e.input_shapes)[:70])] += 1
tot = collections.Counter()
for (n, s), v in c.items():
    tot[n] += v
print(dict(tot))
for (n, s), v in c.most_common(25):
    print(f"{v:5d} {n:18s} {s}")
