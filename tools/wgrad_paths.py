#!/usr/bin/env python3
"""Which weight-gradient GEMMs of a HAT x4 step take the atomic-free path (token slices + sr_batch_sum) and which still split K with atomics."""
import collections
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402
from studiosr_amd import autograd as AG  # noqa: E402

c = collections.Counter()
orig = AG.wgrad


def spy(dy, x, rows, cols, T):
    ks = AG._ksplit(rows, cols, T)
    chunk = T // ks if ks > 1 and T % ks == 0 else 0
    c[("slices" if chunk and chunk % 32 == 0 else "atomics", rows, cols, T, ks)] += 1
    return orig(dy, x, rows, cols, T)


AG.wgrad = spy
dev = torch.device("cuda")
m = S.HAT(scale=4).to(dev).train()
x, y = torch.rand(4, 3, 64, 64, device=dev), torch.rand(4, 3, 256, 256, device=dev)
with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
    loss = F.l1_loss(m(x), y)
loss.backward()
torch.cuda.synchronize()
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(v, k)
