#!/usr/bin/env python3
"""sr_colsum (bias gradients / pooling) at the shapes of a HAT x4 training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from studiosr_amd import autograd as AG  # noqa: E402
from bgemm_bench import timeit  # noqa: E402

dev = torch.device("cuda")
for (nb, P, C) in [(1, 16384, 540), (1, 16384, 180), (1, 16384, 360), (1, 16384, 60), (4, 4096, 180), (1, 65536, 256), (1, 262144, 64), (1, 262144, 3)]:
    x = torch.randn(nb, P, C, device=dev)
    out = torch.zeros(nb, C, device=dev)
    us = timeit(lambda: AG.colsum(x, out, nb, P, C))
    ref = x.sum(dim=1)
    out.zero_()
    AG.colsum(x, out, nb, P, C)
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"colsum nb={nb} P={P} C={C}: {us:7.1f} us  {x.numel() * 4 / us / 1e3:7.0f} GB/s  rel err {err:.1e}", flush=True)
