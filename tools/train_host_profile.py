"""cProfile of the host side of the fused HAT training step (where the ~16 ms of enqueue time per step go): python tools/train_host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402
from studiosr_amd.optim import Adam  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = S.HAT(scale=4).to(dev).train()
opt = Adam(m.parameters(), model=m, lr=2e-4, betas=(0.9, 0.99))
x, y = torch.rand(4, 3, 64, 64, device=dev), torch.rand(4, 3, 256, 256, device=dev)


def step():
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.l1_loss(m(x), y)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
