"""HAT x4 training step (BASELINE config 5) captured in ONE HIP graph: forward + L1 + backward + Adam.  python tools/train_graph.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = S.HAT(scale=4).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.9, 0.99), fused=True, capturable=True)
x, y = torch.rand(4, 3, 64, 64, device=dev), torch.rand(4, 3, 256, 256, device=dev)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.l1_loss(m(x), y)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t0 = time.perf_counter()
for _ in range(n):
    l = step()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step loss {l.item():.4f}")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
print(f"graph: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step loss {static_loss.item():.4f}")
