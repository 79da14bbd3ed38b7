"""Probe: do 3 torch.cuda graphs (each ~300 kernel nodes, no custom kernels) captured on side streams replay cleanly?"""
import copy, sys, torch
sys.path.insert(0, '/root/repo')
from studiosr_amd.runtime import GraphedForward
layers = []
for i in range(100):
    layers += [torch.nn.Conv2d(32 if i else 3, 32, 3, padding=1), torch.nn.GELU(), torch.nn.GroupNorm(4, 32)]
m0 = torch.nn.Sequential(*layers).cuda().eval().to(torch.bfloat16)
x = torch.rand(8, 3, 72, 72, device='cuda', dtype=torch.bfloat16)
pipes = []
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for i in range(n):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st), torch.no_grad():
        gf = GraphedForward((lambda inp: m0(inp)), x)
    pipes.append((gf, st))
    torch.cuda.synchronize(); print("captured", i, flush=True)
for it in range(12):
    f, st = pipes[it % len(pipes)]
    with torch.cuda.stream(st):
        f.replay()
    torch.cuda.synchronize(); print("replayed", it, flush=True)
print("ok", len(pipes), float(pipes[0][0].static_out.float().sum()))
