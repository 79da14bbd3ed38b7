"""Time one block's nn.Linear weight-gradient launch alone (qkv 576 x 192, proj 192 x 192, fc1 384 x 192, fc2 192 x 384 over T tokens): python tools/wgrad_time.py [T] [ks]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from studiosr_amd import fasttrain as FT

dev = "cuda:0"
bf = torch.bfloat16
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.manual_seed(0)
shapes = [(576, 192), (192, 192), (384, 192), (192, 384)]
ops = [((torch.randn(T, n, device=dev) * 0.5).to(bf), (torch.randn(T, k, device=dev) * 0.5).to(bf), torch.empty(ks, n, k, device=dev)) for n, k in shapes]
jobs = [dict(A=a.data_ptr(), B=b.data_ptr(), out=o.data_ptr(), lda=a.shape[1], ldb=b.shape[1], Np=a.shape[1], Kp=b.shape[1], T=T, taps=1, H=1, W=1, ones_col=-1, ks=ks) for a, b, o in ops]
for _ in range(5):
    FT._wgrad(jobs)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        FT._wgrad(jobs)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
gf = sum(2.0 * T * n * k for n, k in shapes) / 1e9
print(f"T={T} ks={ks}: {best:.1f} us per launch = {gf / best * 1e-3:.0f} TFLOP/s")
