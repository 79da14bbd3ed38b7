"""Where do the __amd_rocclr_copyBuffer launches of a fused HAT training step come from?  torch.profiler with stacks: device-side memcpy / copy kernels
grouped by the host op that enqueued them.  python tools/train_copy_probe.py"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S
from studiosr_amd.optim import Adam
from torch.profiler import profile, ProfilerActivity
dev = "cuda:0"
torch.manual_seed(0)
m = S.HAT(scale=4).to(dev).train()
opt = Adam(m.parameters(), model=m, lr=2e-4, betas=(0.9, 0.99))
x, y = torch.rand(4, 3, 64, 64, device=dev), torch.rand(4, 3, 256, 256, device=dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.l1_loss(m(x), y)
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ev = prof.events()
dev_ev = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
names = collections.Counter(e.name[:70] for e in dev_ev)
print("device events:", len(dev_ev))
for n, c in names.most_common(12): print(f"  {c:5d}  {n}")
cpu_ops = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and ("copy" in e.name.lower() or "memcpy" in e.name.lower() or "memset" in e.name.lower() or "fill" in e.name.lower() or "zero" in e.name.lower()):
        st = [s for s in (e.stack or []) if "studiosr_amd" in s or "bench" in s or "tools/" in s]
        cpu_ops[(e.name, tuple(st[:2]))] += 1
print("host ops that copy / fill:")
for (n, st), c in cpu_ops.most_common(25): print(f"  {c:5d}  {n}   {' <- '.join(st)}")
