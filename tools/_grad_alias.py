import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import studiosr_amd as S
dev = "cuda:0"
torch.manual_seed(0)
m = S.HAT(scale=4, depths=[2], num_heads=[6]).to(dev).train()
x, y = torch.rand(2, 3, 32, 32, device=dev), torch.rand(2, 3, 128, 128, device=dev)
with torch.autocast("cuda", dtype=torch.bfloat16):
    loss = torch.nn.functional.l1_loss(m(x), y)
loss.backward()
plan = m._fast_plan
G = plan.fp.G
lo, hi = G.data_ptr(), G.data_ptr() + G.numel() * 4
fast = set(id(p) for s in plan.stages for p in s.params)
n_alias = sum(1 for p in m.parameters() if id(p) in fast and lo <= p.grad.data_ptr() < hi)
print("fast params", len(fast), "grads aliasing G", n_alias)
