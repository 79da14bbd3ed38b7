import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import studiosr_amd as S
from torch.profiler import profile, ProfilerActivity
dev = "cuda:0"
torch.manual_seed(0)
m = S.HAT(scale=4).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.9, 0.99), fused=True)
x, y = torch.rand(4, 3, 64, 64, device=dev), torch.rand(4, 3, 256, 256, device=dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.l1_loss(m(x), y)
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60, max_shapes_column_width=70))
