import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S
from studiosr_amd.optim import Adam
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = S.HAT(scale=4).to(dev).train()
opt = Adam(m.parameters(), model=m, lr=2e-4, betas=(0.9, 0.99))
x, y = torch.rand(4, 3, 64, 64, device=dev), torch.rand(4, 3, 256, 256, device=dev)
def step():
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.l1_loss(m(x), y)
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("host enqueue ms", sorted(a for a, _ in ts)[5] * 1e3, "total ms", sorted(b for _, b in ts)[5] * 1e3)
