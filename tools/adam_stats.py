"""Debug: torch.optim.Adam(fused) vs studiosr_amd.optim.Adam on the fused HAT training path, step by step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
import test_training as T
from studiosr_amd.optim import Adam
DEV = "cuda:0"
torch.manual_seed(11)
xa, ya = torch.rand(2, 3, 32, 32, device=DEV), torch.rand(2, 3, 64, 64, device=DEV)
kw = dict(lr=1e-3, betas=(0.9, 0.99))
ms = [T._default_width_hat(), T._default_width_hat()]
opts = [torch.optim.Adam(ms[0].parameters(), fused=True, **kw), Adam(ms[1].parameters(), model=ms[1], **kw)]
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
def fresh_grad(m):
    m2 = T._default_width_hat()
    m2.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    with torch.autocast("cuda", dtype=torch.bfloat16):
        F.l1_loss(m2(xa).float(), ya).backward()
    return torch.cat([p.grad.flatten() for p in m2.parameters()]).clone()
for it in range(3):
    gs = []
    fr = [fresh_grad(m) for m in ms]
    for m, opt in zip(ms, opts):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.l1_loss(m(xa).float(), ya)
        loss.backward()
        gs.append(torch.cat([p.grad.flatten() for p in m.parameters()]).clone())
        opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    P = [torch.cat([p.detach().flatten() for p in m.parameters()]) for m in ms]
    M = [torch.cat([o.state[p]["exp_avg"].flatten() for p in m.parameters()]) for m, o in zip(ms, opts)]
    V = [torch.cat([o.state[p]["exp_avg_sq"].flatten() for p in m.parameters()]) for m, o in zip(ms, opts)]
    print(f"   vs a fresh model on the same parameters: torch-run grad rel {rel(gs[0], fr[0]):.3e}  ours-run grad rel {rel(gs[1], fr[1]):.3e}")
    print(f"step {it + 1}: grad rel {rel(gs[1], gs[0]):.3e}  P rel {rel(P[1], P[0]):.3e} max {float((P[1] - P[0]).abs().max()):.3e}  m rel {rel(M[1], M[0]):.3e}  v rel {rel(V[1], V[0]):.3e}  steps {float(opts[0].state[next(ms[0].parameters())]['step'])} {float(opts[1].state[next(ms[1].parameters())]['step'])}")
