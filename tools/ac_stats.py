import sys, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch.nn.functional as F
import studiosr_amd as S
from test_training import load_golden, golden_cfg, golden_sd
for tag, kind in [("hat","HAT"),("edsr","EDSR"),("swinir","SwinIR"),("rcan","RCAN")]:
    g = load_golden(f"f15_grads_{tag}")
    m = getattr(S, kind)(**golden_cfg(g)); m.load_state_dict(golden_sd(g)); m = m.cuda().train()
    x, tgt = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["target"]).cuda()
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        loss = F.l1_loss(m(x), tgt)
    loss.backward()
    ref = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("grad/")}
    num = den = 0.0; worst = (0, ""); worstmax = (0, "")
    for n, p in m.named_parameters():
        if p.grad is None: continue
        d = (p.grad.cpu() - ref[n]).double(); r = ref[n].double()
        num += float((d*d).sum()); den += float((r*r).sum())
        rel = float(d.norm() / max(r.norm(), 1e-30)); relmax = float(d.abs().max() / max(r.abs().max(), 1e-30))
        if rel > worst[0]: worst = (rel, n)
        if relmax > worstmax[0]: worstmax = (relmax, n)
    print(tag, "loss", loss.item(), float(g["loss"]), "global relL2", (num/den)**0.5, "worst relL2", worst, "worst relmax", worstmax)
