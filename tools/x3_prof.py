import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import studiosr_amd as S
dev = torch.device("cuda")
m = S.SwinIR(scale=4).eval().to(dev).set_precision("fp32x3")
x = torch.rand(8, 3, 64, 64, device=dev)
with torch.no_grad():
    for _ in range(4):
        m(x)
torch.cuda.synchronize()
