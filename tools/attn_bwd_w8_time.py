"""Time sr_tr_attn_bwd on 8 x 8 windows alone (the SwinIR training shape: 4 x 64 x 64 tokens = 256 windows, 6 heads): python tools/attn_bwd_w8_time.py [groups]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from studiosr_amd import _lib as L, fasttrain as FT
import studiosr_amd as S

dev = "cuda:0"
bf = torch.bfloat16
Bn, H, W, ws, h, N, T = 4, 64, 64, 8, 6, 64, 225
nb = Bn * (H // ws) * (W // ws)
groups = int(sys.argv[1]) if len(sys.argv) > 1 else nb // 4
shift = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.manual_seed(0)
q, k, v = (torch.randn(nb, h, N, 32, device=dev).to(bf) for _ in range(3))
dO, o = (torch.randn(nb, N, h * 32, device=dev).to(bf) for _ in range(2))
rpi = S.SwinIR(depths=[2], num_heads=[6]).layers[0].residual_group.blocks[0].attn.relative_position_index.to(dev).to(torch.int32).contiguous()
bias = torch.randn(h, N, N, device=dev)
biasT = bias.transpose(1, 2).contiguous()
dq, dk, dv = (torch.empty(nb, h, N, 32, device=dev, dtype=bf) for _ in range(3))
lse, delta = torch.zeros(nb, h, N, device=dev), torch.zeros(nb, h, N, device=dev)
dtp = torch.zeros(h * groups, 256, device=dev)
qT, kT = q.transpose(-1, -2).contiguous(), k.transpose(-1, -2).contiguous()
dOT = dO.reshape(nb, N, h, 32).permute(0, 2, 3, 1).contiguous()


def run():
    FT._call(L.lib().sr_tr_attn_bwd, L.SrTrAttnBwd, "sr_tr_attn_bwd", q=q.data_ptr(), qT=qT.data_ptr(), k=k.data_ptr(), kT=kT.data_ptr(), v=v.data_ptr(), o=o.data_ptr(),
             dO=dO.data_ptr(), dOT=dOT.data_ptr(), bias=bias.data_ptr(), biasT=biasT.data_ptr(), dq=dq.data_ptr(), dk=dk.data_ptr(), dv=dv.data_ptr(), lse=lse.data_ptr(),
             delta=delta.data_ptr(), dtab_part=dtp.data_ptr(), rpi=rpi.data_ptr(), n_bwin=nb, heads=h, hd_p=32, Nq=N, Nk=N, ldo=h * 32, groups=groups, T=T, Tpad=256,
             toeplitz16=int(groups * 4 == nb), H=H, W=W, ws=ws, shift=shift)


for _ in range(5):
    run()
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
print(f"groups={groups} shift={shift}: {best:.1f} us per call")
