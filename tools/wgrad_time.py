"""Launch time of one HAB's weight-gradient launch (sr_tr_wgrad: 4 linear jobs + the CAB's two 3 x 3 jobs) at the training shape 4 x 64 x 64, and of its parts:
python tools/wgrad_time.py  (SR_WG_KS=n token slices)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from studiosr_amd import fasttrain as F  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = 4, 64, 64
T, CP, HP = B * H * W, 192, 384
bf = torch.bfloat16
mk = lambda *s: torch.randn(*s, device=dev).to(bf)  # noqa: E731
dqkvw, n1w, dx1sw, o, dhw, n2w, doutw, gw = mk(T, 3 * CP), mk(T, CP), mk(T, CP), mk(T, CP), mk(T, HP), mk(T, CP), mk(T, CP), mk(T, HP)
dmid, n1, dyc, mid_g = mk(T, 64), mk(T, CP), mk(T, CP), mk(T, 64)
ks, ksc = F.WG_KS, F.conv_ks(B, H, W)
out = torch.empty(max(ks, ksc) * 9 * 3 * CP * HP + 16, device=dev)
lin = [
    dict(A=dqkvw.data_ptr(), B=n1w.data_ptr(), out=out.data_ptr(), lda=3 * CP, ldb=CP, Np=3 * CP, Kp=CP, T=T, taps=1, H=H, W=W, ones_col=-1, ks=ks),
    dict(A=dx1sw.data_ptr(), B=o.data_ptr(), out=out.data_ptr(), lda=CP, ldb=CP, Np=CP, Kp=CP, T=T, taps=1, H=H, W=W, ones_col=30, ks=ks),
    dict(A=dhw.data_ptr(), B=n2w.data_ptr(), out=out.data_ptr(), lda=HP, ldb=CP, Np=HP, Kp=CP, T=T, taps=1, H=H, W=W, ones_col=-1, ks=ks),
    dict(A=doutw.data_ptr(), B=gw.data_ptr(), out=out.data_ptr(), lda=CP, ldb=HP, Np=CP, Kp=HP, T=T, taps=1, H=H, W=W, ones_col=-1, ks=ks),
]
conv = [
    dict(A=dmid.data_ptr(), B=n1.data_ptr(), out=out.data_ptr(), lda=64, ldb=CP, Np=64, Kp=CP, T=T, taps=9, H=H, W=W, ones_col=-1, ks=ksc),
    dict(A=dyc.data_ptr(), B=mid_g.data_ptr(), out=out.data_ptr(), lda=CP, ldb=64, Np=CP, Kp=64, T=T, taps=9, H=H, W=W, ones_col=60, ks=ksc),
]


def timed(jobs, n=30):
    for _ in range(3):
        F._wgrad(jobs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        F._wgrad(jobs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"ks {ks} conv ks {ksc}: all six {timed(lin + conv):.1f} us, linear {timed(lin):.1f} us, conv {timed(conv):.1f} us, qkv {timed(lin[:1]):.1f} us, fc1 {timed(lin[2:3]):.1f} us")
