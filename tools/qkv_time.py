import sys, os, torch
sys.path.insert(0, "/root/repo")
import studiosr_amd as S
from studiosr_amd import _lib as L, ops
from studiosr_amd.runtime import x3_mode
dev = torch.device("cuda:0")
m = S.HAT(scale=4, depths=[2], num_heads=[6]).to(dev).eval()
for prec, cdt in (("bf16", torch.bfloat16), ("fp32x3", torch.float32)):
    m.set_precision(prec)
    lp = m._get_packed(cdt)["layers"][0]
    geo = lp["geo"]; bp = lp["blocks"][0]
    B, H, W = 4, 64, 64
    t = torch.randn(B, H, W, geo.Cp, device=dev); t[..., geo.C:] = 0
    nb = B * H * W // 256
    q = torch.empty(nb, 6, 256, 32, device=dev, dtype=cdt); k = torch.empty_like(q); vt = torch.empty_like(q)
    def run():
        ops.swin_qkv(x=t.data_ptr(), q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), wstream=bp["qkv_stream"].data_ptr(), B=B, H=H, W=W, C=geo.C, Cp=geo.Cp, ldx=geo.Cp,
                     heads=6, hd_p=32, ws=16, shift=0, eps=1e-5, y_mode=L.Y_ROLL, compute_dtype=bp["qkv_dtype"], frag_order=0)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(prec, "sr_swin_qkv", round(e0.elapsed_time(e1) / 50 * 1e3, 1), "us")
