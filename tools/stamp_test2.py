import sys, ctypes, torch
sys.path.insert(0, '/root/repo')
import studiosr_amd as S, studiosr_amd._lib as L
from studiosr_amd.models import swinir as SW
dev = torch.device('cuda'); cdt = torch.bfloat16
lib = L.lib()
m = S.SwinIR(scale=4, depths=[2], num_heads=[6]).eval().to(dev).set_precision("bf16")
P = m._get_packed(cdt); lp = P["layers"][0]; geo, bp = lp["geo"], lp["blocks"][1]
for B in (1, 8):
    t = torch.randn(B, 72, 72, geo.Cp, device=dev); t[..., geo.C:] = 0
    ws_ = S.runtime.Workspace(dev)
    for name, fn, sym in (("mlp", lambda: SW.run_mlp(bp, bp["ln2"], geo, t, ws_, cdt), "sr_debug_mlp_stamps"),
                          ("swa", lambda: SW.run_window_msa(bp, bp["ln1"], geo, t, t, t, ws_, cdt, bp["shift"]), "sr_debug_swa_stamps"),
                          ("blk", lambda: SW.run_swin_block(bp, geo, t, t, ws_, cdt, bp["shift"]), "sr_debug_swa_stamps")):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong*16)()
        f = getattr(lib, sym); f.argtypes=[ctypes.c_void_p]; f(buf)
        v=[buf[i] for i in range(16)]
        order=[0,1,2,3,4,5,6,7,8,10,11,12,13,14,15,9] if name=='blk' else list(range(10))
        w=[v[i] for i in order]
        print(name, 'B', B, 'deltas', [w[i+1]-w[i] for i in range(len(w)-1)], 'total', max(v)-v[0])
