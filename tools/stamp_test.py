import sys, ctypes, torch
sys.path.insert(0, '/root/repo')
import studiosr_amd as S, studiosr_amd._lib as L
from studiosr_amd.models.common import conv_call
from studiosr_amd import packing
dev = torch.device('cuda'); cdt = torch.bfloat16
lib = L.lib()
for (B,H,W,cin,cout) in [(1,144,144,64,256),(1,72,72,192,192),(8,144,144,64,256)]:
    w = torch.randn(cout,cin,3,3,device=dev)*0.05; b = torch.randn(cout,device=dev)
    wp,bp = packing.pack_conv3x3(w,b,cin,packing.identity_idx(cout,cout),cdt)
    x = torch.randn(B,H,W,cin,device=dev).to(cdt); out = torch.empty(B,H,W,cout,device=dev,dtype=cdt)
    for _ in range(3): conv_call(x,wp,bp,out,cdt)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong*16)()
    lib.sr_debug_conv_stamps.argtypes=[ctypes.c_void_p]; lib.sr_debug_conv_stamps(buf)
    t=[buf[i] for i in range(5)]
    print((B,H,W,cin,cout), 'stage', t[1]-t[0], 'barrier', t[2]-t[1], 'main', t[3]-t[2], 'epi', t[4]-t[3], 'cycles')
