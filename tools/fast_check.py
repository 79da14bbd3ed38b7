"""GPU checks of the fast training path (studiosr_amd/fasttrain.py, C ABI v7) piece by piece against torch fp32 math, then a reduced-depth
default-width HAT step against the generic engine.  `python tools/fast_check.py [wgrad] [attn] [block] [model]`"""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import studiosr_amd as S  # noqa: E402
from studiosr_amd import _lib as L  # noqa: E402
from studiosr_amd import fasttrain as F  # noqa: E402

dev = "cuda:0"
bf = torch.bfloat16


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def check_wgrad():
    torch.manual_seed(0)
    T, Np, Kp = 4096, 576, 192
    A = (torch.randn(T, Np, device=dev) * 0.5).to(bf)
    B = (torch.randn(T, Kp, device=dev) * 0.5).to(bf)
    ks = 8
    out = torch.zeros(ks, 1, Np, Kp, device=dev)
    F._wgrad([dict(A=A.data_ptr(), B=B.data_ptr(), out=out.data_ptr(), lda=Np, ldb=Kp, Np=Np, Kp=Kp, T=T, taps=1, H=1, W=1, ones_col=180, ks=ks)])
    Bo = B.float().clone()
    Bo[:, 180] = 1.0
    ref = A.float().t() @ Bo
    print("wgrad linear rel err", rel(out.sum(0)[0], ref))
    # conv taps
    Bn, H, W, Co, Ci = 2, 16, 32, 64, 192
    T = Bn * H * W
    dy = (torch.randn(Bn, H, W, Co, device=dev) * 0.5).to(bf)
    x = (torch.randn(Bn, H, W, Ci, device=dev) * 0.5).to(bf)
    out = torch.zeros(ks, 9, Co, Ci, device=dev)
    out2 = torch.zeros(ks, 9, Ci, Co, device=dev)
    F._wgrad([dict(A=dy.data_ptr(), B=x.data_ptr(), out=out.data_ptr(), lda=Co, ldb=Ci, Np=Co, Kp=Ci, T=T, taps=9, H=H, W=W, ones_col=-1, ks=ks),
              dict(A=x.data_ptr(), B=dy.data_ptr(), out=out2.data_ptr(), lda=Ci, ldb=Co, Np=Ci, Kp=Co, T=T, taps=9, H=H, W=W, ones_col=60, ks=ks)])
    xw = x.float().permute(0, 3, 1, 2).requires_grad_(False)
    w = torch.zeros(Co, Ci, 3, 3, device=dev, requires_grad=True)
    y = torch.nn.functional.conv2d(xw, w, padding=1)
    y.backward(dy.float().permute(0, 3, 1, 2))
    ref = w.grad.permute(2, 3, 0, 1).reshape(9, Co, Ci)
    print("wgrad conv rel err", rel(out.sum(0), ref))
    dyo = dy.float().clone()
    dyo[..., 60] = 1.0
    # job 2: dW2[tap][n = ci][k = co] = sum_p x[p][ci] * dy[p + off(tap)][co]  (dy with column 60 := 1 where the source pixel exists)
    pad = torch.nn.functional.pad(dyo, (0, 0, 1, 1, 1, 1))
    ref2 = torch.stack([torch.einsum("bhwi,bhwo->io", x.float(), pad[:, t // 3:t // 3 + H, t % 3:t % 3 + W, :]) for t in range(9)])
    print("wgrad conv job2 rel err", rel(out2.sum(0), ref2))


def attn_ref(q, k, v, bias, mask):
    # q,k,v [nb, h, N, 32] fp32, bias [h, Nq, Nk], mask [nb, Nq, Nk] or None
    s = q @ k.transpose(-1, -2) + bias[None]
    if mask is not None:
        s = s + mask[:, None]
    p = torch.softmax(s, -1)
    return p @ v


def check_attn(shift=8):
    from studiosr_amd import autograd as A

    torch.manual_seed(1)
    Bn, H, W = 2, 32, 32
    nb, h, N = Bn * (H // 16) * (W // 16), 6, 256
    q = (torch.randn(nb, h, N, 32, device=dev) * 0.3)
    k = (torch.randn(nb, h, N, 32, device=dev) * 0.3)
    v = torch.randn(nb, h, N, 32, device=dev)
    for t in (q, k, v):
        t[..., 30:] = 0
    from studiosr_amd.models.hat import rpi_sa

    rpi = rpi_sa(16).to(dev)
    table = (torch.randn(961, h, device=dev) * 0.2).requires_grad_(True)
    bias = table[rpi.reshape(-1)].reshape(N, N, h).permute(2, 0, 1).contiguous().detach()
    dO = torch.randn(nb, N, h * 32, device=dev)
    qb, kb, vb, dOb = (t.to(bf) for t in (q, k, v, dO))
    mask = None
    if shift:
        m = A.shift_mask(H, W, 16, shift, dev)  # [nW, N, N]
        mask = m.repeat(Bn, 1, 1)
    q_, k_, v_ = (t.float().clone().requires_grad_(True) for t in (qb, kb, vb))
    o = attn_ref(q_, k_, v_, bias, mask)  # [nb, h, N, 32]
    o_rows = o.permute(0, 2, 1, 3).reshape(nb, N, h * 32)
    bias_ = table[rpi.reshape(-1)].reshape(N, N, h).permute(2, 0, 1)
    o2 = attn_ref(q_, k_, v_, bias_, mask).permute(0, 2, 1, 3).reshape(nb, N, h * 32)
    o2.backward(dOb.float())
    ob = o_rows.detach().to(bf).contiguous()
    groups = 4
    dq, dk, dv = (torch.zeros(nb, h, N, 32, device=dev, dtype=bf) for _ in range(3))
    lse, delta = torch.zeros(nb, h, N, device=dev), torch.zeros(nb, h, N, device=dev)
    tpad = 1024
    dtp = torch.zeros(h * groups * 4, tpad, device=dev)
    rpi32 = rpi.to(torch.int32).contiguous()
    qT, kT = qb.transpose(-1, -2).contiguous(), kb.transpose(-1, -2).contiguous()
    dOT = dOb.reshape(nb, N, h, 32).permute(0, 2, 3, 1).contiguous()
    biasT = bias.transpose(1, 2).contiguous()
    F._call(L.lib().sr_tr_attn_bwd, L.SrTrAttnBwd, "attn_bwd", q=qb.data_ptr(), qT=qT.data_ptr(), k=kb.data_ptr(), kT=kT.data_ptr(), v=vb.data_ptr(), o=ob.data_ptr(),
            dO=dOb.data_ptr(), dOT=dOT.data_ptr(), bias=bias.data_ptr(), biasT=biasT.data_ptr(), dq=dq.data_ptr(), dk=dk.data_ptr(), dv=dv.data_ptr(), lse=lse.data_ptr(),
            delta=delta.data_ptr(), dtab_part=dtp.data_ptr(), rpi=rpi32.data_ptr(), n_bwin=nb, heads=h, hd_p=32, Nq=N, Nk=N, ldo=h * 32, groups=groups, T=961, Tpad=tpad, toeplitz16=1,
            H=H, W=W, ws=16, shift=shift)
    torch.cuda.synchronize()
    print(f"attn bwd shift={shift}: dq {rel(dq, q_.grad):.3e} dk {rel(dk, k_.grad):.3e} dv {rel(dv, v_.grad):.3e} dtable {rel(dtp.reshape(h, groups * 4, tpad).sum(1)[:, :961].t(), table.grad):.3e}")


def time_attn():
    """launch times of the two attention-backward passes at the training shape (4 x 64 x 64: 64 windows)"""
    from studiosr_amd.models.hat import rpi_sa

    torch.manual_seed(1)
    nb, h, N = 64, 6, 256
    mk = lambda *s: (torch.randn(*s, device=dev) * 0.3).to(bf)  # noqa: E731
    q, k, v, qT, kT, dOT = mk(nb, h, N, 32), mk(nb, h, N, 32), mk(nb, h, N, 32), mk(nb, h, 32, N), mk(nb, h, 32, N), mk(nb, h, 32, N)
    o, dO = mk(nb, N, 192), mk(nb, N, 192)
    bias, biasT = torch.randn(h, N, N, device=dev), torch.randn(h, N, N, device=dev)
    dq, dk, dv = (torch.zeros(nb, h, N, 32, device=dev, dtype=bf) for _ in range(3))
    lse, delta = torch.zeros(nb, h, N, device=dev), torch.zeros(nb, h, N, device=dev)
    groups = 16
    dtp = torch.zeros(h * groups * 4, 1024, device=dev)
    rpi32 = rpi_sa(16).to(dev).to(torch.int32).contiguous()
    def run():
        F._call(L.lib().sr_tr_attn_bwd, L.SrTrAttnBwd, "attn_bwd", q=q.data_ptr(), qT=qT.data_ptr(), k=k.data_ptr(), kT=kT.data_ptr(), v=v.data_ptr(), o=o.data_ptr(),
                dO=dO.data_ptr(), dOT=dOT.data_ptr(), bias=bias.data_ptr(), biasT=biasT.data_ptr(), dq=dq.data_ptr(), dk=dk.data_ptr(), dv=dv.data_ptr(), lse=lse.data_ptr(),
                delta=delta.data_ptr(), dtab_part=dtp.data_ptr(), rpi=rpi32.data_ptr(), n_bwin=nb, heads=h, hd_p=32, Nq=N, Nk=N, ldo=192, groups=groups, T=961, Tpad=1024, toeplitz16=1,
                H=64, W=64, ws=16, shift=8)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"attn bwd (both passes) {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  QVAR={os.environ.get('SR_TR_QVAR', '0')}")


def make_hat(depth=2):
    torch.manual_seed(0)
    m = S.HAT(scale=2, depths=[depth], num_heads=[6], drop_path_rate=0.0)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.05)
            if "relative_position_bias_table" in n:
                p.add_(torch.randn_like(p) * 0.2)
    return m.to(dev).train()


def step(m, x, y, autocast, fast):
    os.environ["SR_FAST_TRAIN"] = "1" if fast else "0"
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=bf, enabled=autocast):
        out = m(x)
        loss = torch.nn.functional.l1_loss(out.float(), y)
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, out.detach().float()


def check_model(depth=2, B=2, size=32):
    m = make_hat(depth)
    torch.manual_seed(3)
    x = torch.rand(B, 3, size, size, device=dev)
    y = torch.rand(B, 3, size * 2, size * 2, device=dev)
    l32, g32, o32 = step(m, x, y, False, False)
    lac, gac, oac = step(m, x, y, True, False)
    lf, gf, of = step(m, x, y, True, True)
    print(f"loss fp32 {l32:.6f} generic-autocast {lac:.6f} fast {lf:.6f}; out err generic {rel(oac, o32):.3e} fast {rel(of, o32):.3e}")
    tot = lambda g: torch.cat([g[n].flatten() for n in sorted(g32)])  # noqa: E731
    print(f"whole gradient rel L2 vs fp32: generic-autocast {rel(tot(gac), tot(g32)):.3e}  fast {rel(tot(gf), tot(g32)):.3e}")
    worst = []
    for n in sorted(g32):
        if n not in gf:
            print("MISSING grad", n)
            continue
        worst.append((rel(gf[n], g32[n]), rel(gac[n], g32[n]), n, g32[n].abs().max().item()))
    worst.sort(reverse=True)
    for e in worst[:25]:
        print(f"  fast {e[0]:.3e}  generic {e[1]:.3e}  max|g| {e[3]:.2e}  {e[2]}")


if __name__ == "__main__":
    what = sys.argv[1:] or ["wgrad", "attn", "model"]
    for w in what:
        try:
            if w == "wgrad":
                check_wgrad()
            elif w == "attn":
                check_attn(8)
                check_attn(0)
            elif w == "tattn":
                time_attn()
            elif w == "model":
                check_model()
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()
