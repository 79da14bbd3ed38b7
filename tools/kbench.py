#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at the bench shapes (HIP events, back-to-back launches)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import studiosr_amd as S  # noqa: E402
import studiosr_amd._lib as L  # noqa: E402
from studiosr_amd import ops  # noqa: E402
from studiosr_amd.models import swinir as SW  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    torch.manual_seed(0)
    dev = torch.device("cuda")
    m = S.SwinIR(scale=4, depths=[2], num_heads=[6]).eval().to(dev).set_precision("bf16")
    cdt = torch.bfloat16
    P = m._get_packed(cdt)
    lp = P["layers"][0]
    geo, bp = lp["geo"], lp["blocks"][1]
    which = sys.argv[1:] or ["mlp", "msa", "blk"]
    for B in (1, 2, 4, 8, 16):
        H = W = 72
        t = torch.randn(B, H, W, geo.Cp, device=dev)
        t[..., geo.C:] = 0
        ws_ = S.runtime.Workspace(dev)
        M = B * H * W
        res = {}
        if "mlp" in which:
            res["mlp"] = timeit(lambda: SW.run_mlp(bp, bp["ln2"], geo, t, ws_, cdt))
        if "blk" in which:
            res["blk"] = timeit(lambda: SW.run_swin_block(bp, geo, t, t, ws_, cdt, bp["shift"]))
        if "msa" in which:
            res["msa"] = timeit(lambda: SW.run_window_msa(bp, bp["ln1"], geo, t, t, t, ws_, cdt, bp["shift"]))
        print(f"B={B:2d} M={M:6d} WG64={M // 64:5d} " + " ".join(f"{k}={v:8.1f}us" for k, v in res.items()), flush=True)




def conv_bench():
    """conv3x3 micro-benchmarks: python tools/kbench.py conv"""
    from studiosr_amd.models.common import conv_call
    from studiosr_amd import packing
    dev = torch.device("cuda")
    cdt = torch.bfloat16
    only = os.environ.get("KB_CONV")  # e.g. KB_CONV=16,64,64,256,256 runs that case alone (any shape; KB_POOL=1 adds the pool-partial output)
    extra = [tuple(int(v) for v in only.split(",")) + ("nhwc",)] if only else []
    for (B, H, W, cin, cout, mode) in extra + [(1, 144, 144, 64, 256, "nhwc"), (2, 144, 144, 64, 256, "nhwc"), (4, 144, 144, 64, 256, "nhwc"), (1, 72, 72, 192, 192, "nhwc"), (8, 144, 144, 64, 256, "ps"), (8, 144, 144, 64, 256, "nhwc"), (8, 72, 72, 64, 256, "ps"), (8, 72, 72, 192, 192, "nhwc"),
                                       (8, 72, 72, 192, 192, "nhwc_f32"), (8, 288, 288, 64, 16, "nhwc"),
                                       (8, 288, 288, 64, 16, "final"), (16, 256, 256, 256, 16, "final"), (16, 64, 64, 256, 256, "nhwc"), (16, 64, 64, 256, 1024, "ps"), (16, 128, 128, 256, 1024, "ps"), (16, 64, 64, 64, 64, "nhwc")]:
        if only and only != f"{B},{H},{W},{cin},{cout}":
            continue
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        if mode == "ps":
            rows = packing.pixel_shuffle_rows(cout // 4, cout // 4, 2)
        else:
            rows = packing.identity_idx(cout, cout)
        wp, bp = packing.pack_conv3x3(w, b, cin, rows, cdt)
        xdt = torch.float32 if mode == "nhwc_f32" else cdt
        x = torch.randn(B, H, W, cin, device=dev).to(xdt)
        if mode == "final":
            out = torch.empty(B, 3, H, W, device=dev)
            fs, fb = torch.ones(3, device=dev), torch.zeros(3, device=dev)
            fn = lambda: conv_call(x, wp, bp, out, cdt, out_mode=L.OUT_FINAL_NCHW, fin=(fs, fb, 3, H, W), cout_p=16)
        elif mode == "ps":
            out = torch.empty(B, 2 * H, 2 * W, cout // 4, device=dev, dtype=cdt)
            fn = lambda: conv_call(x, wp, bp, out, cdt, out_mode=L.OUT_PIXEL_SHUFFLE, ps_r=2, cps_p=cout // 4)
        else:
            out = torch.empty(B, H, W, cout, device=dev, dtype=torch.float32 if (mode == "nhwc_f32" or os.environ.get("KB_F32OUT")) else cdt)
            pool = None
            if os.environ.get("KB_POOL"):
                pool = torch.zeros(B, ops.conv_pool_tiles(H, W, out.shape[-1], L.SR_BF16), out.shape[-1], device=dev)
            fn = lambda: conv_call(x, wp, bp, out, cdt, pool=pool)
        us = timeit(fn)
        gf = 2.0 * B * H * W * 9 * cin * cout / 1e9
        print(f"conv B={B} {H}x{W} {cin}->{cout} {mode:9s}: {us:8.1f} us  {gf / us * 1e3:8.1f} TF/s  input {x.numel() * x.element_size() / us / 1e6:6.2f} TB/s", flush=True)
        if extra:
            break
        if os.environ.get("KB_STAMPS"):  # needs a `make STAMPS=1` build: s_memtime stamps of workgroup 7, wave 0 (sr_conv.hip only)
            import ctypes
            buf = (ctypes.c_ulonglong * 16)()
            f = L.lib().sr_debug_conv_stamps
            f.argtypes = [ctypes.c_void_p]
            f(buf)
            v = [buf[i] for i in range(5)]
            print("   stamps: stage", v[1] - v[0], "barrier", v[2] - v[1], "mfma", v[3] - v[2], "epilogue", v[4] - v[3], "total", v[4] - v[0], flush=True)


def elem_bench():
    """HBM-bound helpers in GB/s (bytes = read + written): python tools/kbench.py elem"""
    from studiosr_amd import ops
    dev = torch.device("cuda")
    rows = []
    for (B, C, H, W, r, dt) in [(8, 64, 144, 144, 2, torch.bfloat16), (16, 256, 128, 128, 2, torch.bfloat16), (8, 64, 72, 72, 2, torch.bfloat16),
                                (8, 20, 96, 96, 3, torch.float32), (8, 4, 64, 64, 4, torch.float32)]:
        x = torch.randn(B, C * r * r, H, W, device=dev).to(dt)
        us = timeit(lambda: ops.pixel_shuffle(x, r))
        ref = torch.nn.functional.pixel_shuffle(x, r)
        ok = torch.equal(ops.pixel_shuffle(x, r), ref)
        nbytes = 2 * x.numel() * x.element_size()
        rows.append((f"pixel_shuffle [{B},{C * r * r},{H},{W}] r={r} {str(dt)[6:]}", us, nbytes, ok))
    t = torch.randn(8, 72, 72, 192, device=dev)
    o = torch.empty_like(t)
    g, b = torch.ones(192, device=dev), torch.zeros(192, device=dev)
    rows.append(("layernorm [8,72,72,192] fp32", timeit(lambda: ops.layernorm(t, o, g, b, 180)), 2 * t.numel() * 4, True))
    x = torch.rand(8, 3, 64, 64, device=dev)
    xin = torch.empty(8, 72, 72, 32, device=dev, dtype=torch.bfloat16)
    sc, bi = torch.ones(3, device=dev), torch.zeros(3, device=dev)
    rows.append(("ingest [8,3,64,64] -> [8,72,72,32] bf16", timeit(lambda: ops.ingest_nchw(x, xin, L.PAD_EVAL_MIRROR, sc, bi)), x.numel() * 4 + xin.numel() * 2, True))
    y = torch.rand(8, 3, 256, 256, device=dev)
    rows.append(("nchw_to_u8 [8,3,256,256]", timeit(lambda: ops.nchw_to_u8(y, 255.0)), y.numel() * 5, True))
    for name, us, nbytes, ok in rows:
        print(f"{name:54s}: {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s  {nbytes / 1e6:8.1f} MB  exact={ok}", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "elem":
    elem_bench()
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "conv":
    conv_bench()
    sys.exit(0)

if __name__ == "__main__":
    main()
