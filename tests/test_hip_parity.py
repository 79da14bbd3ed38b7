"""GPU parity tests (run with -m gpu on an MI355X): the HIP hot path, called through the C ABI, against
(1) golden vectors produced by the reference itself and (2) the CPU oracle on the same seeded inputs.

Tolerances: the exact-fp32 path ('fp32': v_mfma_f32_16x16x4_f32, reference op order) must agree to fp32 rounding
(<= 2e-5 of the output range: summation order differs from ATen); the bf16 path (bf16 operands, fp32 accumulate /
stream / LayerNorm / softmax) to 1.5 % of the output range.  Integer index maps are bit-exact."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_cfg, golden_sd, load_golden

pytestmark = pytest.mark.gpu

import studiosr_amd as S  # noqa: E402
from studiosr_amd import _lib as L  # noqa: E402
from studiosr_amd import ops, packing  # noqa: E402
from studiosr_amd.models.common import conv_call  # noqa: E402
from oracle import functional as OF  # noqa: E402
from oracle import metrics as OMT  # noqa: E402
from oracle import models as OM  # noqa: E402

DEV = "cuda:0"
FP32_TOL, BF16_TOL = 2e-5, 1.5e-2
BF16_PSNR_DB = 48.0  # fixture models, bf16 operands vs the fp32 reference output (range-normalised)


def build(kind, name):
    g = load_golden(name)
    cfg, sd = golden_cfg(g), golden_sd(g)
    m = getattr(S, kind)(**cfg)
    m.load_state_dict(sd)
    return g, cfg, sd, m.to(DEV).eval()


# ----------------------------------------------------------------------------- integer index maps
@pytest.mark.parametrize("r", [2, 3, 4])
def test_pixel_shuffle_kernel_bit_exact(r):
    g = load_golden("f01_pixel_shuffle")
    x = torch.from_numpy(g[f"in_r{r}"]).to(DEV)  # int32 payload through the 4-byte copy kernel
    out = ops.pixel_shuffle(x.view(torch.float32), r).view(torch.int32).cpu()
    assert torch.equal(out, torch.from_numpy(g[f"out_r{r}"]))
    xb = torch.arange(2 * 4 * r * r * 3 * 5, dtype=torch.int16).reshape(2, 4 * r * r, 3, 5)
    outb = ops.pixel_shuffle(xb.to(DEV).view(torch.bfloat16), r).view(torch.int16).cpu()
    assert torch.equal(outb, OF.pixel_shuffle(xb, r))
    # widths that are a multiple of 16 bytes take the vector kernel (16-B loads, in-register interleave): same index map
    for dt, w in ((torch.int16, 24), (torch.int32, 12)):
        n = 3 * 5 * r * r * 6 * w
        xv = (torch.arange(n, dtype=torch.int64) % 30011).to(dt).reshape(3, 5 * r * r, 6, w)
        fdt = torch.bfloat16 if dt == torch.int16 else torch.float32
        outv = ops.pixel_shuffle(xv.to(DEV).view(fdt), r).view(dt).cpu()
        want = xv.reshape(3, -1)[:, torch.from_numpy(OF.pixel_shuffle_index_map(5, 6, w, r)).reshape(-1)].reshape(3, 5, 6 * r, w * r)
        assert torch.equal(outv, want)


@pytest.mark.parametrize("r,cps", [(2, 64), (3, 32), (4, 4)])
def test_conv_fused_pixel_shuffle_index_map_bit_exact(r, cps):
    """A 0/1 'selector' conv (centre tap only) makes the conv output an exact integer copy of its input channels, so the
    fused PixelShuffle store can be checked bit for bit against the oracle's index map."""
    cin, H, W, B = 32, 9, 20, 2
    cout = r * r * cps
    w = torch.zeros(cout, cin, 3, 3)
    w[torch.arange(cout), torch.arange(cout) % cin, 1, 1] = 1.0
    x = torch.randint(0, 128, (B, cin, H, W)).float()
    ref = OF.pixel_shuffle(x[:, torch.arange(cout) % cin], r)  # [B, cps, H*r, W*r]
    for dt in (torch.bfloat16, torch.float32):
        cps_p = packing.round_up(cps, 4) if cps < 32 else packing.round_up(cps, 32)
        while (r * r * cps_p) % 16:
            cps_p += 4
        wp, bp = packing.pack_conv3x3(w.to(DEV), None, cin, packing.pixel_shuffle_rows(cps, cps_p, r), dt)
        xin = x.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
        out = torch.zeros(B, H * r, W * r, cps_p, device=DEV, dtype=dt)
        conv_call(xin, wp, bp, out, dt, out_mode=L.OUT_PIXEL_SHUFFLE, ps_r=r, cps_p=cps_p)
        got = out[..., :cps].permute(0, 3, 1, 2).float().cpu()
        assert torch.equal(got, ref), (r, cps, dt)


def test_window_gather_scatter_through_the_gemm_is_exact():
    """sr_gemm with an identity weight: window-order gather (roll + partition) then scatter (reverse + roll back)."""
    B, H, W, C, ws, shift = 2, 16, 24, 64, 8, 4
    x = torch.randint(-64, 64, (B, H, W, C)).float().to(DEV)
    wp, _ = packing.pack_linear(torch.eye(C, device=DEV), None, packing.identity_idx(C, C), packing.identity_idx(C, C), torch.float32)
    mid = torch.zeros(B * H * W, C, device=DEV)
    ops.gemm(A=x.data_ptr(), Wp=wp.data_ptr(), out=mid.data_ptr(), M=B * H * W, K=C, N=C, lda=C, ldo=C, a_dtype=L.SR_F32, out_dtype=L.SR_F32,
             compute_dtype=L.SR_F32, out_scale=1.0, a_map=L.MAP_WINDOW, o_map=L.MAP_IDENTITY, H=H, W=W, ws=ws, shift=shift, epi=L.EPI_STD)
    src = torch.from_numpy(OF.window_token_source(H, W, ws, shift)).reshape(-1)
    ref = x.reshape(B, H * W, C).cpu()[:, src].reshape(-1, C)
    assert torch.equal(mid.cpu(), ref)
    back = torch.zeros_like(x)
    ops.gemm(A=mid.data_ptr(), Wp=wp.data_ptr(), out=back.data_ptr(), M=B * H * W, K=C, N=C, lda=C, ldo=C, a_dtype=L.SR_F32, out_dtype=L.SR_F32,
             compute_dtype=L.SR_F32, out_scale=1.0, a_map=L.MAP_IDENTITY, o_map=L.MAP_WINDOW, H=H, W=W, ws=ws, shift=shift, epi=L.EPI_STD)
    assert torch.equal(back.cpu(), x.cpu())


@pytest.mark.parametrize("mode,shape", [(L.PAD_EVAL_MIRROR, (5, 6)), (L.PAD_EVAL_MIRROR, (8, 8)), (L.PAD_REFLECT, (13, 17)), (L.PAD_NONE, (8, 16))])
def test_ingest_padding_bit_exact(mode, shape):
    h, w = shape
    x = torch.arange(2 * 3 * h * w, dtype=torch.float32).reshape(2, 3, h, w)
    ref = {L.PAD_EVAL_MIRROR: OF.pad_eval, L.PAD_REFLECT: OF.pad_reflect}.get(mode, lambda t, _: t)(x, 8)
    Hp, Wp = ref.shape[2:]
    out = torch.zeros(2, Hp, Wp, 32, device=DEV)
    one, zero = torch.ones(3, device=DEV), torch.zeros(3, device=DEV)
    ops.ingest_nchw(x.to(DEV), out, mode, one, zero)
    assert torch.equal(out[..., :3].permute(0, 3, 1, 2).cpu(), ref)
    assert float(out[..., 3:].abs().max()) == 0.0


# ----------------------------------------------------------------------------- whole models vs reference vectors
WHOLE = [
    ("f11_swinir_x2", "SwinIR"), ("f11_swinir_x3", "SwinIR"), ("f11_swinir_x4", "SwinIR"), ("f11_swinir_direct_x4", "SwinIR"),
    ("f11_swinir_c180_x4", "SwinIR"), ("f11_edsr_x2", "EDSR"), ("f11_edsr_x3", "EDSR"), ("f11_edsr_x4", "EDSR"), ("f11_edsr_r255_x2", "EDSR"),
    ("f11_rcan_x4", "RCAN"), ("f11_rcan_x3", "RCAN"), ("f11_hat_w8_x4", "HAT"), ("f11_hat_w16_x2", "HAT"),
]


@pytest.mark.parametrize("name,kind", WHOLE)
def test_whole_models_against_reference_vectors(name, kind):
    g, cfg, sd, m = build(kind, name)
    n = 0
    for k in sorted(g):
        if not k.startswith("y_"):
            continue
        mode, b, h, w = k[2:].split("_")
        x = torch.from_numpy(g[f"x_{b}_{h}_{w}"]).to(DEV)
        ref = torch.from_numpy(g[k])
        rng = float(ref.abs().max())
        m.train(mode == "train")
        for prec, tol in (("fp32", FP32_TOL), ("bf16", BF16_TOL)):
            m.set_precision(prec)
            with torch.no_grad():
                y = m(x).cpu()
            assert y.shape == ref.shape
            err = float((y - ref).abs().max())
            assert err <= tol * rng, f"{name}:{k}:{prec}: max|d|={err:.3e} range={rng:.3e}"
            # the max-abs bound alone would let a wrong bias fold / GELU form through in bf16: bound the MEAN error too
            # (bf16 operand rounding gives ~2^-9 relative per contraction; a mis-folded term shows up at the 1e-2 level)
            psnr = 10 * np.log10(rng * rng / max(float(((y - ref) ** 2).mean()), 1e-30))
            assert psnr >= (BF16_PSNR_DB if prec == "bf16" else 100.0), f"{name}:{k}:{prec}: PSNR {psnr:.1f} dB"
        n += 1
    assert n > 0


def test_inference_uint8_and_self_ensemble_against_reference():
    """Model.inference / inference_with_self_ensemble (common.py:36-67): uint8 in, round-half-even, clip, uint8 out."""
    g, cfg, sd, m = build("EDSR", "f12_inference")
    m.set_precision("fp32")
    y = m.inference(g["img"])
    assert y.dtype == np.uint8 and y.shape == g["y"].shape
    d = np.abs(y.astype(int) - g["y"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3  # a value within 1e-6 of .5 may round the other way
    ye = m.inference_with_self_ensemble(g["img"])
    de = np.abs(ye.astype(int) - g["y_ens"].astype(int))
    assert de.max() <= 1 and (de > 0).mean() < 2e-3
    # batched front end: images of two sizes, results in input order and equal to the one-image calls
    rng = np.random.default_rng(0)
    other = rng.integers(0, 256, size=(7, 10, 3), dtype=np.uint8)
    ys = m.inference_batch([g["img"], other, g["img"][::-1].copy()])
    assert np.array_equal(ys[0], y) and np.array_equal(ys[1], m.inference(other)) and np.array_equal(ys[2], m.inference(g["img"][::-1].copy()))


def test_uint8_front_and_back_end_are_bit_exact():
    """sr_u8_to_nchw / sr_nchw_to_u8 against the reference's own ops (common.py:42-45) incl. ties, negatives, > 255."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(2, 5, 7, 3), dtype=np.uint8)
    for div in (255.0, 1.0):
        got = ops.u8_to_nchw(torch.from_numpy(img).to(DEV), div).cpu()
        want = torch.from_numpy(img.astype(np.float32) / div).permute(0, 3, 1, 2)
        assert torch.equal(got, want)
    x = torch.cat([torch.rand(2, 3, 5, 7) * 1.2 - 0.1, (torch.arange(210).float().view(2, 3, 5, 7) + 0.5) / 255.0])  # incl. exact .5 ties
    for mult in (255.0, 1.0):
        got = ops.nchw_to_u8(x.to(DEV).contiguous(), mult).cpu()
        want = (x.permute(0, 2, 3, 1) * mult).round().clip(0, 255).to(torch.uint8)
        assert torch.equal(got, want)


def test_autocast_selects_bf16_like_the_reference_trainer():
    g, cfg, sd, m = build("EDSR", "f11_edsr_x2")
    x = torch.from_numpy(g["x_1_8_8"]).to(DEV)
    m.set_precision("auto")
    with torch.no_grad():
        y32 = m(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y16 = m(x)
        m.set_precision("bf16")
        yb = m(x)
    assert y16.dtype == torch.float32 and torch.equal(y16, yb) and not torch.equal(y16, y32)


# ----------------------------------------------------------------------------- full BASELINE sizes: oracle + invariants
@pytest.fixture(scope="module")
def swinir_full():
    torch.manual_seed(0)
    m = S.SwinIR(scale=4).eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.ndim == 1:
                p.add_(torch.randn_like(p) * 0.05)
    return m


def test_swinir_x4_full_size_one_tile_against_oracle(swinir_full):
    """Default SwinIR x4 (C 180, 36 blocks) on one 64x64 LR tile -- the bench workload -- vs the CPU oracle; also the
    metric's PSNR delta (<= 1e-3 dB for fp32) against a fixed synthetic target."""
    m = swinir_full
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 3, 64, 64, generator=g)
    tgt = (torch.rand(256, 256, 3, generator=g) * 255).round().to(torch.uint8).numpy()
    with torch.no_grad():
        ref = OM.swinir_forward(sd, x, m.get_model_config())
    m = m.to(DEV)
    u8 = lambda t: (t[0].permute(1, 2, 0) * 255.0).round().clip(0, 255).to(torch.uint8).numpy()  # noqa: E731
    p_ref = OMT.compute_psnr(u8(ref), tgt, y_only=True, crop_border=4)
    # fp32x3: fp32 tensors / op order, contractions as split-operand bf16 (hi*hi + hi*lo + lo*hi): must meet the metric's fp32 bar
    for prec, tol, dtol in (("fp32", 5e-5, 1e-3), ("fp32x3", 3e-4, 1e-3), ("bf16", 2e-2, 1e-2)):
        m.set_precision(prec)
        with torch.no_grad():
            y = m(x.to(DEV)).cpu()
        assert float((y - ref).abs().max()) <= tol * float(ref.abs().max()), (prec, float((y - ref).abs().max()))
        assert abs(OMT.compute_psnr(u8(y), tgt, y_only=True, crop_border=4) - p_ref) <= dtol, prec
    m.cpu()


def test_swinir_x4_batch8_invariants(swinir_full):
    """BASELINE config (batch 8, 64x64): tiles are independent, so every tile of the batched forward must equal its
    single-tile forward bit for bit; a HIP-graph replay must equal the eager launch sequence bit for bit."""
    from studiosr_amd.runtime import GraphedForward

    m = swinir_full.to(DEV).set_precision("bf16")
    x = torch.rand(8, 3, 64, 64, generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.no_grad():
        y = m(x).clone()
        assert y.shape == (8, 3, 256, 256) and bool(torch.isfinite(y).all())
        for i in (0, 3, 7):
            assert torch.equal(m(x[i : i + 1].contiguous())[0], y[i]), i
        gf = GraphedForward(lambda t: m(t), x)
        assert torch.equal(gf(x), y)
        xs = x.flip(0).contiguous()
        assert torch.equal(gf(xs), y.flip(0))
    m.cpu()


def test_graphed_forward_keeps_its_buffers_alive():
    """A HIP graph records raw device pointers; GraphedForward must own the callable (and through it the model and its
    workspace).  Regression test: dropping every outside reference and emptying the cache once faulted at replay."""
    import gc

    from studiosr_amd.runtime import GraphedForward

    g, cfg, sd, m = build("EDSR", "f11_edsr_x2")
    m.set_precision("bf16")
    x = torch.from_numpy(g["x_2_12_12"]).to(DEV)
    with torch.no_grad():
        ref = m(x).clone()
        gf = GraphedForward(lambda t, mm=m: mm(t), x)
    del m
    gc.collect()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 20,), 7.0, device=DEV) for _ in range(8)]  # would land in any memory the graph wrongly gave up
    assert torch.equal(gf(x), ref)
    del junk


@pytest.mark.parametrize("kind,cfg", [
    ("HAT", dict(scale=2, embed_dim=60, depths=[2], num_heads=[6], window_size=16)),            # side stream for the conv branch of a HAB
    ("SwinFIR", dict(scale=2, embed_dim=60, depths=[2], num_heads=[6], window_size=8)),         # torch glue + generic-engine SFB between fused blocks
    ("HAN", dict(scale=2, n_feats=64, n_resblocks=3, n_resgroups=2, reduction=16)),             # chained RCABs + LAM / CSAM on the generic engine
    ("RCAN", dict(scale=2, n_feats=64, n_resblocks=3, n_resgroups=2, reduction=16)),
])
def test_forwards_with_side_streams_or_torch_glue_replay_from_a_hip_graph(kind, cfg):
    """Every model's inference forward must be capturable: the second stream of HAT joins the capture, SwinFIR's / HAN's torch ops and
    generic-engine launches allocate from the graph's pool; three replays on fresh inputs equal the eager forward bit for bit."""
    from studiosr_amd.runtime import GraphedForward

    torch.manual_seed(5)
    m = _randomised(getattr(S, kind)(**cfg), seed=5).to(DEV).eval().set_precision("bf16")
    xs = [torch.rand(2, 3, 32, 32, device=DEV) for _ in range(3)]
    with torch.no_grad():
        want = [m(x).clone() for x in xs]
        gf = GraphedForward(lambda t: m(t), xs[0])
        for x, w in zip(xs, want):
            assert torch.equal(gf(x), w)


def test_rcan_batch16_graph_replay_on_four_streams_equals_the_eager_forward():
    """RCAN at B >= 16 captures as four quarter batches on four streams (models/rcan.py: the launches of one quarter run under the MFMAs of the
    others); eager forwards stay one launch sequence.  Both must give the same bits, image by image."""
    from studiosr_amd.runtime import GraphedForward

    torch.manual_seed(6)
    m = _randomised(S.RCAN(scale=2, n_feats=64, n_resblocks=3, n_resgroups=2, reduction=16), seed=6).to(DEV).eval().set_precision("bf16")
    xs = [torch.rand(16, 3, 24, 20, device=DEV) for _ in range(2)]
    with torch.no_grad():
        want = [m(x).clone() for x in xs]
        assert torch.equal(m(xs[0][3:4].contiguous())[0], want[0][3])  # batch independence (the gate is per image)
        gf = GraphedForward(lambda t: m(t), xs[0])
        for x, w in zip(xs, want):
            assert torch.equal(gf(x), w)


def test_edsr_x4_batch16_invariants():
    """BASELINE config 2 (EDSR x4, batch 16, 64x64): batch independence + eval-pad-free shape; fp32 vs bf16 PSNR."""
    torch.manual_seed(0)
    m = S.EDSR(scale=4).to(DEV).eval().set_precision("bf16")
    x = torch.rand(16, 3, 64, 64, generator=torch.Generator().manual_seed(2)).to(DEV)
    with torch.no_grad():
        y = m(x).clone()
        assert y.shape == (16, 3, 256, 256)
        assert torch.equal(m(x[5:6].contiguous())[0], y[5])
        m.set_precision("fp32")
        y32 = m(x[:2].contiguous())
    mse = float(((y[:2] - y32) ** 2).mean())
    assert 10 * np.log10(1.0 / mse) > 55.0



# ----------------------------------------------------------------------------- BASELINE configs 1, 2, 5 at full depth
def _u8(t):
    return (t[0].permute(1, 2, 0) * 255.0).round().clip(0, 255).to(torch.uint8).numpy()


def _sd_cpu(m):
    return {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in m.state_dict().items()}


def test_config1_edsr_baseline_x2_through_an_evaluator_loop():
    """BASELINE configs[0]: EDSR-baseline x2 = EDSR(scale=2, n_feats=64, n_resblocks=16, res_scale=1.0), a seeded
    48x48 uint8 LR image through the Evaluator's loop (evaluator.py:53-79: func(lq) -> compute_psnr(sr, gt, y_only, crop=scale))
    with func = model.inference (common.py:36-48), against the oracle's inference on the same weights."""
    from studiosr_amd.evaluator import Evaluator

    torch.manual_seed(0)
    m = _randomised(S.EDSR(scale=2, n_feats=64, n_resblocks=16, res_scale=1.0), seed=21)
    sd = _sd_cpu(m)
    cfg = m.get_model_config()
    rng = np.random.default_rng(0)
    pairs = [(rng.integers(0, 256, size=(48, 48, 3), dtype=np.uint8), rng.integers(0, 256, size=(96, 96, 3), dtype=np.uint8)) for _ in range(2)]
    m = m.to(DEV).eval().set_precision("auto")  # what Evaluator(model.inference) runs: the reference-precision path

    class Pairs:  # PairedImageDataset surface without files
        def __len__(self):
            return len(pairs)

        def __getitem__(self, i):
            if i >= len(pairs):
                raise IndexError(i)
            return pairs[i]

    ev = Evaluator.__new__(Evaluator)
    ev.scale, ev.dataset, ev.testset = 2, "synthetic", Pairs()
    got = []
    psnr_hip, _ = ev.run(lambda lq: got.append(m.inference(lq)) or got[-1])
    want = [OM.inference(lambda t: OM.edsr_forward(sd, t, cfg), lq, cfg["img_range"]) for lq, _ in pairs]
    psnr_ref = float(np.mean([OMT.compute_psnr(w, gt, y_only=True, crop_border=2) for w, (_, gt) in zip(want, pairs)]))
    for y, w in zip(got, want):
        assert y.dtype == np.uint8 and y.shape == (96, 96, 3)
        d = np.abs(y.astype(int) - w.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, (d.max(), (d > 0).mean())
    assert abs(psnr_hip - psnr_ref) <= 1e-3


FULL_DEPTH = [
    ("EDSR", dict(scale=4), OM.edsr_forward, False),            # config 2's model: 256 features, 32 resblocks
    ("RCAN", dict(scale=4), OM.rcan_forward, False),            # 10 groups x 20 RCABs
    ("HAT", dict(scale=4, drop_path_rate=0.0), OM.hat_forward, True),  # config 5's forward: 6 x (6 HAB + OCAB), ws 16, reflect-pad geometry
    # the reference's lightweight SwinIR (from_pretrained(light=True), swinir.py:418-427): 4 x 6 blocks on sr_swin_light (one launch per block), pixelshuffledirect
    ("SwinIR", dict(scale=4, embed_dim=60, depths=[6, 6, 6, 6], num_heads=[6, 6, 6, 6], upsampler="pixelshuffledirect"), OM.swinir_forward, False),
]


@pytest.mark.parametrize("kind,cfg,oracle_fwd,train", FULL_DEPTH, ids=["EDSR", "RCAN", "HAT", "SwinIR-light"])
def test_full_depth_one_tile_against_oracle(kind, cfg, oracle_fwd, train):
    """Default-depth models on ONE 64x64 LR tile against the CPU oracle: error growth over 65 convs / 200 RCABs / 42 HAT blocks.
    fp32 path <= 5e-5 of the output range; bf16 path: the metric's PSNR delta <= 1e-2 dB and >= 50 dB against the oracle output."""
    torch.manual_seed(1)
    m = getattr(S, kind)(**cfg)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if p_.ndim == 1 and "mean" not in n_:
                p_.add_(torch.randn_like(p_) * 0.05)
    sd = _sd_cpu(m)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 3, 64, 64, generator=g)
    tgt = (torch.rand(256, 256, 3, generator=g) * 255).round().to(torch.uint8).numpy()
    with torch.no_grad():
        ref = oracle_fwd(sd, x, m.get_model_config())
    p_ref = OMT.compute_psnr(_u8(ref), tgt, y_only=True, crop_border=4)
    m = m.to(DEV)
    m.train(train)  # HAT: the training forward's geometry (reflect pad, no-op at 64 = 4 x 16) with DropPath off
    rngv = max(1.0, float(ref.abs().max()))
    for prec, tol, dtol in (("fp32", 5e-5, 1e-3), ("fp32x3", 3e-4, 1e-3), ("bf16", 3e-2, 1e-2)):
        m.set_precision(prec)
        with torch.no_grad():
            y = m(x.to(DEV)).cpu()
        err = float((y - ref).abs().max())
        assert err <= tol * rngv, (kind, prec, err)
        assert abs(OMT.compute_psnr(_u8(y), tgt, y_only=True, crop_border=4) - p_ref) <= dtol, (kind, prec)
        if prec == "bf16":
            assert 10 * np.log10(rngv * rngv / float(((y - ref) ** 2).mean())) >= 50.0, kind
    m.cpu()


# ----------------------------------------------------------------------------- error behaviour
# ----------------------------------------------------------------------------- one image, row-strip sharded (section 8e, config 4)
def _randomised(m, seed=3):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p_ in m.named_parameters():  # _init_weights zeroes biases / sets LN to identity: make every term count
            if "sub_mean" in name or "add_mean" in name:  # frozen MeanShift (diagonal by construction, common.py:108-121)
                continue
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.02 if p_.dim() > 1 else 0.1) + (1.0 if p_.dim() == 1 and p_.numel() in (60, 180) else 0.0))
    return m


@pytest.mark.parametrize("cfg,prec", [
    (dict(scale=4, embed_dim=180, depths=[2, 2], num_heads=[6, 6]), "bf16"),   # fused one-launch block kernel
    (dict(scale=4, embed_dim=180, depths=[2, 2], num_heads=[6, 6]), "fp32"),   # exact-fp32 GEMM + window-attention path
    (dict(scale=2, embed_dim=60, depths=[2], num_heads=[6]), "bf16"),          # generic bf16 path
    (dict(scale=3, embed_dim=60, depths=[2, 2], num_heads=[6, 6], upsampler="pixelshuffledirect"), "fp32"),
])
def test_swinir_row_strips_with_halo_exchange_equal_the_unsharded_forward(cfg, prec):
    """Strips see exactly the operands the unsharded kernels see (same windows, same conv neighbourhoods), so the
    results must be IDENTICAL, not just close: 1, 2, 3 and 5 strips of uneven height, all in this process."""
    from studiosr_amd.strips import LocalStripComm

    torch.manual_seed(0)
    m = _randomised(S.SwinIR(**cfg)).to(DEV).eval().set_precision(prec)
    x = torch.rand(1, 3, 37, 52, device=DEV)  # eval pad -> 40 x 56: 5 window rows, 7 window columns
    with torch.no_grad():
        ref = m(x)
        for world in (1, 2, 3, 5):
            out = m.forward_strips(x, LocalStripComm(world))
            assert out.shape == ref.shape
            assert torch.equal(out, ref), f"{world} strips: max |diff| = {float((out - ref).abs().max())}"
    if prec == "fp32":  # and the unsharded forward itself is pinned to the oracle
        sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in m.state_dict().items()}
        o = OM.swinir_forward(sd, x.cpu(), m.get_model_config())
        assert float((ref.cpu() - o).abs().max()) <= FP32_TOL * max(1.0, float(o.abs().max()))


@pytest.mark.parametrize("prec,tol", [("fp32", FP32_TOL), ("bf16", BF16_TOL)])
def test_edsr_full_width_against_oracle(prec, tol):
    """Default-width EDSR (256 features: the wide-tile conv with its two-phase halo tile, PixelShuffle output, fp32 stream I/O)
    on an image whose sides are not multiples of the tile, against the CPU oracle."""
    torch.manual_seed(5)
    m = _randomised(S.EDSR(scale=2, n_resblocks=2), seed=5).to(DEV).eval().set_precision(prec)
    x = torch.rand(6, 3, 50, 70)
    with torch.no_grad():
        y = m(x.to(DEV)).cpu()
    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in m.state_dict().items()}
    ref = OM.edsr_forward(sd, x, m.get_model_config())
    assert y.shape == ref.shape == (6, 3, 100, 140)
    assert float((y - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("ws", [16, 8])
@pytest.mark.parametrize("prec,tol", [("fp32", FP32_TOL), ("bf16", BF16_TOL)])
def test_hat_full_width_against_oracle(prec, tol, ws):
    """Default-width HAT (embed 180: the K = 192 GEMMs incl. the OCA epilogue, flash window attention with 256 keys and the
    shift mask, flash overlapping cross attention with 576 keys, CAB convs + channel attention) against the CPU oracle."""
    torch.manual_seed(7)
    m = _randomised(S.HAT(scale=2, depths=[2], num_heads=[6], window_size=ws), seed=7).to(DEV).eval().set_precision(prec)
    x = torch.rand(2, 3, 3 * ws, 2 * ws)  # 3 x 2 windows per image: border and interior windows
    with torch.no_grad():
        y = m(x.to(DEV)).cpu()
    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in m.state_dict().items()}
    ref = OM.hat_forward(sd, x, m.get_model_config())
    assert y.shape == ref.shape
    assert float((y - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))


def test_swin_qkv_and_tail_kernels_equal_the_gemm_path(monkeypatch):
    """sr_swin_qkv (LayerNorm1 + QKV) and sr_swin_tail (projection + shortcut + gated CAB term + LayerNorm2 + MLP + the next block's
    LayerNorm1 in one launch), ABI v6, against the launches they replace (sr_gemm SR_EPI_QKV; sr_gemm with the gated second residual,
    sr_mlp_fused, sr_layernorm_to) inside the same HAT forward: 16 x 16 windows in four 64-token parts, shifted
    and unshifted blocks, a non-square image, the overlapping cross-attention block's tail; and against the CPU oracle."""
    torch.manual_seed(11)
    m = _randomised(S.HAT(scale=2, depths=[2, 2], num_heads=[6, 6], window_size=16), seed=11).to(DEV).eval().set_precision("bf16")
    x = torch.rand(2, 3, 48, 32)
    with torch.no_grad():
        monkeypatch.setenv("SR_SWIN_TAIL", "1")
        monkeypatch.setenv("SR_SWIN_QKV", "1")
        y1 = m(x.to(DEV)).cpu()
        monkeypatch.setenv("SR_SWIN_TAIL", "0")  # the projection GEMM + MLP kernel, LayerNorm1 launches for the conv branch
        monkeypatch.setenv("SR_SWIN_QKV", "0")   # the QKV GEMM
        monkeypatch.setenv("SR_CAB_FUSED", "0")  # two sr_conv3x3 launches + sr_channel_gate
        y0 = m(x.to(DEV)).cpu()
    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in m.state_dict().items()}
    ref = OM.hat_forward(sd, x, m.get_model_config())
    scale = max(1.0, float(ref.abs().max()))
    e1, e0 = float((y1 - ref).abs().max()) / scale, float((y0 - ref).abs().max()) / scale
    assert e1 <= BF16_TOL and e0 <= BF16_TOL
    assert float((y1 - y0).abs().max()) <= 0.5 * BF16_TOL * scale  # same operands, same products: only the summation order differs


@pytest.mark.parametrize("prec,tol", [("fp32", FP32_TOL), ("bf16", BF16_TOL), ("fp32x3", 3e-4)])
def test_rcan_full_width_against_oracle(prec, tol):
    """Default-width RCAN (64 features: in bf16 the conv-ReLU-conv of every RCAB is the one-launch sr_rcab_conv_pair with its
    14 x 14 tiles, zero-padded intermediate and per-tile pooling) on an image that is not a multiple of the tile."""
    torch.manual_seed(9)
    m = _randomised(S.RCAN(scale=2, n_resblocks=2, n_resgroups=2), seed=9).to(DEV).eval().set_precision(prec)
    x = torch.rand(3, 3, 33, 45)
    with torch.no_grad():
        y = m(x.to(DEV)).cpu()
    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in m.state_dict().items()}
    ref = OM.rcan_forward(sd, x, m.get_model_config())
    assert y.shape == ref.shape == (3, 3, 66, 90)
    assert float((y - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))


def test_rcab_pair_equals_two_conv_launches_bit_for_bit():
    """sr_rcab_conv_pair against conv3x3(ReLU) + conv3x3 through the same packed weights: identical bits, and the pooled
    channel sums equal the sum of y."""
    torch.manual_seed(11)
    B, H, W, Cc = 2, 30, 41, 64
    w1, w2 = torch.randn(Cc, Cc, 3, 3, device=DEV) * 0.05, torch.randn(Cc, Cc, 3, 3, device=DEV) * 0.05
    b1, b2 = torch.randn(Cc, device=DEV) * 0.1, torch.randn(Cc, device=DEV) * 0.1
    ident = packing.identity_idx(Cc, Cc)
    c1, c2 = packing.pack_conv3x3(w1, b1, Cc, ident, torch.bfloat16), packing.pack_conv3x3(w2, b2, Cc, ident, torch.bfloat16)
    x = torch.randn(B, H, W, Cc, device=DEV)
    mid = torch.empty(B, H, W, Cc, device=DEV, dtype=torch.bfloat16)
    want = torch.empty(B, H, W, Cc, device=DEV)
    conv_call(x, *c1, mid, torch.bfloat16, act=L.ACT_RELU)
    conv_call(mid, *c2, want, torch.bfloat16)
    got = torch.full_like(want, float("nan"))
    n_tiles = ops.rcab_pool_tiles(H, W)
    pool = torch.zeros(B, n_tiles, Cc, device=DEV)
    ops.rcab_conv_pair(x=x.data_ptr(), w1p=c1[0].data_ptr(), b1=c1[1].data_ptr(), w2p=c2[0].data_ptr(), b2=c2[1].data_ptr(), y=got.data_ptr(),
                       pool_partial=pool.data_ptr(), B=B, H=H, W=W, C_p=Cc, x_dtype=L.SR_F32, y_dtype=L.SR_F32)
    assert torch.equal(got, want)
    torch.testing.assert_close(pool.sum(dim=1), want.sum(dim=(1, 2)), rtol=1e-4, atol=1e-2)


def test_split_operand_rcab_pair_against_two_split_operand_conv_launches_and_conv2d():
    """ABI v11, precision "fp32x3" (what inference() runs, common.py:36-48): sr_rcab_conv_pair with compute_dtype SR_BF16X3 (every operand a hi + lo bf16 pair, 32-byte image
    cells, one workgroup per CU) against (a) the two split-operand sr_conv3x3 launches it replaces -- same K walk: identical bits -- and (b) torch conv2d in fp32
    (fp32-class accuracy); plus the gated form (the previous block's channel-attention tail folded into the halo staging) against sr_channel_attention + the plain pair."""
    from studiosr_amd.models.rcan import pack_ca, run_channel_attention
    from studiosr_amd.runtime import X3_KEY, x3_mode

    torch.manual_seed(13)
    B, H, W, Cc, Cr = 2, 30, 41, 64, 4
    w1, w2 = torch.randn(Cc, Cc, 3, 3, device=DEV) * 0.05, torch.randn(Cc, Cc, 3, 3, device=DEV) * 0.05
    b1, b2 = torch.randn(Cc, device=DEV) * 0.1, torch.randn(Cc, device=DEV) * 0.1
    ident = packing.identity_idx(Cc, Cc)
    c1, c2 = packing.pack_conv3x3(w1, b1, Cc, ident, X3_KEY), packing.pack_conv3x3(w2, b2, Cc, ident, X3_KEY)
    x = torch.randn(B, H, W, Cc, device=DEV)
    n_tiles = ops.rcab_pool_tiles(H, W)
    kw = dict(w1p=c1[0].data_ptr(), b1=c1[1].data_ptr(), w2p=c2[0].data_ptr(), b2=c2[1].data_ptr(), B=B, H=H, W=W, C_p=Cc, x_dtype=L.SR_F32, y_dtype=L.SR_F32)
    with x3_mode(True):
        mid, want = torch.empty(B, H, W, Cc, device=DEV), torch.empty(B, H, W, Cc, device=DEV)
        conv_call(x, *c1, mid, torch.float32, act=L.ACT_RELU)
        conv_call(mid, *c2, want, torch.float32)
        got, pool = torch.full_like(want, float("nan")), torch.zeros(B, n_tiles, Cc, device=DEV)
        ops.rcab_conv_pair(x=x.data_ptr(), y=got.data_ptr(), pool_partial=pool.data_ptr(), **kw)
        assert torch.equal(got, want), f"max diff {float((got - want).abs().max()):.3e}, nan {int(torch.isnan(got).sum())}"
        torch.testing.assert_close(pool.sum(dim=1), want.sum(dim=(1, 2)), rtol=1e-4, atol=1e-2)
        ref = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w1, b1, padding=1)), w2, b2, padding=1).permute(0, 2, 3, 1)
        assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
        # gated form
        ca = pack_ca(torch.randn(Cr, Cc, 1, 1, device=DEV) * 0.3, torch.randn(Cr, device=DEV) * 0.1, torch.randn(Cc, Cr, 1, 1, device=DEV) * 0.3,
                     torch.randn(Cc, device=DEV) * 0.1)
        x_want = torch.empty_like(x)
        run_channel_attention(ca, got, pool, n_tiles, Cc, x_want, skip=x)
        y_want, pool_want = torch.empty_like(x), torch.zeros_like(pool)
        ops.rcab_conv_pair(x=x_want.data_ptr(), y=y_want.data_ptr(), pool_partial=pool_want.data_ptr(), **kw)
        x_got, y_got, pool_got = torch.full_like(x, float("nan")), torch.full_like(x, float("nan")), torch.zeros_like(pool)
        w1c, b1c, w2c, b2c = ca
        ops.rcab_conv_pair(x=x.data_ptr(), gate_y=got.data_ptr(), gate_pool=pool.data_ptr(), gate_w1=w1c.data_ptr(), gate_b1=b1c.data_ptr(), gate_w2=w2c.data_ptr(),
                           gate_b2=b2c.data_ptr(), x_out=x_got.data_ptr(), gate_C=Cc, gate_Cr=Cr, y=y_got.data_ptr(), pool_partial=pool_got.data_ptr(), **kw)
        assert torch.equal(x_got, x_want), f"x_out: max diff {float((x_got - x_want).abs().max()):.3e}"
        assert torch.equal(y_got, y_want), f"y: max diff {float((y_got - y_want).abs().max()):.3e}"
        assert torch.equal(pool_got, pool_want)


def test_split_operand_cab_against_two_split_operand_conv_launches_and_conv2d():
    """ABI v11, precision "fp32x3": sr_cab_fused with dtype SR_BF16X3 (HAT's CAB, hat.py:41-49: conv 180 -> 60, GELU, conv 60 -> 180 on fp32 tensors, every operand a hi + lo
    bf16 pair, two-phase K walk) against the two split-operand sr_conv3x3 launches it replaces (their GELU is erff, the fused kernel's the 1.5e-7 erf: fp32-level
    agreement, not bits) and against torch conv2d + GELU in fp32; pool partials = sums of y.  An image that is no multiple of the 14 x 6 tile."""
    from studiosr_amd.runtime import X3_KEY, x3_mode

    torch.manual_seed(14)
    B, H, W, Ci, Cm = 2, 33, 45, 180, 60
    w1, w2 = torch.randn(Cm, Ci, 3, 3, device=DEV) * 0.03, torch.randn(Ci, Cm, 3, 3, device=DEV) * 0.05
    b1, b2 = torch.randn(Cm, device=DEV) * 0.1, torch.randn(Ci, device=DEV) * 0.1
    c1 = packing.pack_conv3x3(w1, b1, 192, packing.identity_idx(Cm, 64), X3_KEY)
    c2 = packing.pack_conv3x3(w2, b2, 64, packing.identity_idx(Ci, 192), X3_KEY)
    x = torch.zeros(B, H, W, 192, device=DEV)
    x[..., :Ci] = torch.randn(B, H, W, Ci, device=DEV)
    with x3_mode(True):
        mid, want = torch.empty(B, H, W, 64, device=DEV), torch.empty(B, H, W, 192, device=DEV)
        conv_call(x, *c1, mid, torch.float32, act=L.ACT_GELU)
        conv_call(mid, *c2, want, torch.float32)
        assert ops.cab_supported(192, 64, 192, L.SR_BF16X3)
        n_tiles = ops.cab_pool_tiles_rows(H, W, 0)
        got, pool = torch.full_like(want, float("nan")), torch.zeros(B, n_tiles, 192, device=DEV)
        ops.cab_fused(x=x.data_ptr(), w1p=c1[0].data_ptr(), b1=c1[1].data_ptr(), w2p=c2[0].data_ptr(), b2=c2[1].data_ptr(), y=got.data_ptr(), pool_partial=pool.data_ptr(),
                      B=B, H=H, W=W, Cin_p=192, Cmid_p=64, Cout_p=192, dtype=L.SR_BF16X3, tile_rows=0)
    assert not bool(torch.isnan(got).any())
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= 5e-6 * scale, float((got - want).abs().max())
    ref = torch.nn.functional.conv2d(torch.nn.functional.gelu(torch.nn.functional.conv2d(x[..., :Ci].permute(0, 3, 1, 2), w1, b1, padding=1)), w2, b2, padding=1).permute(0, 2, 3, 1)
    assert float((got[..., :Ci] - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    torch.testing.assert_close(pool.sum(dim=1), got.sum(dim=(1, 2)), rtol=1e-4, atol=1e-2)


def test_gated_rcab_equals_channel_attention_then_conv_pair():
    """ABI v4 gated input of sr_rcab_conv_pair: x_eff = x + gate * y_prev folded into the halo staging must give the bits of the
    two-launch sequence (sr_channel_attention -> plain sr_rcab_conv_pair): same skip tensor (x_out), same y, same pool partials."""
    from studiosr_amd.models.rcan import pack_ca, run_channel_attention

    torch.manual_seed(12)
    B, H, W, Cc, Cr = 2, 30, 41, 64, 4
    w1, w2 = torch.randn(Cc, Cc, 3, 3, device=DEV) * 0.05, torch.randn(Cc, Cc, 3, 3, device=DEV) * 0.05
    b1, b2 = torch.randn(Cc, device=DEV) * 0.1, torch.randn(Cc, device=DEV) * 0.1
    ident = packing.identity_idx(Cc, Cc)
    c1, c2 = packing.pack_conv3x3(w1, b1, Cc, ident, torch.bfloat16), packing.pack_conv3x3(w2, b2, Cc, ident, torch.bfloat16)
    ca = pack_ca(torch.randn(Cr, Cc, 1, 1, device=DEV) * 0.3, torch.randn(Cr, device=DEV) * 0.1, torch.randn(Cc, Cr, 1, 1, device=DEV) * 0.3,
                 torch.randn(Cc, device=DEV) * 0.1)
    n_tiles = ops.rcab_pool_tiles(H, W)
    x_prev = torch.randn(B, H, W, Cc, device=DEV)
    y_prev = torch.empty(B, H, W, Cc, device=DEV)
    pool_prev = torch.zeros(B, n_tiles, Cc, device=DEV)
    kw = dict(w1p=c1[0].data_ptr(), b1=c1[1].data_ptr(), w2p=c2[0].data_ptr(), b2=c2[1].data_ptr(), B=B, H=H, W=W, C_p=Cc, x_dtype=L.SR_F32, y_dtype=L.SR_F32)
    ops.rcab_conv_pair(x=x_prev.data_ptr(), y=y_prev.data_ptr(), pool_partial=pool_prev.data_ptr(), **kw)  # a realistic (y, pool) pair
    # two launches
    x_want = torch.empty_like(x_prev)
    run_channel_attention(ca, y_prev, pool_prev, n_tiles, Cc, x_want, skip=x_prev)
    y_want, pool_want = torch.empty_like(x_prev), torch.zeros_like(pool_prev)
    ops.rcab_conv_pair(x=x_want.data_ptr(), y=y_want.data_ptr(), pool_partial=pool_want.data_ptr(), **kw)
    # one launch
    x_got, y_got, pool_got = torch.full_like(x_prev, float("nan")), torch.full_like(x_prev, float("nan")), torch.zeros_like(pool_prev)
    w1c, b1c, w2c, b2c = ca
    ops.rcab_conv_pair(x=x_prev.data_ptr(), gate_y=y_prev.data_ptr(), gate_pool=pool_prev.data_ptr(), gate_w1=w1c.data_ptr(), gate_b1=b1c.data_ptr(),
                       gate_w2=w2c.data_ptr(), gate_b2=b2c.data_ptr(), x_out=x_got.data_ptr(), gate_C=Cc, gate_Cr=Cr, y=y_got.data_ptr(),
                       pool_partial=pool_got.data_ptr(), **kw)
    assert torch.equal(x_got, x_want), f"x_out: max diff {float((x_got - x_want).abs().max()):.3e}, nan {int(torch.isnan(x_got).sum())}"
    assert torch.equal(y_got, y_want), f"y: max diff {float((y_got - y_want).abs().max()):.3e}"
    assert torch.equal(pool_got, pool_want)
    with pytest.raises(L.HipLibraryError):  # aliasing the skip output with the input is refused
        ops.rcab_conv_pair(x=x_prev.data_ptr(), gate_y=y_prev.data_ptr(), gate_pool=pool_prev.data_ptr(), gate_w1=w1c.data_ptr(), gate_b1=b1c.data_ptr(),
                           gate_w2=w2c.data_ptr(), gate_b2=b2c.data_ptr(), x_out=x_prev.data_ptr(), gate_C=Cc, gate_Cr=Cr, y=y_got.data_ptr(),
                           pool_partial=pool_got.data_ptr(), **kw)


@pytest.mark.parametrize("cin,shape", [(64, (3, 37, 45)), (256, (2, 19, 50)), (64, (1, 8, 16))])
def test_rgb_tail_conv_persistent_kernel_against_conv2d(cin, shape):
    """sr_conv_narrow.hip (persistent workgroups, register-resident weights, K split over waves for 256 channels) through sr_conv3x3's
    final-NCHW mode, on images that are no multiple of the tile in either direction, against F.conv2d on the bf16-rounded operands
    (fp32 accumulate on both sides: only the summation order differs) incl. the un-normalise affine and the crop."""
    torch.manual_seed(cin)
    B, H, W = shape
    w = (torch.randn(3, cin, 3, 3) * 0.05).to(DEV)
    b = torch.randn(3).to(DEV)
    x = torch.randn(B, H, W, cin, device=DEV).to(torch.bfloat16)
    wp, bp = packing.pack_conv3x3(w, b, cin, packing.identity_idx(3, 16), torch.bfloat16)
    fs, fb = torch.tensor([2.0, 0.5, 1.5], device=DEV), torch.tensor([0.1, -0.2, 0.3], device=DEV)
    fh, fw = H - 1, W - 2  # cropped output
    out = torch.full((B, 3, fh, fw), float("nan"), device=DEV)
    conv_call(x, wp, bp, out, torch.bfloat16, out_mode=L.OUT_FINAL_NCHW, fin=(fs, fb, 3, fh, fw), cout_p=16)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), b, padding=1)
    ref = (ref * fs.view(1, 3, 1, 1) + fb.view(1, 3, 1, 1))[:, :, :fh, :fw]
    assert not torch.isnan(out).any()
    assert float((out - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("ws,prec", [(8, "bf16"), (16, "bf16"), (8, "fp32x3"), (16, "fp32x3")])
def test_swin_qkv_and_tail_kernels_at_both_window_sizes_against_the_gemm_launches(ws, prec, monkeypatch):
    """run_window_msa + MLP of one packed HAT block on a stream tensor, with the stream-form kernels (sr_swin_qkv, sr_swin_tail) and with the
    launches they replace (sr_gemm SR_EPI_QKV, sr_gemm projection, sr_mlp_fused): windows of 8 x 8 (one 64-token part) and 16 x 16 (four
    parts), shifted and unshifted, separate skip tensor (so that the one-kernel attention half of the 8 x 8 geometry is not taken); bf16 operands and
    the split-operand instantiations of precision "fp32x3" (fp32 q / k / v^T / o; agreement at fp32 level)."""
    from studiosr_amd.models import swinir as SW
    from studiosr_amd.runtime import Workspace, x3_mode

    torch.manual_seed(31)
    m = _randomised(S.HAT(scale=2, depths=[2], num_heads=[6], window_size=ws), seed=31).to(DEV).eval().set_precision(prec)
    cdt = torch.bfloat16 if prec == "bf16" else torch.float32
    lp = m._get_packed(cdt)["layers"][0]
    geo = lp["geo"]
    B, H, W = 2, 3 * ws, 2 * ws
    t_in = torch.randn(B, H, W, geo.Cp, device=DEV)
    t_in[..., geo.C:] = 0
    skip = torch.randn(B, H, W, geo.Cp, device=DEV)
    skip[..., geo.C:] = 0
    for bp in lp["blocks"]:  # shift 0 and ws / 2
        outs = []
        for flag in ("1", "0"):
            monkeypatch.setenv("SR_SWIN_QKV", flag)
            monkeypatch.setenv("SR_SWIN_TAIL", flag)
            S.runtime.reset_knobs()  # (the switches are read once per forward; this test drives the block helpers directly)
            out = torch.full_like(t_in, float("nan"))
            with x3_mode(prec == "fp32x3"):
                used = SW.run_window_msa(bp, bp["ln1"], geo, t_in, out, skip, Workspace(DEV), cdt, bp["shift"], name=f"t{flag}", with_mlp=True)
                assert (used == "tail") == (flag == "1")
                if used != "tail":
                    SW.run_mlp(bp, bp["ln2"], geo, out, Workspace(DEV), cdt)
            torch.cuda.synchronize()
            outs.append(out.clone())
        new, old = outs
        assert not torch.isnan(new).any() and float(new[..., geo.C:].abs().max()) == 0.0
        scale = float(old.abs().max())
        tol, rms = (1.0e-2, 1.5e-3) if prec == "bf16" else (3.0e-5, 5.0e-6)
        assert float((new - old).abs().max()) <= tol * scale, (f"shift {bp['shift']}", float((new - old).abs().max()) / scale)
        assert float((new - old).pow(2).mean().sqrt()) <= rms * scale


@pytest.mark.parametrize("ws,wg_tokens", [(8, 64), (16, 64), (8, 32), (16, 32)])
def test_swin_tail_fused_next_block_qkv_equals_a_separate_qkv_launch(ws, wg_tokens):
    """sr_swin_tail's optional last stage (LayerNorm1 + QKV of the NEXT block on the same tokens, scattered into that block's window order:
    q2 / k2 / vt2, shift2) against sr_swin_qkv run on the tail's output with the next block's stream: same bits, both shift orders."""
    from studiosr_amd.models import swinir as SW

    torch.manual_seed(33)
    m = _randomised(S.HAT(scale=2, depths=[2], num_heads=[6], window_size=ws), seed=33).to(DEV).eval().set_precision("bf16")
    cdt = torch.bfloat16
    lp = m._get_packed(cdt)["layers"][0]
    geo = lp["geo"]
    b0, b1 = lp["blocks"]  # shifts 0 and ws / 2
    B, H, W = 2, 3 * ws, 2 * ws
    M = B * H * W
    nb = M // geo.ntok
    skip = torch.randn(B, H, W, geo.Cp, device=DEV)
    skip[..., geo.C:] = 0
    o = (torch.randn(M, geo.HP, device=DEV) * 0.5).to(cdt)
    o.view(M, geo.heads, geo.hd_p)[..., geo.hd:] = 0
    for cur, nxt in ((b0, b1), (b1, b0)):
        stream = torch.cat([cur["tail_stream"], nxt["qkv_stream"]]).contiguous()
        out = torch.full_like(skip, float("nan"))
        q2 = torch.full((nb, geo.heads, geo.ntok, geo.hd_p), float("nan"), device=DEV).to(cdt)
        k2, vt2 = torch.full_like(q2, float("nan")), torch.full((nb, geo.heads, geo.hd_p, geo.ntok), float("nan"), device=DEV).to(cdt)
        ops.swin_tail(x=skip.data_ptr(), out=out.data_ptr(), o=o.data_ptr(), wstream=stream.data_ptr(), bproj=cur["proj_b"].data_ptr(), B=B, H=H, W=W,
                      C=geo.C, Cp=geo.Cp, ldx=geo.Cp, heads=geo.heads, hd_p=geo.hd_p, ws=geo.ws, shift=cur["shift"], Hp=geo.hid_p, eps=1e-5,
                      y_mode=L.Y_ROLL, compute_dtype=L.SR_BF16, q2=q2.data_ptr(), k2=k2.data_ptr(), vt2=vt2.data_ptr(), shift2=nxt["shift"], wg_tokens=wg_tokens)
        want = [torch.full_like(q2, float("nan")), torch.full_like(k2, float("nan")), torch.full_like(vt2, float("nan"))]
        ops.swin_qkv(x=out.data_ptr(), q=want[0].data_ptr(), k=want[1].data_ptr(), vt=want[2].data_ptr(), wstream=nxt["qkv_stream"].data_ptr(), B=B, H=H, W=W,
                     C=geo.C, Cp=geo.Cp, ldx=geo.Cp, heads=geo.heads, hd_p=geo.hd_p, ws=geo.ws, shift=nxt["shift"], eps=1e-5, y_mode=L.Y_ROLL,
                     compute_dtype=L.SR_BF16)
        torch.cuda.synchronize()
        plain = torch.full_like(skip, float("nan"))  # and the stream output itself does not depend on the fused stage
        ops.swin_tail(x=skip.data_ptr(), out=plain.data_ptr(), o=o.data_ptr(), wstream=cur["tail_stream"].data_ptr(), bproj=cur["proj_b"].data_ptr(), B=B, H=H,
                      W=W, C=geo.C, Cp=geo.Cp, ldx=geo.Cp, heads=geo.heads, hd_p=geo.hd_p, ws=geo.ws, shift=cur["shift"], Hp=geo.hid_p, eps=1e-5,
                      y_mode=L.Y_ROLL, compute_dtype=L.SR_BF16, wg_tokens=64)
        assert torch.equal(out, plain) and not torch.isnan(out).any()
        for got, ref, name in zip((q2, k2, vt2), want, "q k vt".split()):
            assert not torch.isnan(got.float()).any(), name
            assert torch.equal(got, ref), f"{name}: shift {cur['shift']} -> {nxt['shift']}, max diff {float((got.float() - ref.float()).abs().max()):.3e}"


def test_swin_qkv_fragment_order_is_a_permutation_of_the_row_major_layouts():
    """frag_order = 1 of sr_swin_qkv and of sr_swin_tail's fused QKV stage (what sr_window_attention reads with qkv_frag = 1): q / k as
    [16-token tile][g][i][8] = token 16 tile + i, features 8 g ..; v^T as [64-key block][d tile][32-key step][g][i][8] = d 16 dt + i, key 64 kb + 32 ks +
    16 (e >> 2) + 4 g + (e & 3) -- exactly the row-major tensors, permuted; and both producers agree bit for bit."""
    ws = 16
    torch.manual_seed(37)
    m = _randomised(S.HAT(scale=2, depths=[2], num_heads=[6], window_size=ws), seed=37).to(DEV).eval().set_precision("bf16")
    cdt = torch.bfloat16
    lp = m._get_packed(cdt)["layers"][0]
    geo = lp["geo"]
    b0, b1 = lp["blocks"]
    B, H, W = 2, 2 * ws, 3 * ws
    M = B * H * W
    nb = M // geo.ntok
    t = torch.randn(B, H, W, geo.Cp, device=DEV)
    t[..., geo.C:] = 0

    def qkv(bp, frag):
        q = torch.full((nb, geo.heads, geo.ntok, geo.hd_p), float("nan"), device=DEV).to(cdt)
        k, vt = torch.full_like(q, float("nan")), torch.full((nb, geo.heads, geo.hd_p, geo.ntok), float("nan"), device=DEV).to(cdt)
        ops.swin_qkv(x=t.data_ptr(), q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), wstream=bp["qkv_stream"].data_ptr(), B=B, H=H, W=W, C=geo.C, Cp=geo.Cp,
                     ldx=geo.Cp, heads=geo.heads, hd_p=geo.hd_p, ws=geo.ws, shift=bp["shift"], eps=1e-5, y_mode=L.Y_ROLL, compute_dtype=L.SR_BF16, frag_order=frag)
        torch.cuda.synchronize()
        return q, k, vt

    for bp in (b0, b1):
        q0, k0, v0 = qkv(bp, 0)
        q1, k1, v1 = qkv(bp, 1)
        for row, frag in ((q0, q1), (k0, k1)):
            want = row.view(nb, geo.heads, 16, 16, 4, 8).permute(0, 1, 2, 4, 3, 5)  # [tile][i][g][e] -> [tile][g][i][e]
            assert torch.equal(frag.view(nb, geo.heads, 16, 4, 16, 8), want)
        want = v0.view(nb, geo.heads, 2, 16, 4, 2, 2, 4, 4).permute(0, 1, 4, 2, 5, 7, 3, 6, 8)  # [dt][i][kb][ks][hi][g][lo] -> [kb][dt][ks][g][i][hi][lo]
        assert torch.equal(v1.view(nb, geo.heads, 4, 2, 2, 4, 16, 2, 4), want)
    # the fused stage of sr_swin_tail writes the same fragment-order tensors as sr_swin_qkv on its output
    skip = torch.randn(B, H, W, geo.Cp, device=DEV)
    skip[..., geo.C:] = 0
    o = (torch.randn(M, geo.HP, device=DEV) * 0.5).to(cdt)
    o.view(M, geo.heads, geo.hd_p)[..., geo.hd:] = 0
    out = torch.empty_like(skip)
    q2 = torch.full((nb, geo.heads, geo.ntok, geo.hd_p), float("nan"), device=DEV).to(cdt)
    k2, vt2 = torch.full_like(q2, float("nan")), torch.full((nb, geo.heads, geo.hd_p, geo.ntok), float("nan"), device=DEV).to(cdt)
    stream = torch.cat([b0["tail_stream"], b1["qkv_stream"]]).contiguous()
    ops.swin_tail(x=skip.data_ptr(), out=out.data_ptr(), o=o.data_ptr(), wstream=stream.data_ptr(), bproj=b0["proj_b"].data_ptr(), B=B, H=H, W=W, C=geo.C, Cp=geo.Cp,
                  ldx=geo.Cp, heads=geo.heads, hd_p=geo.hd_p, ws=geo.ws, shift=b0["shift"], Hp=geo.hid_p, eps=1e-5, y_mode=L.Y_ROLL, compute_dtype=L.SR_BF16,
                  q2=q2.data_ptr(), k2=k2.data_ptr(), vt2=vt2.data_ptr(), shift2=b1["shift"], frag_order=1)
    t = out
    qw, kw, vw = qkv(b1, 1)
    assert torch.equal(q2, qw) and torch.equal(k2, kw) and torch.equal(vt2, vw)


@pytest.mark.parametrize("ws", [8, 16])
def test_swin_qkv_overlapping_cross_attention_layouts_against_the_gemm_epilogue(ws):
    """sr_swin_qkv with oca_pad > 0 (hat.py:247-264: k in zero-bordered image order, v^T in zero-bordered planes, q in window order) against
    sr_gemm's SR_EPI_QKV_OCA on the same stream tensor: same values to bf16 rounding, and the zero border is never written."""
    torch.manual_seed(35)
    m = _randomised(S.HAT(scale=2, depths=[1], num_heads=[6], window_size=ws), seed=35).to(DEV).eval().set_precision("bf16")
    cdt = torch.bfloat16
    P = m._get_packed(cdt)
    op, geo, e = P["layers"][0]["ocab"], P["layers"][0]["geo"], P["border"]
    assert e % 4 == 0 and "qkv_stream" in op
    B, H, W = 2, 2 * ws, 3 * ws
    M = B * H * W
    nb = M // geo.ntok
    t = torch.randn(B, H, W, geo.Cp, device=DEV)
    t[..., geo.C:] = 0

    def buffers():
        return (torch.zeros(nb, geo.heads, geo.ntok, geo.hd_p, device=DEV).to(cdt), torch.zeros(B, H + 2 * e, W + 2 * e, geo.heads, geo.hd_p, device=DEV).to(cdt),
                torch.zeros(B, geo.heads, geo.hd_p, H + 2 * e, W + 2 * e, device=DEV).to(cdt))

    q1, k1, v1 = buffers()
    ops.swin_qkv(x=t.data_ptr(), q=q1.data_ptr(), k=k1.data_ptr(), vt=v1.data_ptr(), wstream=op["qkv_stream"].data_ptr(), B=B, H=H, W=W, C=geo.C, Cp=geo.Cp,
                 ldx=geo.Cp, heads=geo.heads, hd_p=geo.hd_p, ws=geo.ws, shift=0, eps=1e-5, y_mode=L.Y_ROLL, compute_dtype=L.SR_BF16, oca_pad=e)
    q0, k0, v0 = buffers()
    ops.gemm(A=t.data_ptr(), Wp=op["qkv_w"].data_ptr(), bias=op["qkv_b"].data_ptr(), ln_gamma=None, ln_beta=None, ln_norm_only=1, out=q0.data_ptr(),
             out_k=k0.data_ptr(), out_vt=v0.data_ptr(), M=M, K=geo.Cp, N=3 * geo.HP, k_real=geo.C, lda=geo.Cp, a_dtype=L.SR_F32, out_dtype=L.SR_BF16,
             compute_dtype=L.SR_BF16, act=L.ACT_NONE, out_scale=1.0, a_map=L.MAP_WINDOW, o_map=L.MAP_IDENTITY, H=H, W=W, ws=geo.ws, shift=0,
             epi=L.EPI_QKV_OCA, heads=geo.heads, hd_p=geo.hd_p, ntok=geo.ntok, ln_eps=1e-5, oca_pad=e)
    torch.cuda.synchronize()
    for got, ref, name in ((q1, q0, "q"), (k1, k0, "k"), (v1, v0, "vt")):
        g, r = got.float(), ref.float()
        assert float((g - r).abs().max()) <= 2.0e-2 * max(1.0, float(r.abs().max())), name
    assert float(k1[:, :e].abs().max()) == 0.0 and float(k1[:, -e:].abs().max()) == 0.0 and float(k1[:, :, :e].abs().max()) == 0.0 and float(k1[:, :, -e:].abs().max()) == 0.0
    assert float(v1[..., :e, :].abs().max()) == 0.0 and float(v1[..., -e:, :].abs().max()) == 0.0 and float(v1[..., :e].abs().max()) == 0.0 and float(v1[..., -e:].abs().max()) == 0.0
    assert float(k1[:, e:-e, e:-e].abs().max()) > 0.0 and float(v1[..., e:-e, e:-e].abs().max()) > 0.0


def test_cab_fused_equals_two_conv_launches_and_a_torch_reference():
    """sr_cab_fused (hat.py:41-49: conv 180 -> 60, GELU, conv 60 -> 180 in one launch, intermediate in LDS, per-tile pool sums) against the
    two sr_conv3x3 launches it replaces (same packed weights) and against torch convs on the bf16-rounded operands, on an image that is not a
    multiple of the 14 x 6 tile; the pool partials must add up to the channel sums of y."""
    from studiosr_amd.models.common import conv_call

    torch.manual_seed(23)
    B, H, W, C, Cp, c3, c3p = 2, 37, 50, 180, 192, 60, 64
    w1, b1 = torch.randn(c3, C, 3, 3, device=DEV) * 0.03, torch.randn(c3, device=DEV) * 0.1
    w2, b2 = torch.randn(C, c3, 3, 3, device=DEV) * 0.05, torch.randn(C, device=DEV) * 0.1
    p1 = packing.pack_conv3x3(w1, b1, Cp, packing.identity_idx(c3, c3p), torch.bfloat16)
    p2 = packing.pack_conv3x3(w2, b2, c3p, packing.identity_idx(C, Cp), torch.bfloat16)
    x = torch.randn(B, H, W, Cp, device=DEV).to(torch.bfloat16)
    x[..., C:] = 0
    assert ops.cab_supported(Cp, c3p, Cp, L.SR_BF16)
    nt = ops.cab_pool_tiles(H, W)
    assert nt == 4 * 7
    y = torch.full((B, H, W, Cp), float("nan"), device=DEV).to(torch.bfloat16)
    pool = torch.full((B, nt, Cp), float("nan"), device=DEV)
    ops.cab_fused(x=x.data_ptr(), w1p=p1[0].data_ptr(), b1=p1[1].data_ptr(), w2p=p2[0].data_ptr(), b2=p2[1].data_ptr(), y=y.data_ptr(),
                  pool_partial=pool.data_ptr(), B=B, H=H, W=W, Cin_p=Cp, Cmid_p=c3p, Cout_p=Cp, dtype=L.SR_BF16)
    mid = torch.empty(B, H, W, c3p, device=DEV).to(torch.bfloat16)
    want = torch.empty_like(y)
    conv_call(x, *p1, mid, torch.bfloat16, act=L.ACT_GELU)
    conv_call(mid, *p2, want, torch.bfloat16)
    torch.cuda.synchronize()
    yf, wf = y.float(), want.float()
    assert not torch.isnan(yf).any() and not torch.isnan(pool).any()
    assert float(yf[..., C:].abs().max()) == 0.0
    scale = float(wf.abs().max())
    assert float((yf - wf).abs().max()) <= 1.0e-2 * scale  # the intermediate is rounded to bf16 in both; GELU: erf polynomial vs erff
    # torch reference on the same rounded operands (fp32 accumulate, bf16 intermediate)
    xr = x.float()[..., :C].permute(0, 3, 1, 2)
    r1 = torch.nn.functional.gelu(torch.nn.functional.conv2d(xr, w1.to(torch.bfloat16).float(), b1, padding=1)).to(torch.bfloat16).float()
    ref = torch.nn.functional.conv2d(r1, w2.to(torch.bfloat16).float(), b2, padding=1).permute(0, 2, 3, 1)
    assert float((yf[..., :C] - ref).abs().max()) <= 1.5e-2 * float(ref.abs().max())
    # pool partials: fp32 sums of the un-rounded outputs over each tile's valid pixels
    assert torch.allclose(pool.sum(1)[:, :C], ref.sum((1, 2)), rtol=2e-2, atol=2e-2 * float(ref.abs().max()) * 8)


@pytest.mark.parametrize("frag", [0, 1])
def test_hab_mid_one_launch_equals_the_attention_and_cab_launches(frag):
    """sr_hab_mid (ABI v8; hat.py:165-176: CAB and window attention of a HAB as ONE launch, CAB tiles first) against sr_window_attention and
    sr_cab_fused on the same operands: the attention role is the same code on the same data (bit-identical output, shifted window mask
    included); the CAB role sums conv1's K in two phases (fp32 order only), its pool partials keep sr_cab_pool_tiles' layout."""
    torch.manual_seed(41)
    B, H, W, C, Cp, c3, c3p, heads, hd_p, ws = 2, 32, 48, 180, 192, 60, 64, 6, 32, 16
    ntok, nb = ws * ws, B * (H // ws) * (W // ws)
    w1, b1 = torch.randn(c3, C, 3, 3, device=DEV) * 0.03, torch.randn(c3, device=DEV) * 0.1
    w2, b2 = torch.randn(C, c3, 3, 3, device=DEV) * 0.05, torch.randn(C, device=DEV) * 0.1
    p1 = packing.pack_conv3x3(w1, b1, Cp, packing.identity_idx(c3, c3p), torch.bfloat16)
    p2 = packing.pack_conv3x3(w2, b2, c3p, packing.identity_idx(C, Cp), torch.bfloat16)
    x = torch.randn(B, H, W, Cp, device=DEV).to(torch.bfloat16)
    x[..., C:] = 0
    q = (torch.randn(nb, heads, ntok, hd_p, device=DEV) * 0.3).to(torch.bfloat16)  # (fragment order is a permutation of these: any values do)
    k = torch.randn(nb, heads, ntok, hd_p, device=DEV).to(torch.bfloat16)
    vt = torch.randn(nb, heads, hd_p, ntok, device=DEV).to(torch.bfloat16)
    bias = torch.randn(heads, ntok, ntok, device=DEV)
    bias_frag = packing.bias_fragments(bias)
    nt = ops.cab_pool_tiles(H, W)
    assert ops.hab_mid_supported(ntok, hd_p, ws, L.SR_BF16, Cp, c3p, Cp, L.SR_BF16)

    def run(one_launch):
        o = torch.full((nb * ntok, heads * hd_p), float("nan"), device=DEV).to(torch.bfloat16)
        y = torch.full((B, H, W, Cp), float("nan"), device=DEV).to(torch.bfloat16)
        pool = torch.full((B, nt, Cp), float("nan"), device=DEV)
        akw = dict(q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), bias=bias.data_ptr(), out=o.data_ptr(), n_bwin=nb, heads=heads, hd_p=hd_p, ntok=ntok,
                   H=H, W=W, ws=ws, shift=ws // 2, dtype=L.SR_BF16, y_mode=L.Y_ROLL, bias_frag=bias_frag.data_ptr(), qkv_frag=frag)
        ckw = dict(x=x.data_ptr(), w1p=p1[0].data_ptr(), b1=p1[1].data_ptr(), w2p=p2[0].data_ptr(), b2=p2[1].data_ptr(), y=y.data_ptr(),
                   pool_partial=pool.data_ptr(), B=B, H=H, W=W, Cin_p=Cp, Cmid_p=c3p, Cout_p=Cp, dtype=L.SR_BF16)
        if one_launch:
            ops.hab_mid(akw, ckw)
        else:
            ops.window_attention(**akw)
            ops.cab_fused(**ckw)
        torch.cuda.synchronize()
        return o.float(), y.float(), pool

    o1, y1, pool1 = run(True)
    o0, y0, pool0 = run(False)
    assert not torch.isnan(o1).any() and not torch.isnan(y1).any() and not torch.isnan(pool1).any()
    assert torch.equal(o1, o0)
    scale = float(y0.abs().max())
    assert float((y1 - y0).abs().max()) <= 1.0e-2 * scale and float((y1 - y0).pow(2).mean().sqrt()) <= 1.0e-3 * scale
    assert float(y1[..., C:].abs().max()) == 0.0
    assert torch.allclose(pool1, pool0, rtol=1e-3, atol=1e-3 * scale * 84)
    with pytest.raises(RuntimeError):  # the flash form needs the fragment-ordered bias
        akw = dict(q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), bias=bias.data_ptr(), out=o1.data_ptr(), n_bwin=nb, heads=heads, hd_p=hd_p, ntok=ntok,
                   H=H, W=W, ws=ws, shift=0, dtype=L.SR_BF16, y_mode=L.Y_ROLL)
        ops.hab_mid(akw, dict(x=x.data_ptr()))


@pytest.mark.parametrize("ws,shift", [(16, 0), (16, 8), (8, 4)])
def test_window_attention_split_operand_form_against_exact_fp32_and_torch(ws, shift):
    """ABI v11: sr_window_attention with dtype SR_BF16X3 (precision "fp32x3", what inference() runs: fp32 q / k / v^T / out, every product hi*hi + hi*lo + lo*hi on the bf16
    matrix cores, fp32 softmax) against the exact-fp32 instantiation of the same flash kernel and against torch in fp64 (swinir.py:83-102, hat.py:85-110 incl. the shift mask)."""
    from oracle import functional as OF
    from studiosr_amd.models.hat import rpi_sa
    from studiosr_amd.runtime import x3_mode

    torch.manual_seed(44)
    B, H, W, heads, hd_p = 2, 32, 48, 6, 32
    ntok, nb = ws * ws, B * (H // ws) * (W // ws)
    q = torch.randn(nb, heads, ntok, hd_p, device=DEV) * 0.4
    k = torch.randn(nb, heads, ntok, hd_p, device=DEV)
    vt = torch.randn(nb, heads, hd_p, ntok, device=DEV)
    for t_ in (q, k):
        t_[..., 30:] = 0
    table = torch.randn((2 * ws - 1) ** 2, heads, device=DEV)
    rpi = rpi_sa(ws) if ws == 16 else S.SwinIR(depths=[2], num_heads=[6]).layers[0].residual_group.blocks[0].attn.relative_position_index
    bias = packing.gather_bias(table, rpi, ntok, ntok)
    bias_frag = packing.bias_fragments(bias)

    def run(x3):
        o = torch.full((nb * ntok, heads * hd_p), float("nan"), device=DEV)
        with x3_mode(x3):
            ops.window_attention(q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), bias=bias.data_ptr(), out=o.data_ptr(), n_bwin=nb, heads=heads, hd_p=hd_p, ntok=ntok,
                                 H=H, W=W, ws=ws, shift=shift, dtype=L.SR_F32, y_mode=L.Y_ROLL, bias_frag=bias_frag.data_ptr(), qkv_frag=0, bias_tiles=None)
        torch.cuda.synchronize()
        return o

    new, old = run(True), run(False)
    assert not torch.isnan(new).any()
    scale = float(old.abs().max())
    assert float((new - old).abs().max()) <= 2e-5 * scale, float((new - old).abs().max()) / scale
    s64 = torch.einsum("bhqd,bhkd->bhqk", q.double(), k.double()) + bias[None].double()
    if shift:
        mask = OF.calculate_mask(H, W, ws, shift).to(DEV).double()
        s64 = (s64.reshape(B, -1, heads, ntok, ntok) + mask[None, :, None]).reshape(nb, heads, ntok, ntok)
    ref = torch.einsum("bhqk,bhdk->bqhd", torch.softmax(s64, -1), vt.double()).reshape(nb * ntok, heads * hd_p).float()
    assert float((new - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("frag,shift,y_mode", [(0, 0, "roll"), (0, 8, "roll"), (1, 8, "roll"), (1, 4, "strip")])
def test_window_attention_lds_form_equals_the_flash_form(frag, shift, y_mode):
    """sr_window_attention with SrWindowAttn.bias_tiles (ABI v8: K / V^T / the 31 distinct bias tiles of a head staged in LDS once per (window, head)) against the
    register-only flash form on the same operands (hat.py:85-110): same products, same online softmax over key blocks of 64, so the outputs agree to bf16
    rounding of o; also inside sr_hab_mid.  The bias is a real relative-position bias (table[rpi_sa(16)]); a bias without that structure is refused by the packer."""
    from studiosr_amd.models.hat import rpi_sa

    torch.manual_seed(43)
    B, H, W, heads, hd_p, ws = 2, 32, 48, 6, 32, 16
    ntok, nb = ws * ws, B * (H // ws) * (W // ws)
    q = (torch.randn(nb, heads, ntok, hd_p, device=DEV) * 0.4).to(torch.bfloat16)
    k = torch.randn(nb, heads, ntok, hd_p, device=DEV).to(torch.bfloat16)
    vt = torch.randn(nb, heads, hd_p, ntok, device=DEV).to(torch.bfloat16)
    table = torch.randn((2 * ws - 1) ** 2, heads, device=DEV)
    bias = packing.gather_bias(table, rpi_sa(ws), ntok, ntok)
    bias_frag = packing.bias_fragments(bias)
    tiles = packing.bias_distinct_tiles(bias)
    assert tiles is not None and tiles.numel() == heads * 31 * 256
    assert packing.bias_distinct_tiles(torch.randn(heads, ntok, ntok, device=DEV)) is None
    ym = L.Y_ROLL if y_mode == "roll" else L.Y_STRIP

    def run(lds):
        o = torch.full((nb * ntok, heads * hd_p), float("nan"), device=DEV).to(torch.bfloat16)
        ops.window_attention(q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), bias=bias.data_ptr(), out=o.data_ptr(), n_bwin=nb, heads=heads, hd_p=hd_p, ntok=ntok,
                             H=H, W=W, ws=ws, shift=shift, dtype=L.SR_BF16, y_mode=ym, bias_frag=bias_frag.data_ptr(), qkv_frag=frag,
                             bias_tiles=tiles.data_ptr() if lds else None)
        torch.cuda.synchronize()
        return o.float()

    new, old = run(True), run(False)
    assert not torch.isnan(new).any()
    scale = float(old.abs().max())
    assert float((new - old).abs().max()) <= 8e-3 * scale, float((new - old).abs().max()) / scale  # one bf16 ulp of o
    assert float((new - old).pow(2).mean().sqrt()) <= 5e-4 * scale
    if frag == 0:  # torch reference on the same rounded operands, shift mask from the reference's labels (common.py:250-274)
        from oracle import functional as OF

        s = torch.einsum("bhqd,bhkd->bhqk", q.float(), k.float()) + bias[None]
        if shift:
            mask = OF.calculate_mask(H, W, ws, shift).to(DEV)  # [nW, ntok, ntok]
            if y_mode == "strip":
                mask = None
            if mask is not None:
                s = (s.reshape(B, -1, heads, ntok, ntok) + mask[None, :, None]).reshape(nb, heads, ntok, ntok)
        if not (shift and y_mode == "strip"):
            ref = torch.einsum("bhqk,bhdk->bqhd", torch.softmax(s, -1), vt.float()).reshape(nb * ntok, heads * hd_p)
            assert float((new - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_overlapping_cross_attention_lds_form_against_the_flash_form_and_torch():
    """sr_oca_attention with SrOcaAttn.bias_rel (ABI v8: K / V^T of the 24 x 24 neighbourhood and the head's relative-position table staged in LDS once per
    (window, head), csrc/sr_oca_lds.hip) against the flash form on the same zero-bordered operands and against torch on the unfolded keys (hat.py:239-283); the bias is
    table[rpi_oca] with the reference's wrapping negative indices; a bias without that structure is refused by the packer."""
    from studiosr_amd.models.hat import rpi_oca

    torch.manual_seed(47)
    B, H, W, heads, hd_p, ws, pad, e = 2, 32, 48, 6, 32, 16, 4, 4
    wse, ntok, nk = ws + 2 * pad, ws * ws, (ws + 2 * pad) ** 2
    nb = B * (H // ws) * (W // ws)
    q = (torch.randn(nb, heads, ntok, hd_p, device=DEV) * 0.4).to(torch.bfloat16)
    kimg = torch.zeros(B, H + 2 * e, W + 2 * e, heads, hd_p, device=DEV)
    kimg[:, e:-e, e:-e] = torch.randn(B, H, W, heads, hd_p, device=DEV)
    vpl = torch.zeros(B, heads, hd_p, H + 2 * e, W + 2 * e, device=DEV)
    vpl[..., e:-e, e:-e] = torch.randn(B, heads, hd_p, H, W, device=DEV)
    kimg, vpl = kimg.to(torch.bfloat16), vpl.to(torch.bfloat16)
    table = torch.randn((ws + wse - 1) ** 2, heads, device=DEV)
    bias = packing.gather_bias(table, rpi_oca(ws, 0.5), ntok, nk)  # [heads, 256, 576]
    rel = packing.oca_bias_rel(bias)
    assert rel is not None and rel.shape == (heads, 1521)
    assert packing.oca_bias_rel(torch.randn(heads, ntok, nk, device=DEV)) is None
    bias_frag = packing.bias_fragments(bias)  # 576 keys: a multiple of 64, no padding columns

    def run(lds):
        o = torch.full((nb * ntok, heads * hd_p), float("nan"), device=DEV).to(torch.bfloat16)
        ops.oca_attention(q=q.data_ptr(), k=kimg.data_ptr(), vt=vpl.data_ptr(), bias=bias.data_ptr(), out=o.data_ptr(), B=B, H=H, W=W, heads=heads, hd_p=hd_p, ws=ws,
                          pad=pad, border=e, nk_pad=nk, dtype=L.SR_BF16, bias_frag=bias_frag.data_ptr(), nk_frag=nk, bias_rel=rel.data_ptr() if lds else None)
        torch.cuda.synchronize()
        return o.float()

    new, old = run(True), run(False)
    assert not torch.isnan(new).any()
    scale = float(old.abs().max())
    assert float((new - old).abs().max()) <= 8e-3 * scale and float((new - old).pow(2).mean().sqrt()) <= 5e-4 * scale
    # torch: unfold the neighbourhoods (nn.Unfold(kernel 24, stride 16, padding 4) of the un-bordered image = windows of the bordered one)
    kf = kimg.float()[:, e - pad : e - pad + H + 2 * pad, e - pad : e - pad + W + 2 * pad]  # [B, H + 8, W + 8, heads, 32]
    vf = vpl.float()[..., e - pad : e - pad + H + 2 * pad, e - pad : e - pad + W + 2 * pad]
    outs = []
    for b in range(B):
        for wy in range(H // ws):
            for wx in range(W // ws):
                kw = kf[b, wy * ws : wy * ws + wse, wx * ws : wx * ws + wse].reshape(nk, heads, hd_p).permute(1, 0, 2)      # [heads, 576, 32]
                vw = vf[b, :, :, wy * ws : wy * ws + wse, wx * ws : wx * ws + wse].reshape(heads, hd_p, nk).transpose(1, 2)  # [heads, 576, 32]
                qq = q[(b * (H // ws) + wy) * (W // ws) + wx].float()
                outs.append((torch.softmax(qq @ kw.transpose(1, 2) + bias, -1) @ vw).permute(1, 0, 2).reshape(ntok, heads * hd_p))
    ref = torch.cat(outs)
    assert float((new - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_overlapping_cross_attention_split_operand_form_against_exact_fp32():
    """ABI v11: sr_oca_attention with dtype SR_BF16X3 (precision "fp32x3": fp32 q / zero-bordered k / v^T planes / out, split-operand bf16 MFMAs) against the exact-fp32
    instantiation of the same flash kernel (hat.py:239-283; 576 keys of the 24 x 24 neighbourhood, wrapping negative bias indices)."""
    from studiosr_amd.models.hat import rpi_oca
    from studiosr_amd.runtime import x3_mode

    torch.manual_seed(48)
    B, H, W, heads, hd_p, ws, pad, e = 2, 32, 48, 6, 32, 16, 4, 4
    wse, ntok, nk = ws + 2 * pad, ws * ws, (ws + 2 * pad) ** 2
    nb = B * (H // ws) * (W // ws)
    q = torch.randn(nb, heads, ntok, hd_p, device=DEV) * 0.4
    kimg = torch.zeros(B, H + 2 * e, W + 2 * e, heads, hd_p, device=DEV)
    kimg[:, e:-e, e:-e] = torch.randn(B, H, W, heads, hd_p, device=DEV)
    vpl = torch.zeros(B, heads, hd_p, H + 2 * e, W + 2 * e, device=DEV)
    vpl[..., e:-e, e:-e] = torch.randn(B, heads, hd_p, H, W, device=DEV)
    table = torch.randn((ws + wse - 1) ** 2, heads, device=DEV)
    bias = packing.gather_bias(table, rpi_oca(ws, 0.5), ntok, nk)
    bias_frag = packing.bias_fragments(bias)

    def run(x3):
        o = torch.full((nb * ntok, heads * hd_p), float("nan"), device=DEV)
        with x3_mode(x3):
            ops.oca_attention(q=q.data_ptr(), k=kimg.data_ptr(), vt=vpl.data_ptr(), bias=bias.data_ptr(), out=o.data_ptr(), B=B, H=H, W=W, heads=heads, hd_p=hd_p, ws=ws,
                              pad=pad, border=e, nk_pad=nk, dtype=L.SR_F32, bias_frag=bias_frag.data_ptr(), nk_frag=nk, bias_rel=None)
        torch.cuda.synchronize()
        return o

    new, old = run(True), run(False)
    assert not torch.isnan(new).any()
    assert float((new - old).abs().max()) <= 2e-5 * float(old.abs().max()), float((new - old).abs().max()) / float(old.abs().max())


def test_hat_graph_replay_with_half_batches_equals_the_eager_forward():
    """Inside a HIP-graph capture a HAT batch of >= 8 runs as two half batches on two streams (models/hat.py forward, SR_HAT_PARTS): same kernels on the same
    per-image data, so the replayed output is bit-identical to the eager one-sequence forward, image by image."""
    from studiosr_amd.runtime import GraphedForward

    torch.manual_seed(9)
    m = _randomised(S.HAT(scale=2, depths=[2], num_heads=[6]), seed=9).to(DEV).eval().set_precision("bf16")
    x = torch.rand(8, 3, 32, 32, device=DEV)
    with torch.no_grad():
        eager = m(x).clone()
        g = GraphedForward(lambda t: m(t), x)
        replayed = g(x).clone()
        torch.cuda.synchronize()
    assert torch.equal(replayed, eager)


@pytest.mark.parametrize("shift", [0, 8])
def test_window_attention_with_its_own_qkv_projection_equals_qkv_launch_then_attention(shift):
    """sr_window_attention with SrWindowAttn.x / wqkv (ABI v8, csrc/sr_wattn_qkv_body.h: every (window, head) workgroup normalises the window's rows and projects its
    own q, k, v from sr_swin_qkv's weight stream, keeps them in LDS and runs the attention) against sr_swin_qkv followed by the LDS-form attention on the same packed
    block (hat.py:164-176, 85-110): same products and the same bf16 rounding of q / k / v; LayerNorm statistics are summed in another order."""
    torch.manual_seed(53)
    m = _randomised(S.HAT(scale=2, depths=[2], num_heads=[6]), seed=53).to(DEV).eval().set_precision("bf16")
    lp = m._get_packed(torch.bfloat16)["layers"][0]
    geo = lp["geo"]
    bp = lp["blocks"][0]
    B, H, W, Cp = 2, 32, 48, geo.Cp
    nb = B * (H // 16) * (W // 16)
    t = torch.randn(B, H, W, Cp, device=DEV)
    t[..., geo.C:] = 0
    assert "bias_tiles" in bp and bp["qkv_dtype"] == L.SR_BF16
    q, k, vt = (torch.empty(nb, 6, 256, 32, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    ops.swin_qkv(x=t.data_ptr(), q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), wstream=bp["qkv_stream"].data_ptr(), B=B, H=H, W=W, C=geo.C, Cp=Cp, ldx=Cp, heads=6,
                 hd_p=32, ws=16, shift=shift, eps=1e-5, y_mode=L.Y_ROLL, compute_dtype=L.SR_BF16, frag_order=1)
    common = dict(bias=bp["bias"].data_ptr(), n_bwin=nb, heads=6, hd_p=32, ntok=256, H=H, W=W, ws=16, shift=shift, dtype=L.SR_BF16, y_mode=L.Y_ROLL,
                  bias_frag=bp["bias_frag"].data_ptr(), bias_tiles=bp["bias_tiles"].data_ptr())
    o_ref = torch.full((nb * 256, 192), float("nan"), device=DEV).to(torch.bfloat16)
    ops.window_attention(q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), out=o_ref.data_ptr(), qkv_frag=1, **common)
    o_new = torch.full((nb * 256, 192), float("nan"), device=DEV).to(torch.bfloat16)
    ops.window_attention(out=o_new.data_ptr(), x=t.data_ptr(), wqkv=bp["qkv_stream"].data_ptr(), ldx=Cp, C=geo.C, eps=1e-5, qkv_frag=0, **common)
    torch.cuda.synchronize()
    a, b = o_new.float(), o_ref.float()
    assert not torch.isnan(a).any()
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 1.6e-2 * scale and float((a - b).pow(2).mean().sqrt()) <= 1e-3 * scale, (float((a - b).abs().max()) / scale)


def test_hat_forward_with_the_one_launch_mid_stage_matches_the_two_stream_form(monkeypatch):
    """A default-width HAT forward (shifted and unshifted HABs, OCAB) with sr_hab_mid (default) and with the two-stream attention || CAB launches."""
    torch.manual_seed(5)
    m = _randomised(S.HAT(scale=2, depths=[2, 2], num_heads=[6, 6]), seed=7).to(DEV).eval().set_precision("bf16")
    x = torch.rand(2, 3, 48, 32, device=DEV)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SR_HAB_MID", flag)
        with torch.no_grad():
            outs.append(m(x).clone())
    torch.cuda.synchronize()
    new, old = outs
    assert not torch.isnan(new).any()
    assert float((new - old).abs().max()) <= 5e-3 * max(1.0, float(old.abs().max()))


def test_hat_last_block_tail_with_the_ocab_qkv_stage_equals_the_separate_qkv_launch(monkeypatch):
    """The last HAB of a group: sr_swin_tail goes on with the OCAB's LayerNorm1 + QKV (q in window order, k / v^T in the zero-bordered layouts; SrSwinTail.oca_pad2,
    default) against the sr_swin_qkv launch with oca_pad (SR_TAIL_OCA=0): the same weights and the same arithmetic on the same rows -> the same bits."""
    torch.manual_seed(6)
    m = _randomised(S.HAT(scale=2, depths=[2, 2], num_heads=[6, 6]), seed=11).to(DEV).eval().set_precision("bf16")
    x = torch.rand(2, 3, 48, 32, device=DEV)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SR_TAIL_OCA", flag)
        with torch.no_grad():
            outs.append(m(x).clone())
    torch.cuda.synchronize()
    new, old = outs
    assert not torch.isnan(new).any()
    assert torch.equal(new, old), float((new - old).abs().max())


def test_hat_forward_with_32_token_tail_workgroups_equals_the_64_token_form(monkeypatch):
    """sr_swin_tail with 32 tokens per workgroup (SrSwinTail.wg_tokens; what small batches run) against 64: the same rows through the same arithmetic (gated second
    residual, LayerNorm side output, fused next-block and OCAB QKV included) -> the same bits."""
    torch.manual_seed(8)
    m = _randomised(S.HAT(scale=2, depths=[3], num_heads=[6]), seed=13).to(DEV).eval().set_precision("bf16")
    x = torch.rand(2, 3, 48, 32, device=DEV)
    outs = []
    for upto in ("1000000", "0"):
        monkeypatch.setenv("SR_TAIL_WG32_UPTO", upto)
        with torch.no_grad():
            outs.append(m(x).clone())
    torch.cuda.synchronize()
    new, old = outs
    assert not torch.isnan(new).any()
    assert torch.equal(new, old), float((new - old).abs().max())


def test_hat_forward_with_8_row_cab_tiles_matches_the_6_row_form(monkeypatch):
    """sr_hab_mid's CAB role with 14 x 8-output tiles (SrCab.tile_rows = 8: what launches from 4 x 64 x 64 pixels on take) against 14 x 6: the same convolutions per
    pixel; the pool partials are sums over another tile partition, so the gate (and with it the output) agrees to fp32 rounding of a 180-term mean, not bit for bit.
    Image heights that are and are not multiples of either tile height."""
    torch.manual_seed(9)
    m = _randomised(S.HAT(scale=2, depths=[2, 2], num_heads=[6, 6]), seed=17).to(DEV).eval().set_precision("bf16")
    for shape in ((2, 3, 48, 32), (1, 3, 64, 64)):
        x = torch.rand(*shape, device=DEV)
        outs = []
        for frm in ("0", "1000000000"):
            monkeypatch.setenv("SR_CAB_ROWS8_FROM", frm)
            with torch.no_grad():
                outs.append(m(x).clone())
        torch.cuda.synchronize()
        new, old = outs
        assert not torch.isnan(new).any()
        assert float((new - old).abs().max()) <= 2e-3 * max(1.0, float(old.abs().max())), float((new - old).abs().max())


def test_gated_second_residual_of_the_projection_gemm_equals_channel_attention():
    """HAT's combine x = shortcut + attn + conv_scale * CA(cab) (hat.py:192): sr_channel_gate + sr_gemm's gated second residual against the
    two-pass form (projection GEMM with the shortcut, then sr_channel_attention over the stream)."""
    from studiosr_amd.models.rcan import pack_ca, run_channel_attention

    torch.manual_seed(21)
    B, H, W, C, Cp, Cr = 2, 16, 16, 180, 192, 6
    M = B * H * W
    a = torch.randn(M, Cp, device=DEV).to(torch.bfloat16)
    wl = torch.randn(C, C, device=DEV) * 0.05
    wp, bp = packing.pack_linear(wl, torch.randn(C, device=DEV) * 0.1, packing.identity_idx(C, Cp), packing.identity_idx(C, Cp), torch.bfloat16)
    shortcut = torch.randn(B, H, W, Cp, device=DEV)
    shortcut[..., C:] = 0
    y = torch.randn(B, H, W, Cp, device=DEV).to(torch.bfloat16)
    y[..., C:] = 0
    n_tiles = 5
    pool = torch.randn(B, n_tiles, Cp, device=DEV)
    ca = pack_ca(torch.randn(Cr, C, 1, 1, device=DEV) * 0.3, torch.randn(Cr, device=DEV) * 0.1, torch.randn(C, Cr, 1, 1, device=DEV) * 0.3, torch.randn(C, device=DEV) * 0.1)
    gkw = dict(A=a.data_ptr(), Wp=wp.data_ptr(), bias=bp.data_ptr(), M=M, K=Cp, N=Cp, lda=Cp, ldo=Cp, ldskip=Cp, a_dtype=L.SR_BF16, out_dtype=L.SR_F32,
               compute_dtype=L.SR_BF16, act=L.ACT_NONE, out_scale=1.0, a_map=L.MAP_IDENTITY, o_map=L.MAP_IDENTITY, epi=L.EPI_STD)
    want = torch.empty(B, H, W, Cp, device=DEV)
    ops.gemm(out=want.data_ptr(), skip=shortcut.data_ptr(), **gkw)
    run_channel_attention(ca, y, pool, n_tiles, C, want, skip=want, y_scale=0.01)
    gate = torch.full((B, Cp), float("nan"), device=DEV)
    w1, b1, w2, b2 = ca
    ops.channel_gate(gate, pool_partial=pool.data_ptr(), w1=w1.data_ptr(), b1=b1.data_ptr(), w2=w2.data_ptr(), b2=b2.data_ptr(), B=B, H=H, W=W, C=C, C_p=Cp,
                     Cr=Cr, n_tiles=n_tiles, y_scale=0.01)
    assert float(gate[:, C:].abs().max()) == 0.0 and float(gate[:, :C].min()) > 0.0 and float(gate[:, :C].max()) < 0.01
    got = torch.full_like(want, float("nan"))
    ops.gemm(out=got.data_ptr(), skip=shortcut.data_ptr(), skip2=y.data_ptr(), skip2_gate=gate.data_ptr(), skip2_dtype=L.SR_BF16, ldskip2=Cp, gate_rows=H * W,
             ld_gate=Cp, **gkw)
    assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))  # same products, one rounding order apart


def test_errors_are_loud():
    m = S.EDSR(scale=2, n_feats=32, n_resblocks=1).to(DEV).eval()
    with pytest.raises(RuntimeError):
        m(torch.rand(1, 4, 8, 8, device=DEV))  # wrong channel count
    with pytest.raises(L.HipLibraryError):
        m(torch.rand(1, 3, 8, 8))  # CPU tensor: no fallback
    with pytest.raises(L.HipLibraryError) as e:
        ops.gemm(A=1, Wp=1, out=1, M=16, K=30, N=64, lda=32)  # K not a multiple of 32: rejected before any launch
    assert "sr_gemm" in str(e.value)
    hat = S.HAT(embed_dim=60, depths=[1], num_heads=[6], window_size=8).to(DEV).eval()
    with pytest.raises(RuntimeError):
        hat(torch.rand(1, 3, 3, 3, device=DEV))  # reflect pad larger than the image, as F.pad raises in the reference


# ----------------------------------------------------------------------------- device-side weight packing (C ABI sr_pack_*)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_device_weight_packing_is_bit_identical_to_the_host_packing(dt):
    """sr_pack_matrix / sr_pack_conv3x3 / sr_pack_vector / sr_pack_bias_fragments against studiosr_amd/packing.py: every element equal
    (the host path writes its zero pads as value * 0, i.e. -0.0 under negative values; the kernels write +0.0 -- equal as numbers)."""
    from studiosr_amd.models import hat as HATM

    torch.manual_seed(0)
    i32 = lambda t: t.to(torch.int32).to(DEV).contiguous()  # noqa: E731
    C, Cp, heads, hd_p = 180, 192, 6, 32
    hd = C // heads
    # qkv Linear with the LayerNorm affine folded in, q rows scaled by hd^-0.5 (packing.pack_qkv + fold_layernorm)
    w, b = (torch.randn(3 * C, C) * 0.1).to(DEV), torch.randn(3 * C).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(C)).to(DEV), (0.1 * torch.randn(C)).to(DEV)
    fw, fb = packing.fold_layernorm(w, b, gamma, beta)
    want_w, want_b = packing.pack_qkv(fw, fb, C, Cp, heads, hd_p, dt)
    h_idx = packing.head_idx(heads, hd, hd_p)
    rows = torch.cat([torch.where(h_idx >= 0, h_idx + p * C, h_idx) for p in range(3)])
    scale = torch.ones(rows.numel())
    scale[: heads * hd_p] = hd ** -0.5
    got_w = ops.pack_matrix(w, rows.numel(), Cp, dt, row_idx=i32(rows), col_idx=i32(packing.identity_idx(C, Cp)), row_scale=scale.to(DEV), col_scale=gamma)
    same = lambda a, b_: a.dtype == b_.dtype and a.shape == b_.shape and torch.equal(a.float(), b_.float())  # noqa: E731
    assert same(got_w, want_w)
    got_b = ops.pack_vector(fb, rows.numel(), idx=i32(rows), scale=scale.to(DEV))
    assert torch.equal(got_b, want_b)
    # proj: identity rows, head-padded columns; identity maps given as NULL
    pw = (torch.randn(C, C) * 0.1).to(DEV)
    want, _ = packing.pack_linear(pw, None, packing.identity_idx(C, Cp), packing.head_idx(heads, hd, hd_p), dt)
    got = ops.pack_matrix(pw, Cp, heads * hd_p, dt, col_idx=i32(packing.head_idx(heads, hd, hd_p)))
    assert same(got, want)
    # 3x3 convs: identity rows and the PixelShuffle row permutation
    cw, cb = (torch.randn(180, 60, 3, 3) * 0.1).to(DEV), torch.randn(180).to(DEV)
    want, wb = packing.pack_conv3x3(cw, cb, 64, packing.identity_idx(180, 192), dt)
    assert same(ops.pack_conv3x3(cw, 64, 192, dt), want)
    assert torch.equal(ops.pack_vector(cb, 192), wb)
    uw = (torch.randn(256, 64, 3, 3) * 0.1).to(DEV)
    ps_rows = packing.pixel_shuffle_rows(64, 64, 2)
    want, _ = packing.pack_conv3x3(uw, None, 64, ps_rows, dt)
    assert same(ops.pack_conv3x3(uw, 64, ps_rows.numel(), dt, row_idx=i32(ps_rows)), want)
    if dt == torch.float32:  # relative-position bias tables (fp32 only): SwinIR 64x64 and HAT's OCA 256x576 with wrapped negative indices
        for ws, rpi in ((8, HATM.rpi_sa(8)), (16, HATM.rpi_oca(16, 0.5))):
            nq, nk = rpi.shape
            table = torch.randn(int(rpi.max() - min(int(rpi.min()), 0)) + 1 if rpi.min() >= 0 else (ws + ws + ws // 2 - 1) ** 2, heads).to(DEV)
            want = packing.bias_fragments(packing.gather_bias(table, rpi.to(DEV), nq, nk))
            got = ops.pack_bias_fragments(table, rpi.to(DEV).contiguous(), nq, nk)
            assert torch.equal(got, want), ws


@pytest.mark.parametrize("shift", [0, 4])
def test_swin_block_stream_kernel_against_the_reference_block(shift):
    """sr_swin_block (ABI v5: one launch, one packed weight stream, biases on constant-one channels, exp2 softmax) on the reference's own
    SwinTransformerBlock vectors (fixture f07: dim 180, 6 heads, 24 x 24 tokens, shift 0 and 4 -> masked windows), next to the un-fused
    launch sequence on the same input; the bf16 path keeps the stream, the LayerNorm statistics and the softmax in fp32."""
    import os

    from studiosr_amd.models import swinir as SW

    g = load_golden("f07_swin_block")
    sd = golden_sd(g, f"sd{shift}/")
    blk = SW.SwinTransformerBlock(180, 6, 8, shift, 2.0)
    blk.load_state_dict(sd)
    blk = blk.to(DEV)
    geo = SW.SwinGeometry(180, 6, 8, 360)
    cdt = torch.bfloat16
    p = dict(shift=shift)
    p.update(SW.pack_attention(blk.attn, geo, cdt, norm=blk.norm1))
    p.update(SW.pack_mlp(blk.mlp, geo, cdt, norm=blk.norm2))
    p.update(SW.pack_block_stream(blk, geo, cdt))
    assert "stream" in p and p["stream"].numel() == 48 * 12 * 64 * 8
    x = torch.from_numpy(g["x"]).to(DEV)
    xin = torch.zeros(1, 24, 24, geo.Cp, device=DEV)
    xin[..., :180] = x
    want = torch.from_numpy(g[f"y_shift{shift}"])
    ws_ = S.runtime.Workspace(torch.device(DEV))
    outs = {}
    out = torch.full_like(xin, float("nan"))
    SW.run_swin_block(p, geo, xin, out, ws_, cdt, shift)  # the one-launch stream kernel
    torch.cuda.synchronize()
    outs["stream"] = out.cpu()
    out = torch.full_like(xin, float("nan"))
    SW.run_window_msa(p, None, geo, xin, out, xin, ws_, cdt, shift)  # the un-fused launch sequence (QKV GEMM, window attention, projection GEMM, MLP)
    SW.run_mlp(p, None, geo, out, ws_, cdt)  # (bf16: the LayerNorm affines are folded into the packed weights)
    torch.cuda.synchronize()
    outs["unfused"] = out.cpu()
    delta = want - torch.from_numpy(g["x"])  # what the block adds to its input (the fixture's weights are large: |delta| up to 26, rms 5.6)
    rng, rms = float(delta.abs().max()), float(delta.pow(2).mean().sqrt())
    for kern, out in outs.items():
        assert bool((out[..., 180:] == 0).all()), kern  # pad channels of the stream stay exactly zero
        d = out[..., :180] - want
        # bf16 operands: 1 % rms of the block's contribution (both kernels measure 0.97-1.0 %), worst element 2.5 % of its range
        assert float(d.pow(2).mean().sqrt()) <= 1.5e-2 * rms and float(d.abs().max()) <= 2.5e-2 * rng, (kern, float(d.pow(2).mean().sqrt()), float(d.abs().max()), rms, rng)
    # the two forms differ only in rounding order (bias path, exp2)
    assert float((outs["stream"] - outs["unfused"]).pow(2).mean().sqrt()) <= 1.5e-2 * rms


@pytest.mark.parametrize("prec", ["bf16", "fp32x3"])
def test_swin_block_persistent_workgroups_equal_one_workgroup_per_window(prec):
    """SrSwinBlock.max_workgroups (ABI v10): a grid smaller than the window count makes every workgroup walk windows b, b + grid, ... with the next window's
    rows fetched under the current result stores.  A window's arithmetic does not depend on who computes it: the output must equal the one-workgroup-per-window
    launch bit for bit, for grids that divide the window count, that do not, and for a single workgroup; shifted block (mask path), in place and out of place."""
    import os

    from studiosr_amd.models import swinir as SW
    from studiosr_amd.runtime import x3_mode

    torch.manual_seed(0)
    m = S.SwinIR(scale=2, depths=[2], num_heads=[6]).eval()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.ndim == 1:
                p_.add_(torch.randn_like(p_) * 0.1)
    m = m.to(DEV).set_precision(prec)
    cdt = torch.bfloat16 if prec == "bf16" else torch.float32
    with x3_mode(prec == "fp32x3"):
        lp = m._get_packed(cdt)["layers"][0]
        geo, ws_ = lp["geo"], S.runtime.Workspace(torch.device(DEV))
        x = torch.randn(3, 40, 24, geo.Cp, device=DEV)  # 3 x 5 x 3 = 45 windows
        x[..., geo.C:] = 0
        for bp in lp["blocks"]:
            outs = {}
            for wgs in (-1, -2, 45, 16, 7, 1):
                S.runtime.reset_knobs()
                os.environ["SR_BLOCK_WGS"] = str(wgs)
                try:
                    o = torch.full_like(x, float("nan"))
                    SW.run_swin_block(bp, geo, x, o, ws_, cdt, bp["shift"])
                    xi = x.clone()
                    SW.run_swin_block(bp, geo, xi, xi, ws_, cdt, bp["shift"])  # in place
                    torch.cuda.synchronize()
                finally:
                    os.environ.pop("SR_BLOCK_WGS", None)
                    S.runtime.reset_knobs()
                assert torch.equal(o, xi), (wgs, "in place")
                outs[wgs] = o
            assert bool(torch.isfinite(outs[-1]).all())
            for wgs, o in outs.items():
                assert torch.equal(o, outs[-1]), wgs


def test_swinfir_bf16_keeps_the_fft_in_fp32_at_dft_sized_images():
    """SwinFIR precision='bf16' on an image large enough (H >= 96, W >= 190) for sr_bgemm's bf16 path: the DFT-as-GEMM rFFT / irFFT of
    the SFB blocks must stay exact fp32 (the reference never runs torch.fft under bf16 autocast); checked against the oracle."""
    torch.manual_seed(0)
    m = S.SwinFIR(scale=2, embed_dim=60, depths=[2], num_heads=[6]).eval()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.ndim == 1:
                p_.add_(torch.randn_like(p_) * 0.05)
    x = torch.rand(1, 3, 120, 248, generator=torch.Generator().manual_seed(2))  # eval pad -> 128 x 256
    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = OM.swinir_forward(sd, x, m.get_model_config())
        y = m.to(DEV).set_precision("bf16")(x.to(DEV)).cpu()
    mse = float(((y - ref) ** 2).mean())
    psnr = 10 * np.log10(max(float(ref.abs().max()), 1.0) ** 2 / max(mse, 1e-20))
    assert float((y - ref).abs().max()) <= BF16_TOL * max(1.0, float(ref.abs().max())) and psnr >= BF16_PSNR_DB, (float((y - ref).abs().max()), psnr)


def test_strips_full_size_2048_local_8_equal_the_unsharded_forward():
    """BASELINE config 4 at its real size: default SwinIR x4 on ONE 2048 x 2048 LR image in 8 window-aligned row strips with per-layer halo
    exchange (LocalStripComm: all strips in this process, the orchestration a DistStripComm rank runs; halo exchanges on the side
    stream) must equal the unsharded forward bit for bit (bf16 path, the one-launch block kernel with y_mode strips)."""
    from studiosr_amd.strips import LocalStripComm

    torch.manual_seed(0)
    m = S.SwinIR(scale=4).to(DEV).eval().set_precision("bf16")
    x = torch.rand(1, 3, 2048, 2048, device=DEV)
    with torch.no_grad():
        ref = m(x)
        out = m.forward_strips(x, LocalStripComm(8))
    assert out.shape == ref.shape == (1, 3, 8192, 8192)
    assert torch.equal(out, ref), float((out - ref).abs().max())
    del out, ref
    torch.cuda.empty_cache()


def test_inference_under_auto_runs_the_split_operand_path():
    """Model.inference with precision 'auto' (outside autocast) = the reference-precision fast path 'fp32x3' -- for SwinIR the fused block
    kernel in its split-operand instantiation -- and stays within one LSB of the exact-fp32 mode on a handful of pixels at most."""
    torch.manual_seed(0)
    m = S.SwinIR(scale=4, depths=[2, 2], num_heads=[6, 6]).to(DEV).eval()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.ndim == 1:
                p_.add_(torch.randn_like(p_) * 0.05)
    img = np.random.default_rng(0).integers(0, 256, size=(40, 56, 3), dtype=np.uint8)
    y_auto = m.set_precision("auto").inference(img)
    assert m.precision == "auto"  # restored
    y_x3 = m.set_precision("fp32x3").inference(img)
    y_32 = m.set_precision("fp32").inference(img)
    assert np.array_equal(y_auto, y_x3)
    d = np.abs(y_auto.astype(int) - y_32.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (d.max(), (d > 0).mean())
    # and the float forward of the split-operand block kernel against the exact path
    x = torch.rand(2, 3, 40, 56, device=DEV)
    with torch.no_grad():
        e = float((m.set_precision("fp32x3")(x) - m.set_precision("fp32")(x)).abs().max())
    assert e <= 2e-5, e
