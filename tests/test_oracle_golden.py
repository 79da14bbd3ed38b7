"""The oracle (oracle/) pinned against vectors produced by the reference itself
(tests/golden/generate.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden_cfg, golden_sd, load_golden
from oracle import functional as OF
from oracle import metrics as OM
from oracle import models as OMD

TOL = dict(rtol=1e-5, atol=2e-6)  # same fp32 torch ops, different association in a few places


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("r", [2, 3, 4])
def test_pixel_shuffle_bit_exact(r):
    g = load_golden("f01_pixel_shuffle")
    out = OF.pixel_shuffle(t(g[f"in_r{r}"]), r)
    assert out.dtype == torch.int32
    assert torch.equal(out, t(g[f"out_r{r}"]))


def test_window_maps_bit_exact():
    g = load_golden("f02_window")
    x = t(g["x"])
    assert torch.equal(OF.window_partition(x, 8), t(g["part_shift0"]))
    src = OF.window_token_source(16, 24, 8, 4)  # gather map: roll + partition in one step
    flat = x.reshape(2, 16 * 24, 3)
    part = flat[:, torch.from_numpy(src)].reshape(-1, 8, 8, 3)  # [B, nW, N, C] -> [B*nW, 8, 8, C]
    assert torch.equal(part, t(g["part_shift4"]))
    back = torch.empty_like(flat)
    back[:, torch.from_numpy(src).reshape(-1)] = part.reshape(2, -1, 3)  # scatter = reverse + roll back
    assert torch.equal(back.reshape(2, 16, 24, 3), t(g["back"]))
    assert torch.equal(OF.window_reverse(OF.window_partition(x, 8), 8, 16, 24), x)


def test_masks_and_rpi_bit_exact():
    g = load_golden("f03_mask_rpi")
    for k, v in g.items():
        if k.startswith("mask_"):
            h, w, ws, sh = map(int, k.split("_")[1:])
            assert torch.equal(OF.calculate_mask(h, w, ws, sh), t(v)), k
    assert torch.equal(OF.relative_position_index(8), t(g["rpi_swin_8"]))
    assert int(OF.relative_position_index(8).sum()) == 458752
    assert torch.equal(OF.relative_position_index(16), t(g["rpi_hat_sa_16"]))
    assert torch.equal(OF.relative_position_index(8), t(g["rpi_hat_sa_8"]))
    oca = OF.relative_position_index_oca(16, 0.5)
    assert torch.equal(oca, t(g["rpi_hat_oca_16"]))
    assert int(oca.min()) == -880 and int(oca.max()) == 640
    assert torch.equal(OF.relative_position_index_oca(8, 0.5), t(g["rpi_hat_oca_8"]))


def test_padding_bit_exact():
    g = load_golden("f05_pad")
    for k, v in g.items():
        mode, h, w = k.split("_")
        x = torch.arange(2 * int(h) * int(w), dtype=torch.float32).reshape(1, 2, int(h), int(w))
        out = OF.pad_eval(x, 8) if mode == "eval" else OF.pad_reflect(x, 8)
        assert torch.equal(out, t(v)), k


def test_window_attention():
    g = load_golden("f06_window_attention")
    sd = {"a." + k: v for k, v in golden_sd(g).items()}
    x, rpi = t(g["x"]), sd["a.relative_position_index"]
    torch.testing.assert_close(OF.window_attention(sd, "a", x, rpi, 6, None), t(g["y_nomask"]), **TOL)
    torch.testing.assert_close(OF.window_attention(sd, "a", x, rpi, 6, t(g["mask"])), t(g["y_mask"]), **TOL)


@pytest.mark.parametrize("shift", [0, 4])
def test_swin_block(shift):
    g = load_golden("f07_swin_block")
    sd = {"b." + k: v for k, v in golden_sd(g, f"sd{shift}/").items()}
    y = OMD.swin_block(sd, "b", t(g["x"]), 8, shift, 6)
    torch.testing.assert_close(y, t(g[f"y_shift{shift}"]), **TOL)


def test_mlp():
    g = load_golden("f08_mlp")
    sd = {"m." + k: v for k, v in golden_sd(g).items()}
    torch.testing.assert_close(OF.mlp(sd, "m", t(g["x"])), t(g["y"]), **TOL)


def test_conv_blocks():
    g = load_golden("f09_conv_blocks")
    x = t(g["x"])
    rb = golden_sd(g, "resblock/")
    y = OF.conv(rb, "body.2", torch.relu(OF.conv(rb, "body.0", x))) * 0.1 + x
    torch.testing.assert_close(y, t(g["y_resblock"]), **TOL)
    ca = golden_sd(g, "ca/")
    y = OF.channel_attention(x, ca["conv_du.0.weight"], ca["conv_du.0.bias"], ca["conv_du.2.weight"], ca["conv_du.2.bias"])
    torch.testing.assert_close(y, t(g["y_ca"]), **TOL)
    rc = golden_sd(g, "rcab/")
    y = OF.conv(rc, "body.2", torch.relu(OF.conv(rc, "body.0", x)))
    y = OF.channel_attention(y, rc["body.3.conv_du.0.weight"], rc["body.3.conv_du.0.bias"], rc["body.3.conv_du.2.weight"], rc["body.3.conv_du.2.bias"]) + x
    torch.testing.assert_close(y, t(g["y_rcab"]), **TOL)


@pytest.mark.parametrize("tag,scale,direct", [("s2", 2, False), ("s3", 3, False), ("s4", 4, False), ("s4direct", 4, True)])
def test_upsampler(tag, scale, direct):
    g = load_golden("f10_upsampler")
    sd = {"u." + k: v for k, v in golden_sd(g, tag + "/").items()}
    torch.testing.assert_close(OF.upsampler(sd, "u", t(g["x"]), scale, direct), t(g["y_" + tag]), **TOL)


WHOLE = [
    ("f11_swinir_x2", "SwinIR"), ("f11_swinir_x3", "SwinIR"), ("f11_swinir_x4", "SwinIR"),
    ("f11_swinir_direct_x4", "SwinIR"), ("f11_swinir_c180_x4", "SwinIR"),
    ("f11_edsr_x2", "EDSR"), ("f11_edsr_x3", "EDSR"), ("f11_edsr_x4", "EDSR"), ("f11_edsr_r255_x2", "EDSR"),
    ("f11_rcan_x4", "RCAN"), ("f11_rcan_x3", "RCAN"),
    ("f11_hat_w8_x4", "HAT"), ("f11_hat_w16_x2", "HAT"),
]


@pytest.mark.parametrize("name,kind", WHOLE)
def test_whole_models(name, kind):
    g = load_golden(name)
    sd, cfg = golden_sd(g), golden_cfg(g)
    fwd = OMD.FORWARDS[kind]
    n = 0
    for k in g:
        if k.startswith("y_"):
            mode, b, h, w = k[2:].split("_")
            x = t(g[f"x_{b}_{h}_{w}"])
            y = fwd(sd, x, cfg, training=(mode == "train"))
            scale = max(1.0, float(np.abs(g[k]).max()))
            torch.testing.assert_close(y, t(g[k]), rtol=1e-4, atol=2e-5 * scale, msg=lambda m: f"{name}:{k}: {m}")
            n += 1
    assert n > 0


def test_inference_uint8_and_self_ensemble():
    g = load_golden("f12_inference")
    sd, cfg = golden_sd(g), golden_cfg(g)
    fwd = lambda x: OMD.edsr_forward(sd, x, cfg)  # noqa: E731
    y = OMD.inference(fwd, g["img"], cfg["img_range"])
    assert y.dtype == np.uint8 and y.shape == g["y"].shape
    assert np.abs(y.astype(int) - g["y"].astype(int)).max() <= 1 and (y != g["y"]).mean() < 1e-3
    ye = OMD.inference_with_self_ensemble(fwd, g["img"], cfg["img_range"])
    assert np.abs(ye.astype(int) - g["y_ens"].astype(int)).max() <= 1 and (ye != g["y_ens"]).mean() < 1e-3


def test_hat_blocks():
    g = load_golden("f13_hat_blocks")
    cab = {"c." + k: v for k, v in golden_sd(g, "cab/").items()}
    torch.testing.assert_close(OMD.hat_cab(cab, "c", t(g["x_cab"])), t(g["y_cab"]), **TOL)
    x = t(g["x_tok"]).reshape(2, 16, 24, 60)
    mask = OF.calculate_mask(16, 24, 8, 4)
    for sh in (0, 4):
        sd = {"h." + k: v for k, v in golden_sd(g, f"hab{sh}/").items()}
        y = OMD.hat_hab(sd, "h", x, 8, sh, 6, OF.relative_position_index(8), mask, 0.01)
        torch.testing.assert_close(y.reshape(2, -1, 60), t(g[f"y_hab{sh}"]), **TOL)
    sd = {"o." + k: v for k, v in golden_sd(g, "ocab/").items()}
    y = OMD.hat_ocab(sd, "o", x, 8, 6, OF.relative_position_index_oca(8, 0.5), 0.5)
    torch.testing.assert_close(y.reshape(2, -1, 60), t(g["y_ocab"]), **TOL)


def test_psnr_metric():
    g = load_golden("f14_psnr")
    a, b = g["a"], g["b"]
    assert abs(OM.compute_psnr(a, b) - float(g["psnr_rgb"])) < 1e-9
    assert abs(OM.compute_psnr(a, b, y_only=True, crop_border=4) - float(g["psnr_y_crop4"])) < 1e-9
    assert abs(OM.compute_psnr(a / 255.0, b / 255.0) - float(g["psnr_float"])) < 1e-9
    np.testing.assert_allclose(OM.to_y(a), g["y"], rtol=0, atol=1e-9)
    # the reference's own known-answer tests (tests/utils/test_compute_psnr.py:15-40)
    z, o = np.zeros((16, 16, 3), np.uint8), np.full((16, 16, 3), 255, np.uint8)
    assert OM.compute_psnr(z, o) == 0.0
    assert np.isinf(OM.compute_psnr(a, a.copy()))
    assert abs(OM.compute_psnr(a, b) - OM.compute_psnr(a / 255.0, b / 255.0)) < 1e-12


@pytest.mark.parametrize("tag,kind", [("swinir", "SwinIR"), ("swinir_direct", "SwinIR"), ("hat", "HAT"), ("edsr", "EDSR"), ("rcan", "RCAN"), ("swinfir", "SwinFIR"), ("han", "HAN")])
def test_oracle_autograd_matches_reference_gradients(tag, kind):
    """f15: gradients of one L1 training step produced by the reference (generate.py grads()).  torch autograd through the oracle's
    forward must reproduce them: this pins the oracle as a gradient reference for geometries without a fixture."""
    import torch.nn.functional as F

    from oracle import models as OM

    g = load_golden(f"f15_grads_{tag}")
    cfg, sd = golden_cfg(g), golden_sd(g)
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "mean" not in k else v) for k, v in sd.items()}
    out = OM.FORWARDS[kind](sdg, torch.from_numpy(g["x"]), cfg, training=True)
    assert float((out.detach() - torch.from_numpy(g["out"])).abs().max()) <= 1e-5 * max(1.0, float(np.abs(g["out"]).max()))
    if "y_eval" in g:  # SwinFIR / HAN: the eval-mode forward too (SwinFIR pads differently in eval)
        with torch.no_grad():
            ye = OM.FORWARDS[kind](sd, torch.from_numpy(g["x"]), cfg, training=False)
        assert float((ye - torch.from_numpy(g["y_eval"])).abs().max()) <= 1e-5 * max(1.0, float(np.abs(g["y_eval"]).max()))
    F.l1_loss(out, torch.from_numpy(g["target"])).backward()
    n = 0
    for k, v in g.items():
        if k.startswith("grad/"):
            got, ref = sdg[k[5:]].grad, torch.from_numpy(v)
            assert got is not None, k
            assert float((got - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-6), k
            n += 1
    assert n > 10
