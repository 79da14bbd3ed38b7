#!/usr/bin/env python3
"""Generate the golden vectors in this directory by RUNNING THE REFERENCE.

Run only in the build container (needs /root/reference):

    python tests/golden/generate.py

It imports veritross/studiosr from /root/reference (read-only, never copied),
registering in-memory stand-ins for the optional third-party modules the
reference imports at module scope but that are not installed here (timm.layers,
cv2, gdown, skimage) -- none of them takes part in eval-mode forward math.
Outputs are data only: seeded inputs, seeded weights (state_dict arrays) and the
reference's outputs, as compressed .npz.  The GPU box never sees the reference.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    timm, layers = types.ModuleType("timm"), types.ModuleType("timm.layers")

    class DropPath(nn.Module):  # identity in eval / p == 0, which is all the fixtures use
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.p, self.sbk = drop_prob, scale_by_keep

        def forward(self, x):
            if self.p == 0.0 or not self.training:
                return x
            keep = 1 - self.p
            r = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * (r.div_(keep) if keep > 0 and self.sbk else r)

    layers.DropPath, layers.trunc_normal_ = DropPath, nn.init.trunc_normal_
    timm.layers = layers
    sk, skm = types.ModuleType("skimage"), types.ModuleType("skimage.metrics")
    skm.structural_similarity = lambda *a, **k: float("nan")
    sk.metrics = skm
    sys.modules.update(
        {
            "timm": timm,
            "timm.layers": layers,
            "cv2": types.ModuleType("cv2"),
            "gdown": types.ModuleType("gdown"),
            "skimage": sk,
            "skimage.metrics": skm,
        }
    )
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    import studiosr.models as M  # noqa
    import studiosr.models.common as C  # noqa
    import studiosr.models.hat as H  # noqa
    import studiosr.models.rcan as R  # noqa
    import studiosr.models.swinir as S  # noqa
    import studiosr.utils.metrics as MT  # noqa

    return M, C, S, H, R, MT


def randomize(model: nn.Module, seed: int) -> None:
    """The reference zero-inits biases / ones LN weights; perturb every float tensor that is
    constant so that nothing is trivially absent from the vectors.  Frozen MeanShift convs and
    integer index buffers are left alone."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            if name.endswith("relative_position_bias_table"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
            elif p.numel() > 1 and bool((p == p.flatten()[0]).all()):
                base = float(p.flatten()[0])
                p.copy_(base + torch.randn(p.shape, generator=g) * 0.1)
            elif p.ndim == 2 and name.endswith("weight"):  # trunc_normal(0.02) Linear: make attention non-flat
                p.mul_(4.0)


def sd_arrays(model: nn.Module, prefix: str = "sd/"):
    return {prefix + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def save(name: str, **arrays) -> None:
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def main() -> None:
    torch.manual_seed(0)
    torch.set_num_threads(4)
    M, C, S, H, R, MT = import_reference()
    rng = np.random.default_rng(0)

    # ---- F1 pixel shuffle (integer ramps, exact) --------------------------------
    arrs = {}
    for r in (2, 3, 4):
        x = torch.arange(2 * 3 * r * r * 5 * 7, dtype=torch.int32).reshape(2, 3 * r * r, 5, 7)
        arrs[f"in_r{r}"] = x.numpy()
        arrs[f"out_r{r}"] = nn.PixelShuffle(r)(x.float()).to(torch.int32).numpy()
    save("f01_pixel_shuffle", **arrs)

    # ---- F2 window partition / reverse / roll ------------------------------------
    x = torch.arange(2 * 16 * 24 * 3, dtype=torch.int32).reshape(2, 16, 24, 3)
    rolled = torch.roll(x, (-4, -4), (1, 2))
    part = C.window_partition(rolled, 8)
    back = torch.roll(C.window_reverse(part, 8, 16, 24), (4, 4), (1, 2))
    save("f02_window", x=x.numpy(), part_shift4=part.numpy(), part_shift0=C.window_partition(x, 8).numpy(), back=back.numpy())

    # ---- F3 masks, F4 rpi ---------------------------------------------------------
    arrs = {}
    for (h, w, ws, sh) in [(72, 72, 8, 4), (64, 64, 8, 4), (16, 16, 8, 4), (64, 64, 16, 8), (8, 8, 8, 0), (16, 24, 8, 0), (24, 40, 8, 4), (32, 48, 16, 8)]:
        arrs[f"mask_{h}_{w}_{ws}_{sh}"] = C.calculate_mask((h, w), ws, sh).numpy().astype(np.float32)
    wa = S.WindowAttention(12, (8, 8), 6)
    arrs["rpi_swin_8"] = wa.relative_position_index.numpy()
    hat = M.HAT(embed_dim=12, depths=[1], num_heads=[2], window_size=16)
    arrs["rpi_hat_sa_16"] = hat.relative_position_index_SA.numpy()
    arrs["rpi_hat_oca_16"] = hat.relative_position_index_OCA.numpy()
    hat8 = M.HAT(embed_dim=12, depths=[1], num_heads=[2], window_size=8)
    arrs["rpi_hat_sa_8"] = hat8.relative_position_index_SA.numpy()
    arrs["rpi_hat_oca_8"] = hat8.relative_position_index_OCA.numpy()
    save("f03_mask_rpi", **arrs)

    # ---- F5 padding ------------------------------------------------------------------
    arrs = {}
    for (h, w) in [(5, 6), (8, 8), (12, 12), (13, 17), (64, 64)]:
        x = torch.arange(1 * 2 * h * w, dtype=torch.float32).reshape(1, 2, h, w)
        arrs[f"eval_{h}_{w}"] = S.check_image_size_for_eval(x, 8).numpy()
        if h >= 8:  # reflect pad needs pad < size
            arrs[f"reflect_{h}_{w}"] = C.check_image_size(x, 8).numpy()
    save("f05_pad", **arrs)

    # ---- F6 WindowAttention, F7 SwinTransformerBlock, F8 Mlp -------------------------
    wa = S.WindowAttention(180, (8, 8), 6).eval()
    randomize(wa, 1)
    x = torch.randn(18, 64, 180)
    mask = C.calculate_mask((24, 24), 8, 4)
    with torch.no_grad():
        save("f06_window_attention", x=x.numpy(), mask=mask.numpy(), y_nomask=wa(x).numpy(), y_mask=wa(x, mask).numpy(), **sd_arrays(wa))
    arrs = {}
    x = torch.randn(1, 24, 24, 180)
    arrs["x"] = x.numpy()
    for sh in (0, 4):
        blk = S.SwinTransformerBlock(180, 6, 8, sh, mlp_ratio=2.0).eval()
        randomize(blk, 2 + sh)
        with torch.no_grad():
            arrs[f"y_shift{sh}"] = blk(x).numpy()
        arrs.update(sd_arrays(blk, f"sd{sh}/"))
    save("f07_swin_block", **arrs)
    mlp = C.Mlp(180, 360).eval()
    randomize(mlp, 5)
    x = torch.randn(3, 50, 180)
    with torch.no_grad():
        save("f08_mlp", x=x.numpy(), y=mlp(x).numpy(), **sd_arrays(mlp))

    # ---- F9 conv blocks, F10 upsampler ---------------------------------------------
    arrs = {}
    x = torch.randn(2, 64, 12, 12)
    rb = C.ResBlock(64, 3, 0.1).eval()
    randomize(rb, 6)
    rcab = R.RCAB(64, 3, 16).eval()
    randomize(rcab, 7)
    ca = C.ChannelAttention(64, 16).eval()
    randomize(ca, 8)
    with torch.no_grad():
        arrs.update(x=x.numpy(), y_resblock=rb(x.clone()).numpy(), y_rcab=rcab(x.clone()).numpy(), y_ca=ca(x.clone()).numpy())
    arrs.update(sd_arrays(rb, "resblock/"))
    arrs.update(sd_arrays(rcab, "rcab/"))
    arrs.update(sd_arrays(ca, "ca/"))
    save("f09_conv_blocks", **arrs)
    arrs = {}
    x = torch.randn(1, 16, 6, 7)
    arrs["x"] = x.numpy()
    for tag, up in {"s2": C.Upsampler(2, 16), "s3": C.Upsampler(3, 16), "s4": C.Upsampler(4, 16), "s4direct": C.Upsampler(4, 16, 3)}.items():
        up = up.eval()
        randomize(up, 9)
        with torch.no_grad():
            arrs["y_" + tag] = up(x).numpy()
        arrs.update(sd_arrays(up, tag + "/"))
    save("f10_upsampler", **arrs)

    # ---- F11 reduced whole models -----------------------------------------------------
    def whole(name, ctor, cfg_kwargs, shapes, seed, train_too=False, in_scale=1.0):
        model = ctor(**cfg_kwargs).eval()
        randomize(model, seed)
        arrs = {"cfg": np.array(json.dumps(model.get_model_config()))}
        arrs.update(sd_arrays(model))
        for (b, h, w) in shapes:
            x = torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(seed + h * 31 + w)) * in_scale
            arrs[f"x_{b}_{h}_{w}"] = x.numpy()
            with torch.no_grad():
                model.eval()
                arrs[f"y_eval_{b}_{h}_{w}"] = model(x.clone()).numpy()
                if train_too:
                    model.train()
                    arrs[f"y_train_{b}_{h}_{w}"] = model(x.clone()).numpy()
                    model.eval()
        save(name, **arrs)
        return model

    small = [(1, 8, 8), (2, 12, 12), (1, 13, 17)]
    for sc in (2, 3, 4):
        whole(f"f11_swinir_x{sc}", M.SwinIR, dict(scale=sc, embed_dim=60, depths=[2, 2], num_heads=[6, 6], drop_path_rate=0.0), small, 10 + sc, train_too=True)
    whole("f11_swinir_direct_x4", M.SwinIR, dict(scale=4, embed_dim=60, depths=[2], num_heads=[6], upsampler="pixelshuffledirect", drop_path_rate=0.0), small, 15, train_too=True)
    whole("f11_swinir_c180_x4", M.SwinIR, dict(scale=4, embed_dim=180, depths=[2], num_heads=[6], drop_path_rate=0.0), [(1, 16, 16)], 16)
    for sc in (2, 3, 4):
        whole(f"f11_edsr_x{sc}", M.EDSR, dict(scale=sc, n_feats=32, n_resblocks=2), small, 20 + sc)
    whole("f11_edsr_r255_x2", M.EDSR, dict(scale=2, n_feats=64, n_resblocks=2, res_scale=1.0, img_range=255.0), [(1, 12, 12)], 24, in_scale=255.0)
    rc = whole("f11_rcan_x4", M.RCAN, dict(scale=4, n_feats=32, n_resblocks=2, n_resgroups=2, reduction=8), small, 30)
    whole("f11_rcan_x3", M.RCAN, dict(scale=3, n_feats=32, n_resblocks=2, n_resgroups=2, reduction=8), small[:2], 31)
    whole("f11_hat_w8_x4", M.HAT, dict(scale=4, embed_dim=60, depths=[2], num_heads=[6], window_size=8, drop_path_rate=0.0), small + [(1, 24, 16)], 40, train_too=True)
    whole("f11_hat_w16_x2", M.HAT, dict(scale=2, embed_dim=48, depths=[2, 2], num_heads=[4, 4], window_size=16, drop_path_rate=0.0, squeeze_factor=12), [(1, 16, 16), (1, 20, 36)], 41)

    # ---- F12 Model.inference + self ensemble (uint8 round trip) ---------------------------
    img = rng.integers(0, 256, size=(9, 11, 3), dtype=np.uint8)
    ed = M.EDSR(scale=2, n_feats=32, n_resblocks=2).eval()
    randomize(ed, 50)
    save("f12_inference", cfg=np.array(json.dumps(ed.get_model_config())), img=img, y=ed.inference(img), y_ens=ed.inference_with_self_ensemble(img), **sd_arrays(ed))

    # ---- F13 HAT single blocks --------------------------------------------------------------
    arrs = {}
    cab = H.CAB(60, 3, 30).eval()
    randomize(cab, 60)
    x = torch.randn(2, 60, 16, 16)
    with torch.no_grad():
        arrs.update(x_cab=x.numpy(), y_cab=cab(x).numpy())
    arrs.update(sd_arrays(cab, "cab/"))
    hatm = M.HAT(embed_dim=60, depths=[1], num_heads=[6], window_size=8)
    rpi_sa, rpi_oca = hatm.relative_position_index_SA, hatm.relative_position_index_OCA
    mask = C.calculate_mask((16, 24), 8, 4)
    x = torch.randn(2, 16 * 24, 60)
    arrs["x_tok"] = x.numpy()
    for sh in (0, 4):
        hab = H.HAB(60, 6, 8, sh, mlp_ratio=2.0).eval()
        randomize(hab, 61 + sh)
        with torch.no_grad():
            arrs[f"y_hab{sh}"] = hab(x, (16, 24), rpi_sa, mask).numpy()
        arrs.update(sd_arrays(hab, f"hab{sh}/"))
    oc = H.OCAB(60, 6, 8, 2.0, 0.5).eval()
    randomize(oc, 66)
    with torch.no_grad():
        arrs["y_ocab"] = oc(x, (16, 24), rpi_oca).numpy()
    arrs.update(sd_arrays(oc, "ocab/"))
    save("f13_hat_blocks", **arrs)

    # ---- F14 PSNR metric ------------------------------------------------------------------------
    a = rng.integers(0, 256, size=(40, 36, 3), dtype=np.uint8)
    b = rng.integers(0, 256, size=(40, 36, 3), dtype=np.uint8)
    save(
        "f14_psnr",
        a=a,
        b=b,
        psnr_rgb=np.float64(MT.compute_psnr(a, b)),
        psnr_y_crop4=np.float64(MT.compute_psnr(a, b, y_only=True, crop_border=4)),
        psnr_float=np.float64(MT.compute_psnr(a / 255.0, b / 255.0)),
        y=MT.to_y(a),
    )
    del rc
    grads(M)


def grads(M) -> None:
    """F15: one training step's gradients of reduced models under the reference Trainer's loss (trainer.py:97-109: train mode,
    forward, L1 loss, backward; DropPath off so the vectors are deterministic).  Every parameter gradient is stored."""
    import torch.nn.functional as F

    torch.set_num_threads(1)  # index_put(accumulate) of the bias-table gradient sums in thread order: one thread makes the files bit-reproducible
    from studiosr.models.han import HAN
    from studiosr.models.swinfir import SwinFIR

    cases = [
        ("swinir", M.SwinIR, dict(scale=2, embed_dim=60, depths=[2, 2], num_heads=[6, 6], drop_path_rate=0.0), (2, 13, 17), 70),
        ("swinir_direct", M.SwinIR, dict(scale=3, embed_dim=60, depths=[2], num_heads=[6], upsampler="pixelshuffledirect", drop_path_rate=0.0), (1, 16, 16), 71),
        ("hat", M.HAT, dict(scale=2, embed_dim=60, depths=[2], num_heads=[6], window_size=8, drop_path_rate=0.0), (2, 16, 24), 72),
        ("edsr", M.EDSR, dict(scale=4, n_feats=32, n_resblocks=2), (2, 12, 10), 73),
        ("rcan", M.RCAN, dict(scale=3, n_feats=32, n_resblocks=2, n_resgroups=2, reduction=8), (2, 9, 12), 74),
        # row f-4 models: SwinFIR (SFB with the rFFT branch; odd and even widths) and HAN (LAM + CSAM; han.py:87 hard-codes 11 groups)
        ("swinfir", SwinFIR, dict(scale=2, embed_dim=60, depths=[2], num_heads=[6], drop_path_rate=0.0), (2, 13, 18), 75),
        ("han", HAN, dict(scale=2, n_feats=16, n_resblocks=1, n_resgroups=10, reduction=4), (2, 9, 10), 76),
    ]
    for tag, ctor, cfg, (b, h, w), seed in cases:
        torch.manual_seed(seed)  # parameter init draws from the global generator: seed it per case so that every file is reproducible on its own
        model = ctor(**cfg)
        randomize(model, seed)
        model.train()
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(b, 3, h, w, generator=g)
        sc = cfg["scale"]
        tgt = torch.rand(b, 3, h * sc, w * sc, generator=g)
        if tag in ("swinfir", "han"):  # also the eval-mode forward (SwinFIR: eval padding); HAN's zero-initialised gammas made non-trivial
            with torch.no_grad():
                for n_, p_ in model.named_parameters():
                    if n_.endswith("gamma"):
                        p_.fill_(0.3)
                model.eval()
                y_eval = model(x).numpy()
                model.train()
        out = model(x)
        loss = F.l1_loss(out, tgt)
        loss.backward()
        arrs = {"cfg": np.array(json.dumps(model.get_model_config())), "x": x.numpy(), "target": tgt.numpy(), "out": out.detach().numpy(), "loss": np.float64(loss.item())}
        arrs.update(sd_arrays(model))
        for n_, p_ in model.named_parameters():
            if p_.grad is not None:
                arrs["grad/" + n_] = p_.grad.numpy().copy()
        if tag in ("swinfir", "han"):
            arrs["y_eval"] = y_eval
        if tag in ("swinir", "hat", "edsr", "rcan"):
            # the same step under the reference Trainer's autocast context (trainer.py:80,102; CPU autocast as its proxy here): how far the
            # REFERENCE's own bf16 step sits from its fp32 step -- the yardstick for our bf16-operand step (tests/test_training.py)
            model.zero_grad()
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out_ac = model(x)
                loss_ac = F.l1_loss(out_ac.float(), tgt)
            loss_ac.backward()
            arrs["loss_ac"] = np.float64(loss_ac.item())
            for n_, p_ in model.named_parameters():
                if p_.grad is not None:
                    arrs["grad_ac/" + n_] = p_.grad.float().numpy().copy()
        save(f"f15_grads_{tag}", **arrs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--grads-only":
        grads(import_reference()[0])  # (every case is independently seeded: regenerating leaves the other fixtures byte-identical)
    else:
        main()
