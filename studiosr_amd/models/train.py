"""Differentiable forwards (the training path): the same modules / parameters as the inference drivers, executed as a graph of
studiosr_amd.autograd ops so that `loss.backward()` reaches every parameter through HIP backward kernels.

Each function restates one reference forward on NHWC / token-major fp32 tensors and cites it:
  SwinIR  studiosr/models/swinir.py:146-174 (block), :245-246 (RSTB), :342-372 (model)
  HAT     studiosr/models/hat.py:153-195 (HAB), :239-293 (OCAB), :385 (RHAG), :519-554 (model)
  EDSR    studiosr/models/edsr.py:39-48, ResBlock common.py:150-153
  RCAN    studiosr/models/rcan.py:21-24,33-36,68-77
Used whenever autograd is recording (Model.forward dispatches here), in train() and in eval() mode alike; DropPath follows the
module's `training` flag exactly as timm's does.
"""
from __future__ import annotations

from typing import List

import torch

from .. import _lib as L
from .. import autograd as A
from .. import ops

Tensor = torch.Tensor
RGB_MEAN = (0.4488, 0.4371, 0.4040)


def _ingest(x: Tensor, Hp: int, Wp: int, pad_mode: int, scale: Tensor, bias: Tensor) -> Tensor:
    """NCHW image -> normalised, padded NHWC image in a 32-channel buffer (channels 3.. are zero); no gradient flows to the image."""
    B = x.shape[0]
    xin = torch.empty(B, Hp, Wp, 32, device=x.device, dtype=torch.float32)
    ops.ingest_nchw(x.detach().to(torch.float32).contiguous(), xin, pad_mode, scale, bias)
    return xin


_AFFINE_CACHE = {}


def _affines(img_range: float, n_colors: int, device):
    """Normalizer affines (common.py:222-233) as device vectors, built once per (range, colours, device): four pageable H2D copies per
    forward would each synchronise the host against the launch queue."""
    key = (float(img_range), int(n_colors), str(device))
    hit = _AFFINE_CACHE.get(key)
    if hit is None:
        mean = torch.tensor(RGB_MEAN[:n_colors], dtype=torch.float32, device=device)
        ing = (torch.full((n_colors,), 1.0 / float(img_range), dtype=torch.float32, device=device), (-mean).contiguous())  # x / range - mean
        fin = (torch.full((n_colors,), float(img_range), dtype=torch.float32, device=device), (mean * img_range).contiguous())  # (x + mean) * range
        if len(_AFFINE_CACHE) > 64:
            _AFFINE_CACHE.clear()
        hit = _AFFINE_CACHE[key] = (ing, fin)
    return hit


def _conv(x: Tensor, m: torch.nn.Conv2d, cin=None) -> Tensor:
    return A.conv3x3(x, m.weight, m.bias, cin)


def _upsampler(up, x: Tensor) -> Tensor:
    """conv -> PixelShuffle stages (common.py:124-137)."""
    for idx, r, _ in up.stages:
        x = A.pixel_shuffle(_conv(x, up[idx]), r)
    return x


def _mlp(mlp, x: Tensor) -> Tensor:
    return A.linear(A.gelu(A.linear(x, mlp.fc1.weight, mlp.fc1.bias)), mlp.fc2.weight, mlp.fc2.bias)


def _check_dropout(model) -> None:
    if model.training and (getattr(model, "drop_rate", 0.0) > 0.0 or getattr(model, "attn_drop_rate", 0.0) > 0.0):
        raise NotImplementedError("drop_rate / attn_drop_rate > 0 (nn.Dropout inside attention / MLP) is not part of the HIP training path; the reference defaults are 0")


def _drop_rates(model) -> List[float]:
    n = sum(model.depths)
    return [float(v) for v in torch.linspace(0, model.drop_path_rate, n)] if n else []  # swinir.py:296, hat.py:440


# --------------------------------------------------------------------------- SwinIR / SwinFIR
def _conv1x1(x: Tensor, m: torch.nn.Conv2d) -> Tensor:
    return A.linear(x, m.weight, m.bias)


def _sfb(m, x: Tensor) -> Tensor:
    """SwinFIR's SFB (swinfir.py:38-81): spatial branch conv-LeakyReLU(0.2)-conv + x; spectral branch 1x1 conv + LeakyReLU ->
    FourierUnit (rfftn -> 1x1 conv on (real | imag) + LeakyReLU -> irfftn) -> 1x1 conv(fu + y); 1x1 fusion conv of the concatenation."""
    s = A.add(_conv(A.leaky_relu(_conv(x, m.S.body[0]), 0.2), m.S.body[2]), x)
    return _sfb_tail(m, x, s)


def _sfb_tail(m, x: Tensor, s: Tensor) -> Tensor:
    """The SFB after its spatial branch s = S(x): spectral branch on x, 1x1 fusion of (s | f)."""
    y = A.leaky_relu(_conv1x1(x, m.F.conv_before_fft[0]), 0.2)
    z = A.leaky_relu(_conv1x1(A.rfft2(y), m.F.fu.conv_layer), 0.2)
    f = _conv1x1(A.add(A.irfft2(z, y.shape[2]), y), m.F.conv_after_fft)
    return _conv1x1(A.concat(s, f), m.fusion)


def _resi(m, t: Tensor) -> Tensor:
    """RSTB.conv / conv_after_body: a 3x3 conv (SwinIR) or an SFB (SwinFIR's resi_connection, swinir.py:241, swinfir.py:112-114)."""
    return _conv(t, m) if isinstance(m, torch.nn.Conv2d) else _sfb(m, t)


def _window_msa(attn, t: Tensor, ws: int, shift: int, heads: int, rpi: Tensor) -> Tensor:
    B, H, W, Cn = t.shape
    win = A.window_partition(t, ws, shift)
    qkv = A.linear(win, attn.qkv.weight, attn.qkv.bias)
    mask = A.shift_mask(H, W, ws, shift, t.device) if shift > 0 else None  # the reference adds an all-zero mask when shift == 0
    o = A.window_attention_packed(qkv, attn.relative_position_bias_table, rpi, mask, heads, ws * ws, Cn)
    return A.window_reverse(A.linear(o, attn.proj.weight, attn.proj.bias), ws, shift, t.shape)


def swinir_forward(model, x: Tensor) -> Tensor:
    _check_dropout(model)
    B, _, H, W = x.shape
    ws, s = model.window_size, model.scale
    if model.training:  # check_image_size: reflect pad to the next multiple (swinir.py:356, common.py:277-282)
        Hp, Wp = H + (ws - H % ws) % ws, W + (ws - W % ws) % ws
        pad_mode = L.PAD_REFLECT if (Hp != H or Wp != W) else L.PAD_NONE
        if Hp - H >= H or Wp - W >= W:
            raise RuntimeError("Padding size should be less than the corresponding input dimension (reflect pad)")
    else:  # check_image_size_for_eval (swinir.py:249-255)
        Hp, Wp, pad_mode = (H // ws + 1) * ws, (W // ws + 1) * ws, L.PAD_EVAL_MIRROR
    ing, fin = _affines(model.img_range, model.n_colors, x.device)
    if model.training:  # the whole step as fused launches (studiosr_amd/fasttrain.py) when the geometry is the default one, as HAT's
        plan = _fast_plan(model, B, Hp, Wp)
        if plan is not None and plan.full:
            from .. import fasttrain

            return fasttrain.run_model(plan, x)
    first = _conv(_ingest(x, Hp, Wp, pad_mode, *ing), model.conv_first, cin=model.n_colors)
    t = A.layer_norm(first, model.patch_embed.norm.weight, model.patch_embed.norm.bias)
    dpr = _drop_rates(model)
    k = 0
    for li, layer in enumerate(model.layers):
        tin = t
        heads = model.num_heads[li]
        for blk in layer.residual_group.blocks:
            a = _window_msa(blk.attn, A.layer_norm(t, blk.norm1.weight, blk.norm1.bias), ws, blk.shift_size, heads, blk.attn.relative_position_index)
            t = A.add(t, A.drop_path(a, dpr[k], model.training))
            m = _mlp(blk.mlp, A.layer_norm(t, blk.norm2.weight, blk.norm2.bias))
            t = A.add(t, A.drop_path(m, dpr[k], model.training))
            k += 1
        t = A.add(_resi(layer.conv, t), tin)  # swinir.py:245-246
    t = A.layer_norm(t, model.norm.weight, model.norm.bias)
    body = A.add(_resi(model.conv_after_body, t), first)
    if model.upsampler == "pixelshuffle":
        f = A.leaky_relu(_conv(body, model.conv_before_upsample[0]), 0.01)
        y = _conv(_upsampler(model.upsample, f), model.conv_last)
    else:  # pixelshuffledirect (swinir.py:367-369)
        y = _upsampler(model.upsample, body)
    return A.nhwc_out(y, *fin, H * s, W * s)


# --------------------------------------------------------------------------- HAT
REPLAN = 3  # recorded forwards in a row at another batch geometry before the fused training plan is rebuilt for it


def _fast_plan(model, B: int, Hp: int, Wp: int):
    """The fused training path (studiosr_amd/fasttrain.py) when it applies: bf16 autocast (the reference Trainer's context, trainer.py:80,102),
    the default block geometry, a padded size that is a multiple of the 16 x 16 windows.  SR_FAST_TRAIN=0 keeps the generic engine."""
    import os

    if os.environ.get("SR_FAST_TRAIN", "1") == "0" or not A.torch_autocast_bf16():
        return None
    from .. import fasttrain

    if Hp % model.window_size or Wp % model.window_size:
        return None
    plan = fasttrain.get_plan(model)
    if plan is None:
        return None
    if plan.geo is not None and plan.geo != (B, Hp, Wp):
        # One geometry per plan.  A stray other size (an evaluation inside a training run, a last partial batch) takes the generic engine; a geometry that PERSISTS
        # (REPLAN consecutive recorded forwards: a warm-up step before the real batch size, a new patch size) gets a new plan -- the old one's static buffers are dropped.
        other = getattr(plan, "_other_geo", None)
        plan._other_geo = ((B, Hp, Wp), other[1] + 1) if other and other[0] == (B, Hp, Wp) else ((B, Hp, Wp), 1)
        if plan._other_geo[1] >= REPLAN:
            object.__delattr__(model, "_fast_plan")
            plan = fasttrain.get_plan(model)
            if plan is None:
                return None
        else:
            if not getattr(plan, "_bypass_logged", False):
                plan._bypass_logged = True
                import warnings

                warnings.warn(f"studiosr_amd: the fused training plan is prepared for batch geometry {plan.geo}; {(B, Hp, Wp)} runs on the generic engine "
                              f"(correct, about 4x slower; gradients are then not views of the flat buffer, so the optimizer takes torch's own step); the plan is rebuilt "
                              f"for it if it persists for {REPLAN} forwards")
            return None
    elif plan.geo is not None:
        plan._other_geo = None
    plan.prepare(B, Hp, Wp)
    plan.pack()
    return plan


def _fast_habs(plan, li: int, t: Tensor, rates, training: bool) -> Tensor:
    from .. import fasttrain

    B = t.shape[0]
    scales = None
    if training and any(r > 0.0 for r in rates):  # DropPath: per block, per branch, per image Bernoulli(keep) / keep (hat.py:148,192-193)
        keep = 1.0 - torch.tensor(list(rates), dtype=torch.float32, device=t.device)
        scales = ((torch.rand(len(rates), 2, B, device=t.device) < keep[:, None, None]).to(torch.float32) / keep[:, None, None].clamp_min(1e-30)).contiguous()
    tp = torch.nn.functional.pad(t, (0, fasttrain.CP - t.shape[-1]))
    out = fasttrain.run_stage(plan.stages[li], tp, scales)
    return out[..., : t.shape[-1]].contiguous()


def _cab(cab, x: Tensor) -> Tensor:
    seq = cab.cab
    y = _conv(A.gelu(_conv(x, seq[0])), seq[2])
    att = seq[3].attention
    return A.channel_attention(y, att[1].weight, att[1].bias, att[3].weight, att[3].bias)


def hat_forward(model, x: Tensor) -> Tensor:
    _check_dropout(model)
    B, _, H, W = x.shape
    ws, s, Cn = model.window_size, model.scale, model.embed_dim
    Hp, Wp = H + (ws - H % ws) % ws, W + (ws - W % ws) % ws  # check_image_size: reflect pad (hat.py:544)
    if Hp - H >= H or Wp - W >= W:
        raise RuntimeError("Padding size should be less than the corresponding input dimension (reflect pad)")
    ing, fin = _affines(model.img_range, model.n_colors, x.device)
    plan = _fast_plan(model, B, Hp, Wp)
    if plan is not None and plan.full:  # the whole step as fused launches behind one autograd node
        from .. import fasttrain

        return fasttrain.run_model(plan, x)
    first = _conv(_ingest(x, Hp, Wp, L.PAD_REFLECT if (Hp != H or Wp != W) else L.PAD_NONE, *ing), model.conv_first, cin=model.n_colors)
    t = A.layer_norm(first, model.patch_embed.norm.weight, model.patch_embed.norm.bias)
    dpr = _drop_rates(model)
    rpi_sa, rpi_oca = model.relative_position_index_SA, model.relative_position_index_OCA
    k = 0
    for li, layer in enumerate(model.layers):
        tin = t
        heads = model.num_heads[li]
        grp = layer.residual_group
        if plan is not None:  # the six HABs as fused launches (studiosr_amd/fasttrain.py); the stream crosses in the kernels' 192-channel padding
            nb = len(grp.blocks)
            t = _fast_habs(plan, li, t, dpr[k:k + nb], model.training)
            k += nb
        else:
            for blk in grp.blocks:  # HAB (hat.py:153-195)
                n1 = A.layer_norm(t, blk.norm1.weight, blk.norm1.bias)
                conv_x = _cab(blk.conv_block, n1)
                a = _window_msa(blk.attn, n1, ws, blk.shift_size, heads, rpi_sa)
                t = A.add(A.add(t, A.drop_path(a, dpr[k], model.training)), conv_x, 1.0, blk.conv_scale)
                m = _mlp(blk.mlp, A.layer_norm(t, blk.norm2.weight, blk.norm2.bias))
                t = A.add(t, A.drop_path(m, dpr[k], model.training))
                k += 1
        if plan is None or not plan.with_oca:
            oc = grp.overlap_attn  # OCAB (hat.py:239-293): no DropPath
            wse = oc.overlap_win_size
            qkv = A.linear(A.layer_norm(t, oc.norm1.weight, oc.norm1.bias), oc.qkv.weight, oc.qkv.bias)  # [B,H,W,3C]
            o = A.cross_window_attention(A.window_partition(qkv, ws, 0), A.oca_unfold(qkv, ws, wse), oc.relative_position_bias_table, rpi_oca, heads, ws * ws, wse * wse, Cn)
            t = A.add(A.linear(A.window_reverse(o, ws, 0, t.shape), oc.proj.weight, oc.proj.bias), t)
            t = A.add(t, _mlp(oc.mlp, A.layer_norm(t, oc.norm2.weight, oc.norm2.bias)))
        t = A.add(_conv(t, layer.conv), tin)  # hat.py:385
    t = A.layer_norm(t, model.norm.weight, model.norm.bias)
    body = A.add(_conv(t, model.conv_after_body), first)
    f = A.leaky_relu(_conv(body, model.conv_before_upsample[0]), 0.01)
    y = _conv(_upsampler(model.upsample, f), model.conv_last)
    return A.nhwc_out(y, *fin, H * s, W * s)


# --------------------------------------------------------------------------- EDSR / RCAN
def _mean_shift(ms):
    """MeanShift (common.py:108-121) as a per-channel affine; its parameters are frozen, so the two small vectors are cached on the
    module and rebuilt only when the weight tensor changes identity / device (`.to()`, load_state_dict)."""
    w, b = ms.weight, ms.bias
    key = (w.data_ptr(), b.data_ptr(), w._version, b._version, str(w.device))
    hit = getattr(ms, "_sr_affine", None)
    if hit is None or hit[0] != key:
        hit = (key, torch.diagonal(w.detach().reshape(3, 3)).to(torch.float32).contiguous(), b.detach().to(torch.float32).contiguous())
        ms._sr_affine = hit
    return hit[1], hit[2]


def edsr_forward(model, x: Tensor) -> Tensor:
    B, _, H, W = x.shape
    s = model.scale
    h = _conv(_ingest(x, H, W, L.PAD_NONE, *_mean_shift(model.sub_mean)), model.head[0], cin=model.n_colors)
    r = h
    for i in range(model.n_resblocks):  # ResBlock: body(x) * res_scale + x (common.py:150-153)
        rb = model.body[i]
        r = A.add(_conv(A.relu(_conv(r, rb.body[0])), rb.body[2]), r, rb.res_scale, 1.0)
    r = A.add(_conv(r, model.body[model.n_resblocks]), h)
    y = _conv(_upsampler(model.tail[0], r), model.tail[1])
    return A.nhwc_out(y, *_mean_shift(model.add_mean), H * s, W * s)


def rcan_forward(model, x: Tensor) -> Tensor:
    B, _, H, W = x.shape
    s = model.scale
    h = _conv(_ingest(x, H, W, L.PAD_NONE, *_mean_shift(model.sub_mean)), model.head[0], cin=model.n_colors)
    g = h
    for gi in range(model.n_resgroups):
        grp = model.body[gi]
        r = g
        for bi in range(model.n_resblocks):  # RCAB (rcan.py:21-24)
            b = grp.body[bi].body
            du = b[3].conv_du
            y = _conv(A.relu(_conv(r, b[0])), b[2])
            r = A.add(A.channel_attention(y, du[0].weight, du[0].bias, du[2].weight, du[2].bias), r)
        g = A.add(_conv(r, grp.body[model.n_resblocks]), g)  # rcan.py:33-36
    r = A.add(_conv(g, model.body[model.n_resgroups]), h)
    y = _conv(_upsampler(model.tail[0], r), model.tail[1])
    return A.nhwc_out(y, *_mean_shift(model.add_mean), H * s, W * s)


def han_attention(model, feats: List[Tensor], h: Tensor) -> Tensor:
    """HAN's attention tail (han.py:96-113) on fp32 NHWC tensors: layer attention over the group outputs + channel-spatial attention on
    the last one, fused by two 3x3 convs, + the head feature.  feats = [group outputs..., body conv output]."""
    B, H, W, F = h.shape
    out1 = feats[-1]
    # LAM over the N = n_resgroups + 1 features, newest first (han.py:96-103): each feature is one vector of H*W*C values
    N = len(feats)
    stack = A.concat(*[f.reshape(B, H * W * F) for f in reversed(feats)]).reshape(B, N, H * W * F)
    la = A.add(A.scale_param(A.layer_attention(stack), model.la.gamma), stack)                      # gamma * out + x  (han.py:31)
    la = A.concat(*[A.slice_channels(la.reshape(B, N * H * W * F), n * H * W * F, H * W * F).reshape(B * H * W, F) for n in range(N)]).reshape(B, H, W, N * F)
    out2 = _conv(la, model.last_conv)
    # CSAM (han.py:44-53): x * (gamma * sigmoid(conv3d(x))) + x
    att = A.scale_param(A.sigmoid(A.conv3d_27(out1, model.csa.conv.weight, model.csa.conv.bias)), model.csa.gamma)
    out1 = A.add(A.mul(out1, att), out1)
    return A.add(_conv(A.concat(out1, out2), model.last), h)


def han_forward(model, x: Tensor) -> Tensor:
    """HAN.forward (han.py:92-115)."""
    B, _, H, W = x.shape
    s, F = model.scale, model.n_feats
    h = _conv(_ingest(x, H, W, L.PAD_NONE, *_mean_shift(model.sub_mean)), model.head[0], cin=model.n_colors)
    feats = []
    res = h
    for gi in range(model.n_resgroups):
        grp = model.body[gi]
        r = res
        for bi in range(model.n_resblocks):
            b = grp.body[bi].body
            du = b[3].conv_du
            y = _conv(A.relu(_conv(r, b[0])), b[2])
            r = A.add(A.channel_attention(y, du[0].weight, du[0].bias, du[2].weight, du[2].bias), r)
        res = A.add(_conv(r, grp.body[model.n_resblocks]), res)
        feats.append(res)
    res = _conv(res, model.body[model.n_resgroups])
    feats.append(res)
    res = han_attention(model, feats, h)
    y = _conv(_upsampler(model.tail[0], res), model.tail[1])
    return A.nhwc_out(y, *_mean_shift(model.add_mean), H * s, W * s)


FORWARDS = {"SwinIR": swinir_forward, "SwinFIR": swinir_forward, "HAT": hat_forward, "EDSR": edsr_forward, "RCAN": rcan_forward, "HAN": han_forward}
