"""CPU-side tests of the host logic: C-ABI library loads and exports what include/studiosr_hip.h declares, weight
packing is a pure (invertible) index shuffle, model classes mirror the reference's state_dict, and the product refuses
to run without a GPU (no CPU fallback)."""
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, golden_cfg, golden_sd, load_golden

import studiosr_amd as S
from studiosr_amd import _lib as L
from studiosr_amd import packing


def header_symbols():
    src = open(os.path.join(ROOT, "include", "studiosr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as G

    if not os.path.exists(L.LIB_PATH):
        G.build()
    lib = L.lib()
    names = header_symbols()
    assert names, "no symbols parsed from the header"
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/studiosr_hip.h but not exported"
        assert n in L.SYMBOLS, f"{n} has no ctypes prototype in studiosr_amd/_lib.py"
    assert sorted(L.SYMBOLS) == names
    assert lib.sr_abi_version() == L.ABI_VERSION == 11


def test_ctypes_struct_sizes_match_the_c_header(tmp_path):
    """Compile a tiny C program against the header and compare sizeof() of every argument struct."""
    import subprocess

    structs = ["SrGemm", "SrConv3x3", "SrWindowAttn", "SrOcaAttn", "SrChannelAttn", "SrMlp", "SrSwinBlock", "SrSwinQkv", "SrSwinTail", "SrSwinLight", "SrRcab", "SrCab", "SrBgemm", "SrTrWgradJob", "SrTrAttnBwd", "SrTrAttnFwd", "SrTrOcaFold", "SrTrQkvFwd", "SrTrTailFwd", "SrTrTailBwd", "SrTrQkvBwd", "SrTrCaBwd", "SrTrLnBwd", "SrTrGelu", "SrTrAdd", "SrTrFinalize", "SrTrUnshuffle", "SrTrLreluBwd", "SrLayernorm", "SrPlanOp"]
    c = tmp_path / "sz.c"
    c.write_text('#include <stdio.h>\n#include "studiosr_hip.h"\nint main(){' + "".join(f'printf("{s} %zu\\n", sizeof({s}));' for s in structs) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    import ctypes

    for s in structs:
        assert int(out[s]) == ctypes.sizeof(getattr(L, s)), s


def test_fragment_packing_is_an_index_shuffle():
    torch.manual_seed(0)
    w = torch.randn(48, 96)
    frag = packing.to_fragments(w, torch.float32).reshape(48 // 16, 96 // 32, 64, 8)
    for (nt, kc, lane, j) in [(0, 0, 0, 0), (2, 1, 37, 5), (1, 2, 63, 7), (0, 2, 16, 3)]:
        assert frag[nt, kc, lane, j] == w[16 * nt + (lane & 15), 32 * kc + 8 * (lane >> 4) + j]
    assert torch.equal(torch.sort(frag.reshape(-1))[0], torch.sort(w.reshape(-1))[0])


def test_pixel_shuffle_rows_and_bias_fragments():
    rows = packing.pixel_shuffle_rows(3, 4, 2)  # packed row (i*r+j)*cps_p + c  <-  c*r*r + i*r + j
    assert rows.tolist() == [0, 4, 8, -1, 1, 5, 9, -1, 2, 6, 10, -1, 3, 7, 11, -1]
    b = torch.arange(2 * 64 * 64, dtype=torch.float32).reshape(2, 64, 64)
    fr = packing.bias_fragments(b).reshape(2, 4, 4, 64, 4)
    for (h, qt, kt, lane, r) in [(0, 0, 0, 0, 0), (1, 3, 2, 45, 3), (0, 2, 1, 17, 1)]:
        assert fr[h, qt, kt, lane, r] == b[h, 16 * qt + (lane & 15), 16 * kt + 4 * (lane >> 4) + r]


def test_layernorm_folding_is_exact_algebra():
    torch.manual_seed(1)
    x = torch.randn(7, 20, dtype=torch.float64)
    w, b = torch.randn(12, 20, dtype=torch.float64), torch.randn(12, dtype=torch.float64)
    g, be = torch.randn(20, dtype=torch.float64), torch.randn(20, dtype=torch.float64)
    ref = F.linear(F.layer_norm(x, (20,), g, be, 1e-5), w, b)
    wf, bf = packing.fold_layernorm(w.float(), b.float(), g.float(), be.float())
    xn = F.layer_norm(x, (20,), None, None, 1e-5)
    torch.testing.assert_close(F.linear(xn, wf.double(), bf.double()), ref, rtol=1e-5, atol=1e-5)


def test_gather_bias_wraps_negative_indices_like_python():
    table = torch.arange(10.0).reshape(5, 2)
    rpi = torch.tensor([[-1, 0], [4, -5]])
    out = packing.gather_bias(table, rpi, 2, 2)  # [heads, 2, 2]
    assert torch.equal(out[0], table[rpi.reshape(-1), 0].reshape(2, 2))


WHOLE = [("f11_swinir_x4", "SwinIR"), ("f11_swinir_direct_x4", "SwinIR"), ("f11_edsr_x3", "EDSR"), ("f11_rcan_x4", "RCAN"),
         ("f11_hat_w8_x4", "HAT"), ("f11_hat_w16_x2", "HAT")]


@pytest.mark.parametrize("name,kind", WHOLE)
def test_model_surface_matches_reference(name, kind):
    """Same kwargs -> same state_dict keys / shapes / integer buffers and the same get_model_config()."""
    g = load_golden(name)
    cfg, sd = golden_cfg(g), golden_sd(g)
    m = getattr(S, kind)(**cfg)
    own = m.state_dict()
    assert sorted(own) == sorted(sd)
    for k, v in sd.items():
        assert tuple(own[k].shape) == tuple(v.shape), k
        if not v.is_floating_point():
            assert torch.equal(own[k], v), k
    m.load_state_dict(sd)  # strict
    assert m.get_model_config() == cfg
    assert json.dumps(m.get_model_config())  # JSON-serialisable like the reference's params.json
    assert (m.scale, m.n_colors, m.img_range) == (cfg["scale"], cfg["n_colors"], cfg["img_range"])


def test_training_configs_match_reference_literals():
    assert S.SwinIR(embed_dim=60, depths=[1], num_heads=[6]).get_training_config()["milestones"] == [250000, 400000, 450000, 475000]
    assert S.EDSR(n_feats=32, n_resblocks=1).get_training_config()["batch_size"] == 16
    assert S.RCAN(n_feats=32, n_resblocks=1, n_resgroups=1).get_training_config()["learning_rate"] == 0.0001
    assert S.HAT(embed_dim=60, depths=[1], num_heads=[6], window_size=8).get_training_config() == {}


def test_no_cpu_fallback():
    m = S.EDSR(scale=2, n_feats=32, n_resblocks=1).eval()
    with pytest.raises(L.HipLibraryError):
        m(torch.rand(1, 3, 8, 8))
    with pytest.raises(L.HipLibraryError):
        m.inference(np.zeros((8, 8, 3), np.uint8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "studiosr_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the test oracle"


def test_tile_partition():
    from studiosr_amd.parallel import tile_partition

    assert tile_partition(8, 3) == [(0, 3), (3, 6), (6, 8)]
    assert tile_partition(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    for n in range(0, 20):
        for w in (1, 2, 3, 8):
            parts = tile_partition(n, w)
            assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))


def test_graft_entry_build_runs():
    """The driver's build check: compiles (incrementally) and loads the library, checks the ABI version."""
    import __graft_entry__ as g

    g.build()



def test_swin_block_stream_packing_places_every_fragment():
    """packing.pack_swin_block_stream (vectorised) against a fragment-by-fragment restatement of the layout the kernel reads
    (sr_swin_block3.hip: slot s, fragment 3 w + t, lane 16 g + i, element j  <->  row 16 tile + i, k = 32 chunk + 8 g + j), including the
    bias rows on the constant-one channels and the folded scales."""
    import torch

    from studiosr_amd import packing as P

    g = torch.Generator().manual_seed(3)
    C, heads, hidden, hd = 180, 6, 360, 30
    qkv_w, qkv_b = torch.randn(3 * C, C, generator=g) * 0.05, torch.randn(3 * C, generator=g) * 0.1
    proj_w, proj_b = torch.randn(C, C, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1
    fc1_w, fc1_b = torch.randn(hidden, C, generator=g) * 0.05, torch.randn(hidden, generator=g) * 0.1
    fc2_w, fc2_b = torch.randn(C, hidden, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1
    st = P.pack_swin_block_stream(qkv_w, qkv_b, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b, C, heads, hidden).to(torch.float32).reshape(48, 12, 64, 8)
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    qs = hd ** -0.5 * P.LOG2E
    c0 = P.gelu_bf16_value(1.0)
    assert 0.83 < c0 < 0.85

    def frag(slot, f, i, k):  # element (row i of the tile, k of the 32-wide chunk)
        return float(st[slot, f, 16 * (k // 8) + i, k % 8])

    rng = torch.Generator().manual_seed(4)
    for _ in range(300):
        p, c, w, i, k = (int(torch.randint(0, n, (1,), generator=rng)) for n in (3, 6, 4, 16, 32))
        hh, half = w >> 1, w & 1
        head, d, ch = 2 * p + hh, 16 * half + i, 32 * c + k
        for t in range(3):  # q, k, v tiles of the QKV slot
            if d < hd and ch < C:
                want = float(bf(qkv_w[t * C + head * hd + d, ch] * (qs if t == 0 else 1.0)))
            elif t == 0 and d < hd and ch in (C, C + 1):
                b = qkv_b[head * hd + d] * qs
                want = float(bf(b)) if ch == C else float(bf(b - bf(b)))
            elif t == 2 and d == hd and ch == C:
                want = 1.0
            else:
                want = 0.0
            assert frag(8 * p + c, 3 * w + t, i, k) == want, (p, c, w, t, i, k)
        # fc1 / fc2 slots of hidden half hf = p % 2
        hf, n = p % 2, c % 3
        row = 192 * hf + 48 * w + 16 * n + i
        if row < hidden:
            want = float(bf(fc1_w[row, ch])) if ch < C else (float(bf(fc1_b[row])) if ch == C else (float(bf(fc1_b[row] - bf(fc1_b[row]))) if ch == C + 1 else 0.0))
        else:
            want = 1.0 if (row in (hidden, hidden + 1) and ch == C) else 0.0
        assert frag(24 + 12 * hf + c, 3 * w + n, i, k) == want, ("fc1", hf, c, w, n, i, k)
        och, hcol = 48 * w + 16 * n + i, 192 * hf + 32 * c + k
        if och < C and hcol < hidden:
            want = float(bf(fc2_w[och, hcol]))
        elif och < C and hcol in (hidden, hidden + 1):
            q = fc2_b[och] / c0
            want = float(bf(q)) if hcol == hidden else float(bf(q - bf(q)))
        else:
            want = 0.0
        assert frag(30 + 12 * hf + c, 3 * w + n, i, k) == want, ("fc2", hf, c, w, n, i, k)
        # proj slot c2 = c % 2 of pass p: rows = output channels, k = feature d of head 2p + c2
        c2 = c % 2
        if och < C and k < hd:
            want = float(bf(proj_w[och, (2 * p + c2) * hd + k]))
        elif och < C and k == hd and 2 * p + c2 in (0, 1):
            bpf = proj_b + proj_w @ qkv_b[2 * C:]
            want = float(bf(bpf[och])) if 2 * p + c2 == 0 else float(bf(bpf[och] - bf(bpf[och])))
        else:
            want = 0.0
        assert frag(8 * p + 6 + c2, 3 * w + n, i, k) == want, ("proj", p, c2, w, n, i, k)


def test_swin_qkv_and_tail_stream_packing_places_every_fragment():
    """packing.pack_swin_qkv_stream / pack_swin_tail_stream (HAT: sr_swin_qkv, sr_swin_tail) against a fragment-by-fragment restatement of the
    layout the kernels read: slot, fragment 3 w + t, lane 16 g + i, element j  <->  row 16 tile + i, k = 32 chunk + 8 g + j."""
    import torch

    from studiosr_amd import packing as P

    g = torch.Generator().manual_seed(5)
    C, heads, hidden, hd = 180, 6, 360, 30
    qkv_w, qkv_b = torch.randn(3 * C, C, generator=g) * 0.05, torch.randn(3 * C, generator=g) * 0.1
    proj_w = torch.randn(C, C, generator=g) * 0.05
    fc1_w, fc1_b = torch.randn(hidden, C, generator=g) * 0.05, torch.randn(hidden, generator=g) * 0.1
    fc2_w, fc2_b = torch.randn(C, hidden, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    sq = P.pack_swin_qkv_stream(qkv_w, qkv_b, C, heads).to(torch.float32).reshape(P.SWIN_QKV_SLOTS, 12, 64, 8)
    stl = P.pack_swin_tail_stream(proj_w, fc1_w, fc1_b, fc2_w, fc2_b, C, heads, hidden).to(torch.float32).reshape(P.SWIN_TAIL_SLOTS, 12, 64, 8)
    full = P.pack_swin_block_stream(torch.zeros(3 * C, C), None, proj_w, None, fc1_w, fc1_b, fc2_w, fc2_b, C, heads, hidden).to(torch.float32).reshape(48, 12, 64, 8)
    assert torch.equal(stl[6:], full[24:])  # the MLP slots are the block kernel's (fc1 / fc2 with their biases on the constant-one channels)
    scale = hd ** -0.5

    def frag(st, slot, f, i, k):
        return float(st[slot, f, 16 * (k // 8) + i, k % 8])

    rng = torch.Generator().manual_seed(6)
    for _ in range(400):
        p, c, w, i, k = (int(torch.randint(0, n, (1,), generator=rng)) for n in (3, 6, 4, 16, 32))
        hh, half = w >> 1, w & 1
        head, d, ch = 2 * p + hh, 16 * half + i, 32 * c + k
        for t in range(3):  # q, k, v tiles: all three biases ride on channels C, C + 1; the scale (no log2 e) is in the q rows
            sc = scale if t == 0 else 1.0
            if d < hd and ch < C:
                want = float(bf(qkv_w[t * C + head * hd + d, ch] * sc))
            elif d < hd and ch in (C, C + 1):
                b = qkv_b[t * C + head * hd + d] * sc
                want = float(bf(b)) if ch == C else float(bf(b - bf(b)))
            else:
                want = 0.0
            assert frag(sq, 6 * p + c, 3 * w + t, i, k) == want, (p, c, w, t, i, k)
        # projection slot c = head c of the attention output: fragment 3 w + n = output channels 48 w + 16 n + i, k = feature d of that head
        n = p
        row, dd = 48 * w + 16 * n + i, k
        want = float(bf(proj_w[row, c * hd + dd])) if (row < C and dd < hd) else 0.0
        assert frag(stl, c, 3 * w + n, i, k) == want, (c, w, n, i, k)



def test_index_form_packers_equal_the_value_packers():
    """studiosr_amd/fasttrain.py builds the training path's packed operands as index maps (one device gather per optimizer step); evaluated on
    the host they must reproduce studiosr_amd/packing.py element for element (where the layouts coincide)."""
    import numpy as np

    from studiosr_amd import fasttrain as F
    from studiosr_amd.models.hat import rpi_sa

    torch.manual_seed(0)

    class FakeFP:
        def __init__(self, tensors):
            self._off, n = {}, 0
            for p in tensors:
                self._off[id(p)] = n
                n += (p.numel() + 3) // 4 * 4
            self.P = np.zeros(n, np.float32)
            for p in tensors:
                self.P[self._off[id(p)]:self._off[id(p)] + p.numel()] = p.numpy().reshape(-1)

        def pidx(self, p):
            return (self._off[id(p)] + np.arange(p.numel(), dtype=np.int64)).reshape(tuple(p.shape))

    def emulate(fp, m, dtype=torch.bfloat16):  # what sr_tr_gather computes
        idx, scl, mode = m.idx.reshape(-1), m.scl.reshape(-1), m.mode.reshape(-1)
        t = torch.from_numpy(np.where(idx >= 0, fp.P[np.clip(idx, 0, None)] * scl, scl).astype(np.float32))
        hi = t.to(torch.bfloat16).float()
        return torch.where(torch.from_numpy(mode == 1), hi, torch.where(torch.from_numpy(mode == 2), t - hi, t)).to(dtype)

    qkv_w, qkv_b, proj_w = torch.randn(540, 180) * 0.1, torch.randn(540) * 0.1, torch.randn(180, 180) * 0.1
    fc1_w, fc1_b, fc2_w, fc2_b = torch.randn(360, 180) * 0.1, torch.randn(360) * 0.1, torch.randn(180, 360) * 0.1, torch.randn(180) * 0.1
    fp = FakeFP([qkv_w, qkv_b, proj_w, fc1_w, fc1_b, fc2_w, fc2_b])
    assert torch.equal(emulate(fp, F.pack_qkv_fwd(fp, qkv_w, qkv_b)), packing.pack_swin_qkv_stream(qkv_w, qkv_b, 180, 6))
    a = emulate(fp, F.pack_tail_fwd(fp, proj_w, fc1_w, fc1_b, fc2_w, fc2_b))
    b = packing.pack_swin_tail_stream(proj_w, fc1_w, fc1_b, fc2_w, fc2_b, 180, 6, 360)
    # the training stream differs by design in the hidden pad columns: no gelu(1) rows in fc1 (the kernel writes the ones), fc2 bias not divided by gelu(1)
    assert int((a != b).sum()) == 2 + 2 * 180
    w = torch.randn(60, 180, 3, 3)
    fp2 = FakeFP([w])
    assert torch.equal(emulate(fp2, F.pack_conv(fp2, w, 192, 64)), packing.pack_conv3x3(w, None, 192, packing.identity_idx(60, 64), torch.bfloat16)[0])
    wt = torch.zeros(180, 60, 3, 3)  # the data-gradient convolution's weight: transposed, taps flipped
    for t in range(9):
        wt[:, :, t // 3, t % 3] = w[:, :, (8 - t) // 3, (8 - t) % 3].t()
    assert torch.equal(emulate(fp2, F.pack_conv(fp2, w, 64, 192, transpose=True)), packing.pack_conv3x3(wt, None, 64, packing.identity_idx(180, 192), torch.bfloat16)[0])
    wu = torch.randn(256, 64, 3, 3)  # a conv feeding PixelShuffle(2): packed rows permuted
    fp4 = FakeFP([wu])
    rows = packing.pixel_shuffle_rows(64, 64, 2)
    assert torch.equal(emulate(fp4, F.pack_conv(fp4, wu, 64, 256, rows=rows.numpy())), packing.pack_conv3x3(wu, None, 64, rows, torch.bfloat16)[0])
    table = torch.randn(961, 6)
    fp3 = FakeFP([table])
    rpi = rpi_sa(16)
    ref = packing.gather_bias(table, rpi, 256, 256)
    b_, bt, bfr, b31 = F.pack_bias(fp3, table, rpi.numpy(), 256, 256)
    tiles = packing.bias_distinct_tiles(packing.gather_bias(table, rpi, 256, 256))  # the 31 distinct tiles of the LDS-form window attention (ABI v8)
    assert tiles is not None and b31 is not None and torch.equal(emulate(fp3, b31, torch.float32), tiles)
    assert packing.bias_distinct_tiles(torch.randn(F.HEADS, 256, 256)) is None
    from studiosr_amd.models.hat import rpi_oca  # the OCAB's bias as its rotated relative-position table (SrOcaAttn.bias_rel / SrTrAttnFwd.bias_rel, ABI v8)

    table_o = torch.randn(39 * 39, 6)
    fp5 = FakeFP([table_o])
    ro = rpi_oca(16, 0.5)
    bo = packing.gather_bias(table_o, ro, 256, 576)
    rel = packing.oca_bias_rel(bo)
    assert rel is not None and torch.equal(rel[:, packing.oca_rel_index()], bo)
    bo_im = F.pack_bias(fp5, table_o, ro.numpy(), 256, 576)[0]
    rel_im = F.pack_bias_rel(bo_im, ro.numpy(), 39 * 39)
    assert rel_im is not None and torch.equal(emulate(fp5, rel_im, torch.float32).reshape(6, 1521), rel)
    assert packing.oca_bias_rel(torch.randn(6, 256, 576)) is None
    assert torch.equal(emulate(fp3, b_, torch.float32).reshape(6, 256, 256), ref)
    assert torch.equal(emulate(fp3, bt, torch.float32).reshape(6, 256, 256), ref.transpose(1, 2))
    assert torch.equal(emulate(fp3, bfr, torch.float32), packing.bias_fragments(ref))


def test_export_routes_to_a_reference_loadable_checkpoint(tmp_path):
    """Model.export (common.py:86-98): the HIP forward cannot be traced to ONNX; format 'checkpoint' writes class / config / state_dict in the
    reference's keys (what the reference class needs to rebuild the model and run its own export), format 'onnx' refuses loudly."""
    import studiosr_amd as S

    m = S.EDSR(scale=2, n_feats=16, n_resblocks=2)
    with pytest.raises(NotImplementedError):
        m.export(str(tmp_path / "m.onnx"))
    with pytest.raises(ValueError):
        m.export(format="tflite")
    path = m.export(str(tmp_path / "m.pth"), format="checkpoint")
    blob = torch.load(path, weights_only=True)
    assert blob["class"] == "EDSR" and blob["config"] == m.get_model_config()
    m2 = S.EDSR(**{k: v for k, v in blob["config"].items() if k in ("scale", "n_feats", "n_resblocks", "n_colors", "res_scale", "img_range")})
    m2.load_state_dict(blob["state_dict"])
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_finalize_items_in_arena_order_are_the_same_sums():
    """fasttrain.FinalMap.finish() turns the per-parameter maps (grad[p] = scale[p] * sum_s part[src[p] + s * stride[p]]) into work items sorted by src with a
    destination index (sr_tr_finalize_to, ABI v9): emulate both forms on random partials -- every parameter is written once, with the same sum in the same order;
    parameters without a source get a zero, parameters of another map are not touched."""
    from studiosr_amd import fasttrain as FT

    rng = np.random.default_rng(0)
    fm = FT.FinalMap.__new__(FT.FinalMap)
    n = 500
    fm.p0, fm.p1 = 40, 40 + n
    fm.src = np.full(n, -1, dtype=np.int64)
    fm.stride = np.zeros(n, dtype=np.int32)
    fm.ns = np.ones(n, dtype=np.int32)
    fm.scale = np.ones(n, dtype=np.float32)
    fm.size = 0
    a = fm.alloc(9 * 20 * 3)  # a "conv" job: 3 slices of [9 taps][20]; parameters (c, tap) -> tap-major packed order
    pidx = 40 + np.arange(180)
    fm.put(pidx, a + (np.arange(180) % 9) * 20 + np.arange(180) // 9, 180, 3, scale=0.5)
    b = fm.alloc(100 * 4)     # a "linear" job: 4 slices of 100
    fm.put(40 + 200 + np.arange(100), b + np.arange(100), 100, 4)
    c = fm.alloc(7 * 64)      # "LayerNorm partials": 64 slices (one per workgroup) of 7 -- items of the eight-lanes-per-element launch
    fm.put(40 + 300 + np.arange(7), c + np.arange(7), 7, 64)
    fm.src[350:360] = -2      # another map's elements
    fm.finish("cpu")
    part = rng.standard_normal(fm.size).astype(np.float32)
    want = np.full(n, np.nan, dtype=np.float32)
    for p in range(n):
        if fm.src[p] == -2:
            continue
        acc = np.float32(0)
        if fm.src[p] >= 0:
            for s in range(fm.ns[p]):
                acc = np.float32(acc + part[fm.src[p] + s * fm.stride[p]])
        want[p] = acc * fm.scale[p]
    got = np.full(n, np.nan, dtype=np.float32)
    seen = []
    assert sorted(lanes for _, lanes, *_ in fm.sets) == [1, 8] and sum(cnt for cnt, *_ in fm.sets) == fm.n_items == n - 10
    for cnt, lanes, *arrs in fm.sets:
        src, dst, st, ns, sc = (t.numpy() for t in arrs)
        assert len(src) == cnt and np.all(np.diff(src) >= 0)
        assert np.all(ns >= 64) if lanes == 8 else np.all(ns < 64)
        seen += dst.tolist()
        for i in range(cnt):
            if lanes == 8:  # lane j adds slices j, j + 8, ...; the eight sums meet in an xor tree (1, 2, 4)
                lane = [np.float32(0)] * 8
                for s in range(ns[i]):
                    lane[s % 8] = np.float32(lane[s % 8] + part[src[i] + s * st[i]])
                for m in (1, 2, 4):
                    lane = [np.float32(lane[j] + lane[j ^ m]) for j in range(8)]
                acc = lane[0]
            else:
                acc = np.float32(0)
                for s in range(ns[i]):
                    acc = np.float32(acc + part[src[i] + s * st[i]])
            got[dst[i]] = acc * sc[i]
    assert len(set(seen)) == fm.n_items
    long_p = 300 + np.arange(7)  # (indices relative to p0)
    short = np.ones(n, dtype=bool)
    short[long_p] = False
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[short & ~np.isnan(got)], want[short & ~np.isnan(want)])
    np.testing.assert_allclose(got[long_p], want[long_p], rtol=1e-5)  # (another summation order)
    assert np.all(got[180:200] == 0) and np.isnan(got[350:360]).all()


def test_xcd_remap_is_a_bijection_with_contiguous_classes():
    """The block-id remap of the XCD-aware kernels (sr_conv3x3, sr_conv_big, sr_rcab, sr_oca_lds, the CAB tiles of sr_hab_mid; cdna guide T1, bijective form): every
    work item exactly once, and the block ids of one residue class mod 8 (= the blocks that share an XCD) take a contiguous range of items."""
    def remap(t, nwg):
        q, r, xcd = nwg >> 3, nwg & 7, t & 7
        return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + (t >> 3)

    for nwg in (1, 7, 8, 9, 55, 220, 240, 648, 1000, 1296):
        items = [remap(t, nwg) for t in range(nwg)]
        assert sorted(items) == list(range(nwg))
        for c in range(min(8, nwg)):
            mine = [items[t] for t in range(c, nwg, 8)]
            assert mine == list(range(mine[0], mine[0] + len(mine)))


def test_launch_plan_recorder_classifies_calls_and_plans_validate():
    """Launch plans (ABI v10, csrc/sr_plan.cpp; studiosr_amd/_lib.py): the recorder turns launches made through lib() into plan operations -- argument-block
    calls become C operations (their blocks are COPIED by sr_plan_create), positional calls stay Python closures, streams become slots -- without enqueueing
    anything (no GPU here: nothing may be launched).  sr_plan_create validates; a plan reports its operation and stream counts."""
    import ctypes as C

    import __graft_entry__ as G

    if not os.path.exists(L.LIB_PATH):
        G.build()
    h = L.lib()
    rec = L.PlanRecorder(main_stream=0x1000)
    with L.recording(rec):
        lib = L.lib()
        assert lib is not h and lib.sr_abi_version() == L.ABI_VERSION  # queries pass through
        a = L.SrTrGelu()
        a.n = 64
        assert lib.sr_tr_gelu_args(C.byref(a), 0x1000) == 0            # CALL1 on the main stream (slot 0)
        e = rec.event()
        rec.add_event_record(0x1000, e)
        rec.add_stream_wait(0x2000, e)
        w, c = L.SrWindowAttn(), L.SrCab()
        assert lib.sr_hab_mid(C.byref(w), C.byref(c), 0x2000) == 0     # CALL2 on a side stream (slot 1)
        jobs = (L.SrTrWgradJob * 2)()
        assert lib.sr_tr_wgrad(jobs, 2, 0x1000) == 0                   # CALLI
        assert lib.sr_tr_add(1, 2, 0, 3, 64, 0x1000) == 0              # positional: a Python closure
    assert L.lib() is h and L.recorder() is None
    kinds = [it[1].kind if it[0] == "op" else "py" for it in rec.items]
    assert kinds == [L.PLAN_CALL1, L.PLAN_EVENT_RECORD, L.PLAN_STREAM_WAIT, L.PLAN_CALL2, L.PLAN_CALLI, "py"]
    assert [it[1].stream for it in rec.items if it[0] == "op"] == [0, 0, 1, 1, 0] and rec.side == [0x2000] and rec.n_launches == 4
    a.n = 0  # the plan owns copies of its argument blocks
    plan = rec.finish()
    assert [k for k, _ in plan.segments] == ["c", "py"]
    seg = plan.segments[0][1]
    assert h.sr_plan_ops(seg) == 5 and h.sr_plan_streams(seg) == 2
    # validation
    bad = (L.SrPlanOp * 1)()
    bad[0].kind = L.PLAN_STREAM_WAIT
    bad[0].ival = 3
    assert not h.sr_plan_create(bad, 1, 1) and b"bad operation" in h.sr_last_error()
    assert h.sr_plan_run(seg, (C.c_void_p * 1)(), 1) != 0 and b"streams" in h.sr_last_error()  # fewer streams than the plan uses: refused before anything is enqueued


def test_every_environment_switch_is_registered():
    """studiosr_amd/knobs.py lists every SR_* variable the package reads (Python and C): a new switch must be documented there, a removed one deleted."""
    import glob

    from studiosr_amd.knobs import KNOBS

    used = set()
    for pat in ("studiosr_amd/**/*.py", "studiosr_amd/csrc/*.hip", "studiosr_amd/csrc/*.h", "studiosr_amd/csrc/*.cpp"):
        for f in glob.glob(os.path.join(ROOT, pat), recursive=True):
            if f.endswith("knobs.py"):
                continue
            used |= set(re.findall(r'(?:knob\(|environ\.get\(|environ\[|getenv\()\s*"(SR_[A-Z0-9_]+)"', open(f).read()))
    assert used == set(KNOBS), (sorted(used - set(KNOBS)), sorted(set(KNOBS) - used))
    assert all(k in ("select", "tune", "diag") and doc for _, k, doc in KNOBS.values())
