"""Host-side evaluation protocol: PSNR pinned by reference-generated vectors and by the
reference's own test cases (tests/utils/test_compute_psnr.py: 0 dB, inf, dtype invariance); SSIM (skimage is absent, so
no reference output can be generated: parity unpinned) checked against a brute-force windowed evaluation of the published
formula; the Evaluator surface (studiosr/engine/evaluator.py:11-79) on a temporary GTmod12/LRbicx4 directory."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import metrics as OMT
from studiosr_amd import Evaluator, compute_psnr, compute_ssim
from studiosr_amd import metrics as M
from studiosr_amd.evaluator import PairedImageDataset, imread, imwrite

SHAPES = [(16, 16, 3), (32, 32, 3), (64, 64, 3)]


def test_psnr_and_luma_match_the_reference_vectors():
    g = load_golden("f14_psnr")
    a, b = g["a"], g["b"]
    assert compute_psnr(a, b) == float(g["psnr_rgb"])
    assert compute_psnr(a, b, y_only=True, crop_border=4) == float(g["psnr_y_crop4"])
    assert compute_psnr(a / 255.0, b / 255.0) == float(g["psnr_float"])
    assert np.array_equal(M.to_y(a), g["y"])
    for kw in (dict(), dict(y_only=True, crop_border=4)):  # product and oracle restatements agree bit for bit
        assert compute_psnr(a, b, **kw) == OMT.compute_psnr(a, b, **kw)


@pytest.mark.parametrize("shape", SHAPES)
def test_psnr_reference_test_cases(shape):
    rng = np.random.default_rng(0)
    assert compute_psnr(np.zeros(shape, np.uint8), np.full(shape, 255, np.uint8)) == 0.0
    a = rng.integers(0, 256, size=shape, dtype=np.uint8)
    assert np.isinf(compute_psnr(a, a.copy()))
    b = rng.integers(0, 256, size=shape, dtype=np.uint8)
    assert abs(compute_psnr(a, b) - compute_psnr(a / 255.0, b / 255.0)) < 1e-12
    big = rng.integers(0, 256, size=(shape[0] + 3, shape[1] + 2, 3), dtype=np.uint8)  # unequal sizes are trimmed bottom/right
    assert compute_psnr(big, big[: shape[0], : shape[1]]) == np.inf


def _ssim_brute(x, y, data_range=255.0, K1=0.01, K2=0.03, sigma=1.5, r=5):
    """Direct evaluation of the SSIM definition with an explicit normalised 11x11 Gaussian window (valid positions only)."""
    k = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    w = np.outer(k, k)
    w /= w.sum()
    x, y = x.astype(np.float64), y.astype(np.float64)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    vals = []
    for i in range(r, x.shape[0] - r):
        for j in range(r, x.shape[1] - r):
            px, py = x[i - r : i + r + 1, j - r : j + r + 1], y[i - r : i + r + 1, j - r : j + r + 1]
            ux, uy = (w * px).sum(), (w * py).sum()
            vx, vy, vxy = (w * px * px).sum() - ux * ux, (w * py * py).sum() - uy * uy, (w * px * py).sum() - ux * uy
            vals.append((2 * ux * uy + C1) * (2 * vxy + C2) / ((ux * ux + uy * uy + C1) * (vx + vy + C2)))
    return float(np.mean(vals))


def test_ssim_matches_the_windowed_definition_and_its_invariants():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, size=(24, 28, 3), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-40, 41, size=a.shape), 0, 255).astype(np.uint8)
    assert compute_ssim(a, a) == 1.0
    assert abs(compute_ssim(a, b) - compute_ssim(b, a)) < 1e-15
    want = np.mean([_ssim_brute(a[..., c], b[..., c]) for c in range(3)])
    assert abs(compute_ssim(a, b) - want) < 1e-9
    ya, yb = M.to_y(a)[2:-2, 2:-2], M.to_y(b)[2:-2, 2:-2]
    assert abs(compute_ssim(a, b, y_only=True, crop_border=2) - _ssim_brute(ya, yb)) < 1e-9
    assert 0.0 < compute_ssim(a, b) < 1.0
    with pytest.raises(ValueError):
        compute_ssim(a[:8, :8], b[:8, :8])  # smaller than the 11 x 11 window, as skimage


def test_evaluator_over_a_directory_of_pairs(tmp_path, capsys):
    rng = np.random.default_rng(2)
    root = tmp_path / "dataset"
    gt_dir, lq_dir = root / "Toy" / "GTmod12", root / "Toy" / "LRbicx4"
    gt_dir.mkdir(parents=True)
    lq_dir.mkdir(parents=True)
    pairs = {}
    for name, (h, w) in {"b.png": (12, 9), "a.png": (9, 12)}.items():
        lq = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        gt = np.clip(np.kron(lq, np.ones((4, 4, 1))) + rng.integers(-9, 10, size=(4 * h, 4 * w, 3)), 0, 255).astype(np.uint8)
        imwrite(str(lq_dir / name), lq)
        imwrite(str(gt_dir / name), gt)
        pairs[name] = (lq, gt)
    assert np.array_equal(imread(str(lq_dir / "a.png")), pairs["a.png"][0])  # PNG round trip is lossless RGB

    ds = PairedImageDataset(str(gt_dir), str(lq_dir))
    assert ds.files == ["a.png", "b.png"] and len(ds) == 2
    func = lambda lq: np.kron(lq, np.ones((4, 4, 1), dtype=np.uint8))  # noqa: E731  nearest-neighbour "model"
    ev = Evaluator("Toy", scale=4, root=str(root))
    psnr, ssim = ev(func)
    want_p = np.mean([compute_psnr(func(lq), gt, y_only=True, crop_border=4) for lq, gt in (pairs["a.png"], pairs["b.png"])])
    want_s = np.mean([compute_ssim(func(lq), gt, y_only=True, crop_border=4) for lq, gt in (pairs["a.png"], pairs["b.png"])])
    assert psnr == pytest.approx(want_p) and ssim == pytest.approx(want_s)
    assert "Average PSNR" in capsys.readouterr().out
    ps, ss = Evaluator.benchmark(func, scale=4, datasets=["Toy"], root=str(root))
    assert ps == [pytest.approx(want_p)] and ss == [pytest.approx(want_s)]
    # visualize=True (evaluator.py:69-72): the nearest | bicubic | SR | GT strip is written next to the dataset instead of shown in a window
    p2, s2 = ev.run(func, visualize=True)
    assert p2 == pytest.approx(want_p) and s2 == pytest.approx(want_s)
    strip = imread(str(root / "Toy" / "visualize_x4" / "0000.png"))
    lq0, gt0 = pairs["a.png"]
    assert strip.shape == (gt0.shape[0], 4 * gt0.shape[1], 3)
    assert np.array_equal(strip[:, :gt0.shape[1]], func(lq0)) and np.array_equal(strip[:, 2 * gt0.shape[1]:3 * gt0.shape[1]], func(lq0)) and np.array_equal(strip[:, 3 * gt0.shape[1]:], gt0)
    with pytest.raises(FileNotFoundError):
        Evaluator("Set5", scale=4, root=str(root))
