"""Evaluation harness with the reference's `Evaluator` surface (studiosr/engine/evaluator.py:11-79): any
`func(np.uint8[H,W,3]) -> np.uint8[sH,sW,3]` -- `model.inference`, `model.inference_with_self_ensemble` -- is scored over
a directory of `GTmod12/` + `LRbicx{scale}/` image pairs with Y-channel PSNR / SSIM and a `scale`-pixel border crop.

Differences, all forced by the MI355X image: images are read with PIL instead of cv2 (same RGB uint8 arrays for PNG/BMP),
datasets are never downloaded (no network: a missing directory is an error that names the path), there is no
visualisation.  `PairedImageDataset` is the evaluation half of studiosr/data/dataset.py:14-78 (no crops / augmentation).
"""
from __future__ import annotations

import os
from typing import Callable, List, Tuple

import numpy as np

from .metrics import compute_psnr, compute_ssim

IMAGE_EXTENSIONS = [".bmp", ".jpeg", ".jpg", ".jpe", ".jp2", ".png", ".webp", ".tiff", ".tif"]  # helpers.py:95-96


def imread(path: str) -> np.ndarray:
    """RGB uint8 [H, W, 3] (helpers.py:40-43: cv2.IMREAD_COLOR + BGR2RGB)."""
    from PIL import Image

    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def imwrite(path: str, image: np.ndarray) -> bool:
    from PIL import Image

    Image.fromarray(np.asarray(image, dtype=np.uint8), "RGB").save(path)
    return True


def get_image_files(root: str) -> List[str]:
    """Sorted file NAMES under root (helpers.py:99-106)."""
    out = []
    for _, _, files in os.walk(root):
        out += [f for f in files if os.path.splitext(f)[1].lower() in IMAGE_EXTENSIONS]
    return sorted(out)


class PairedImageDataset:
    def __init__(self, gt_path: str, lq_path: str) -> None:
        self.gt_path, self.lq_path = gt_path, lq_path
        self.files = get_image_files(gt_path)

    def __len__(self) -> int:
        return len(self.files)

    def __getitem__(self, idx: int) -> Tuple[np.ndarray, np.ndarray]:
        if idx >= len(self.files):
            raise IndexError(idx)
        f = self.files[idx]
        return imread(os.path.join(self.lq_path, f)), imread(os.path.join(self.gt_path, f))


class Evaluator:
    def __init__(self, dataset: str = "DIV2K_mini", scale: int = 4, root: str = "dataset") -> None:
        self.dataset, self.scale, self.root = dataset, scale, root
        path = os.path.join(root, dataset)
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path} not found (datasets are not downloaded here; unpack {dataset} there)")
        gt_mod = 12 if scale in [2, 3, 4] else scale
        self.testset = PairedImageDataset(os.path.join(path, f"GTmod{gt_mod}"), os.path.join(path, f"LRbicx{scale}"))

    def __call__(self, func: Callable[[np.ndarray], np.ndarray], y_only: bool = True, visualize: bool = False, logging: bool = True) -> Tuple[float, float]:
        psnr, ssim = self.run(func, y_only, visualize, logging)
        print(f" {self.dataset:>8} - Average PSNR: {psnr:6.3f}, SSIM: {ssim:6.4f}")
        return psnr, ssim

    def run(self, func: Callable[[np.ndarray], np.ndarray], y_only: bool = True, visualize: bool = False, logging: bool = False) -> Tuple[float, float]:
        crop_border = self.scale
        psnrs, ssims = [], []
        for i, (lq, gt) in enumerate(self.testset):
            sr = func(lq)
            psnrs.append(compute_psnr(sr, gt, crop_border=crop_border, y_only=y_only))
            ssims.append(compute_ssim(sr, gt, crop_border=crop_border, y_only=y_only))
            if logging:
                print(f" {self.dataset:>8} - {i + 1:>3}/{len(self.testset):>3} PSNR: {psnrs[-1]:6.3f}, SSIM: {ssims[-1]:6.4f}", end="\r")
            if visualize:
                self._save_comparison(i, lq, sr, gt)
        return float(np.mean(psnrs)), float(np.mean(ssims))

    def _save_comparison(self, i: int, lq: np.ndarray, sr: np.ndarray, gt: np.ndarray) -> str:
        """The reference shows nearest | bicubic | SR | GT side by side in a window (evaluator.py:69-72, cv2 + helpers.compare); there is
        no display (or cv2) on an MI355X node, so the same strip is written to `<root>/<dataset>/visualize_x<scale>/<index>.png` (PIL resizes)."""
        from PIL import Image

        h, w = gt.shape[:2]
        lq_img = Image.fromarray(lq)
        tiles = [np.asarray(lq_img.resize((w, h), Image.NEAREST)), np.asarray(lq_img.resize((w, h), Image.BICUBIC)), sr[:h, :w], gt]
        out_dir = os.path.join(self.root, self.dataset, f"visualize_x{self.scale}")
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, f"{i:04d}.png")
        Image.fromarray(np.concatenate([np.ascontiguousarray(t[..., :3]) for t in tiles], axis=1)).save(path)
        return path

    @staticmethod
    def benchmark(func: Callable[[np.ndarray], np.ndarray], scale: int = 4, y_only: bool = True,
                  datasets: List[str] = ["Set5", "Set14", "BSD100", "Urban100", "Manga109"], root: str = "dataset") -> Tuple[List[float], List[float]]:
        rows = {"Metric": "| Metric |", "line": "| ------ |", "psnr": "|   PSNR |", "ssim": "|   SSIM |"}
        psnr_list, ssim_list = [], []
        for dataset in datasets:
            psnr, ssim = Evaluator(dataset, scale, root).run(func, y_only, logging=True)
            rows["Metric"] += " %10s |" % dataset
            rows["line"] += " ---------- |"
            rows["psnr"] += " %10.3f |" % psnr
            rows["ssim"] += " %10.4f |" % ssim
            psnr_list.append(psnr)
            ssim_list.append(ssim)
        print("\n".join(rows.values()) + "\n")
        return psnr_list, ssim_list
