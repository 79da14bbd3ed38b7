"""PSNR / SSIM of the reference's evaluation protocol in plain numpy + scipy (studiosr/utils/metrics.py:7-70), without
skimage / cv2 (neither ships on the MI355X image).  Host-side bookkeeping, not a GPU path.

  to_y          BT.601 luma in [16, 235]: dot(rgb / 255, [65.481, 128.553, 24.966]) + 16        (metrics.py:11-17)
  compute_psnr  trim to the common size, crop `crop_border`, optional Y, fp32 MSE, 20 log10(255/sqrt(mse)) (:36-49)
  compute_ssim  the same pre-processing, then skimage.metrics.structural_similarity(K1=.01, K2=.03, gaussian_weights=True,
                sigma=1.5, use_sample_covariance=False, data_range=255) (:52-70): Gaussian window truncated at 3.5 sigma
                (11 x 11), scipy.ndimage.gaussian_filter(mode="reflect") for the local moments, population covariance,
                mean over the image minus a 5-pixel border, averaged over channels.  skimage is absent here, so this is a
                restatement of its published algorithm (scikit-image >= 0.19, the reference's un-pinned dependency).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def is_rgb(im: np.ndarray) -> bool:
    return im.ndim == 3 and im.shape[-1] == 3


def to_y(image: np.ndarray) -> np.ndarray:
    if not is_rgb(image):
        return image
    if image.dtype == np.uint8:
        image = image.astype(np.float32) / 255.0
    return np.dot(image, [65.481, 128.553, 24.966]) + 16.0


def crop_img_to_equal(im1: np.ndarray, im2: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    h, w = min(im1.shape[0], im2.shape[0]), min(im1.shape[1], im2.shape[1])
    return im1[:h, :w], im2[:h, :w]


def _prepare(im1: np.ndarray, im2: np.ndarray, y_only: bool, crop_border: int):
    im1, im2 = crop_img_to_equal(im1, im2)
    if crop_border:
        im1 = im1[crop_border:-crop_border, crop_border:-crop_border]
        im2 = im2[crop_border:-crop_border, crop_border:-crop_border]
    if y_only:
        im1, im2 = to_y(im1), to_y(im2)
    return im1, im2


def compute_psnr(im1: np.ndarray, im2: np.ndarray, y_only: bool = False, crop_border: int = 0) -> float:
    im1, im2 = _prepare(im1, im2, y_only, crop_border)
    if not y_only and im1.dtype != np.uint8:
        im1, im2 = im1 * 255.0, im2 * 255.0
    error = np.mean((im1.astype(np.float32) - im2.astype(np.float32)) ** 2)
    if error == 0:
        return np.inf
    return 20 * np.log10(255.0 / np.sqrt(error))


def _ssim_plane(x: np.ndarray, y: np.ndarray, data_range: float, K1: float, K2: float, sigma: float) -> float:
    from scipy.ndimage import gaussian_filter

    truncate = 3.5
    r = int(truncate * sigma + 0.5)  # window radius 5 -> 11 x 11
    if min(x.shape) < 2 * r + 1:
        raise ValueError("win_size exceeds image extent")
    x, y = x.astype(np.float64), y.astype(np.float64)
    f = lambda a: gaussian_filter(a, sigma, mode="reflect", truncate=truncate)  # noqa: E731
    ux, uy = f(x), f(y)
    vx, vy, vxy = f(x * x) - ux * ux, f(y * y) - uy * uy, f(x * y) - ux * uy  # population (co)variances
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    return float(S[r:-r, r:-r].mean(dtype=np.float64))  # filter edge effects are excluded


def compute_ssim(im1: np.ndarray, im2: np.ndarray, y_only: bool = False, crop_border: int = 0) -> float:
    im1, im2 = _prepare(im1, im2, y_only, crop_border)
    kw = dict(data_range=255.0, K1=0.01, K2=0.03, sigma=1.5)
    if is_rgb(im1):
        return float(np.mean([_ssim_plane(im1[..., c], im2[..., c], **kw) for c in range(3)]))
    return _ssim_plane(im1, im2, **kw)
