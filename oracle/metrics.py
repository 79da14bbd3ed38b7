"""Oracle PSNR (TEST INFRASTRUCTURE, see oracle/__init__.py).

numpy restatement of studiosr/utils/metrics.py:11-49 (BT.601 luma, border crop,
fp32 MSE, 20*log10(255/sqrt(mse))).  Used to report the metric's "PSNR delta".
"""
from __future__ import annotations

import numpy as np


def to_y(img: np.ndarray) -> np.ndarray:
    """RGB -> Y in [16, 235] (studiosr/utils/metrics.py:11-17)."""
    if img.ndim != 3 or img.shape[-1] != 3:
        return img
    if img.dtype == np.uint8:
        img = img.astype(np.float32) / 255.0
    return np.dot(img, [65.481, 128.553, 24.966]) + 16.0


def _crop_equal(a: np.ndarray, b: np.ndarray):
    """Trim the larger image at the bottom/right (studiosr/utils/metrics.py:20-33)."""
    h, w = min(a.shape[0], b.shape[0]), min(a.shape[1], b.shape[1])
    return a[:h, :w], b[:h, :w]


def compute_psnr(a: np.ndarray, b: np.ndarray, y_only: bool = False, crop_border: int = 0) -> float:
    """studiosr/utils/metrics.py:36-49."""
    a, b = _crop_equal(a, b)
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b = b[crop_border:-crop_border, crop_border:-crop_border]
    if y_only:
        a, b = to_y(a), to_y(b)
    elif a.dtype != np.uint8:
        a, b = a * 255.0, b * 255.0
    err = np.mean((a.astype(np.float32) - b.astype(np.float32)) ** 2)
    if err == 0:
        return np.inf
    return 20 * np.log10(255.0 / np.sqrt(err))
