"""numpy restatement of the reference's ``train_transform`` (deadtrees/data/deadtreedata.py:128-146) with the
random draws made explicit.  TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED for the brightness/contrast step: albumentations is not installed here (setup.py pins none), so
``RandomBrightnessContrast`` on uint8 is restated from its published algorithm (a 256-entry LUT
``clip(arange(256) * alpha + beta * mean(img), 0, 255).astype(uint8)``, ``brightness_by_max=False``).
The geometric steps are numpy calls albumentations itself uses (``img[:, ::-1]``, ``img[::-1]``, ``np.rot90``)
and Normalize is ``(img - mean*255) / (std*255)`` in float32 (already pinned by val_transform's tests).
"""
from __future__ import annotations

import numpy as np


def geometric(arr: np.ndarray, flip: int, rot: int) -> np.ndarray:
    """HW... array: OneOf(HorizontalFlip, VerticalFlip) then RandomRotate90 with factor ``rot``."""
    if flip == 1:
        arr = arr[:, ::-1]
    elif flip == 2:
        arr = arr[::-1]
    return np.ascontiguousarray(np.rot90(arr, rot))


def brightness_contrast_u8(img: np.ndarray, alpha: float, beta: float) -> np.ndarray:
    if alpha == 1.0 and beta == 0.0:
        return img
    lut = np.arange(0, 256, dtype=np.float32)
    lut = lut * np.float32(alpha)
    if beta != 0:
        lut = lut + np.float32(np.float64(beta) * np.mean(img))
    return np.clip(lut, 0, 255).astype(np.uint8)[img]


def train_transform(img_u8_hwc: np.ndarray, flip: int, rot: int, alpha: float, beta: float, mean, std, c_dst: int):
    """-> float32 HWC (first ``c_dst`` channels), the image side of the reference pipeline before ToTensorV2"""
    img = geometric(img_u8_hwc, flip, rot)
    img = brightness_contrast_u8(img, alpha, beta)
    m = np.asarray(mean[:c_dst], np.float32) * 255.0
    inv = 1.0 / (np.asarray(std[:c_dst], np.float32) * 255.0)
    return (img[..., :c_dst].astype(np.float32) - m) * inv
