"""Synthetic batches in the reference's batch layout (oracle side; numpy only).

Layout contract (data_generator_mscmrseg.py:274-319): images float32
[B,Cin,H,W] in [0,1); masks one-hot uint8 [B,C,H,W] (utils/utils.py:25-29); vertices
float32 [B,300,3] = sampler(mask)/255 (data_generator_mscmrseg.py:317).  Masks are nested
ellipses with a seeded centre jitter so every label occurs and the foreground exceeds
50 px (npy2point.py:116 takes the sampling branch).
"""
from __future__ import annotations

import numpy as np

from .sampler import mask_to_pointcloud


def synth_labels(b: int, n_class: int, hw: int, rng: np.random.Generator) -> np.ndarray:
    """integer label maps [B,H,W]: label k = k-th nested ellipse (0 = background)."""
    yy, xx = np.mgrid[0:hw, 0:hw].astype(np.float64)
    lab = np.zeros((b, hw, hw), dtype=np.int64)
    for i in range(b):
        cy = hw * (0.5 + 0.08 * (rng.random() - 0.5))
        cx = hw * (0.5 + 0.08 * (rng.random() - 0.5))
        for k in range(1, n_class):
            ry = hw * 0.36 * (n_class - k) / (n_class - 1)
            rx = hw * 0.28 * (n_class - k) / (n_class - 1)
            lab[i][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = k
    return lab


def onehot_u8(lab: np.ndarray, n_class: int) -> np.ndarray:
    return np.ascontiguousarray(np.moveaxis(np.eye(n_class, dtype=np.uint8)[lab], -1, 1))


def synth_batch(b: int, in_channels: int, n_class: int, hw: int, seed: int, gaussian: bool = False):
    """-> (imgA f32, maskA u8 one-hot, vertA f32, imgB f32, vertB f32)."""
    rng = np.random.default_rng(seed)
    draw = (lambda s: rng.normal(0, 1, s).astype(np.float32)) if gaussian else \
           (lambda s: rng.random(s, dtype=np.float32))
    img_a = draw((b, in_channels, hw, hw))
    img_b = draw((b, in_channels, hw, hw))
    lab_a = synth_labels(b, n_class, hw, rng)
    lab_b = synth_labels(b, n_class, hw, rng)
    firsts = rng.integers(0, 1 << 30, size=(2, b))

    def verts(lab, fr):
        v = np.stack([mask_to_pointcloud(lab[i][..., None], 300, first=int(fr[i])) for i in range(b)])
        return (v / 255.0).astype(np.float32)

    return img_a, onehot_u8(lab_a, n_class), verts(lab_a, firsts[0]), img_b, verts(lab_b, firsts[1])
