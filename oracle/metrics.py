"""Per-step host metrics of the reference, restated in numpy (oracle; test-only).

Reference: utils/utils.py:32-40 (soft_to_hard_pred), utils/metric.py:5-36 (dice_coef,
dice_coef_multilabel), train_mscmrseg.py:270-273 (discriminator accuracy).
"""
from __future__ import annotations

import numpy as np


def soft_to_hard_pred(pred: np.ndarray, channel_axis: int = 1) -> np.ndarray:
    """1 where a channel equals the per-pixel max (ties mark several channels), else 0."""
    return (pred == pred.max(axis=channel_axis, keepdims=True)).astype(np.int64)


def dice_coef(y_true: np.ndarray, y_pred: np.ndarray) -> float:
    """(2*|A.B| + 1) / (|A| + |B| + 1) on flattened arrays (metric.py:5-14)."""
    t = y_true.reshape(-1).astype(np.float64)
    q = y_pred.reshape(-1).astype(np.float64)
    return float((2.0 * np.sum(t * q) + 1.0) / (np.sum(t) + np.sum(q) + 1.0))


def dice_coef_multilabel(y_true: np.ndarray, y_pred: np.ndarray, num_labels: int = 4) -> float:
    """Mean Dice over labels 1..num_labels-1, channel-first inputs [B,C,H,W] (metric.py:17-36)."""
    vals = [dice_coef(y_true[:, c], y_pred[:, c]) for c in range(1, num_labels)]
    return float(sum(vals) / (num_labels - 1))


def disc_accuracy(d_out_logits: np.ndarray, source: bool) -> float:
    """train_mscmrseg.py:270-273 / :299-302: mean(sigmoid(D) >= .5), or 1 - that for target."""
    hit = (1.0 / (1.0 + np.exp(-d_out_logits.astype(np.float64))) >= 0.5).mean()
    return float(hit if source else 1.0 - hit)
