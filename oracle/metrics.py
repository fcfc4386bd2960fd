"""Per-step host metrics of the reference, restated in numpy (oracle; test-only).

Reference: utils/utils.py:32-40 (soft_to_hard_pred), utils/metric.py:5-36 (dice_coef,
dice_coef_multilabel), train_mscmrseg.py:270-273 (discriminator accuracy), and for the validation loop
train_mscmrseg.py:85-92 (label maps) + utils/metric.py:39-82 (evaluate: per-class medpy dc).
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


def argmax_labels(pred: np.ndarray) -> np.ndarray:
    """train_mscmrseg.py:85-87: soft_to_hard_pred -> move channels last -> argmax = first channel at the max."""
    return np.argmax(np.moveaxis(soft_to_hard_pred(pred, 1), 1, -1), axis=-1).astype(np.uint8)


def binary_dc(result: np.ndarray, reference: np.ndarray) -> float:
    """medpy.metric.binary.dc (medpy is a third-party dependency absent here, version unpinned by the reference;
    its published definition): 2|A.B| / (|A| + |B|) on boolean arrays, 0.0 when both are empty."""
    a, b = result.astype(bool), reference.astype(bool)
    den = float(np.count_nonzero(a) + np.count_nonzero(b))
    return float(2.0 * np.count_nonzero(a & b) / den) if den > 0 else 0.0


def label_dice(pred_labels: np.ndarray, gt_labels: np.ndarray, num_classes: int) -> np.ndarray:
    """per-class Dice the way evaluate() (metric.py:57-73) computes it: binarise each label, then dc."""
    return np.array([binary_dc(pred_labels == c, gt_labels == c) for c in range(num_classes)], dtype=np.float64)
