"""Per-step training metrics on the device (reference: ``src/utils/metric.py:5-36`` +
``src/utils/utils.py:32-40``).  HD / ASD (medpy) are evaluation-only and out of scope."""
from __future__ import annotations

import torch

from .. import kernels as K


def dice_coef_multilabel(y_true_onehot_u8: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """mean over labels 1..C-1 of (2|A.B|+1)/(|A|+|B|+1), with B = soft_to_hard_pred(logits)
    computed on the fly; returns a 0-dim device tensor (no host sync)."""
    return K.dice_metric(logits, y_true_onehot_u8)


def argmax_labels(x: torch.Tensor) -> torch.Tensor:
    """``np.argmax(soft_to_hard_pred(x, 1), axis=1)`` (``train_mscmrseg.py:85-87``) on the device: uint8 label map,
    first channel holding the per-pixel maximum.  Accepts fp32 logits or the uint8 one-hot ground truth."""
    return K.argmax_labels(x)


def label_dice(pred_labels: torch.Tensor, gt_labels: torch.Tensor, num_classes: int) -> torch.Tensor:
    """per-class Dice ``2|A.B|/(|A|+|B|)`` (0 when both are empty) of two label maps -- what ``evaluate``
    (``metric.py:39-82``) gets from ``medpy.metric.binary.dc`` for classes 1..3.  fp32 ``[num_classes]``."""
    return K.label_dice(pred_labels, gt_labels, num_classes)
