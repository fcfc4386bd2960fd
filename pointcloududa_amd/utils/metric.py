"""Per-step training metrics on the device (reference: ``src/utils/metric.py:5-36`` +
``src/utils/utils.py:32-40``).  HD / ASD (medpy) are evaluation-only and out of scope."""
from __future__ import annotations

import torch

from .. import kernels as K


def dice_coef_multilabel(y_true_onehot_u8: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """mean over labels 1..C-1 of (2|A.B|+1)/(|A|+|B|+1), with B = soft_to_hard_pred(logits)
    computed on the fly; returns a 0-dim device tensor (no host sync)."""
    return K.dice_metric(logits, y_true_onehot_u8)
