"""Validation loop body of the reference, restated (oracle; test-only).

Reference: train_mscmrseg.py:53-99 (valid_model_with_one_dataset) with metric.py:39-82 (evaluate)."""
from __future__ import annotations

import numpy as np
import torch

from . import losses as OL
from . import metrics as OM
from . import nets as ON


def valid_batch(params, x, y_onehot, z, cfg: ON.SegCfg, d4: bool = True):
    """One iteration (train_mscmrseg.py:67-92): eval-mode forward, l1 = BCE, l2 = Jaccard, l3 = NN loss,
    labels = argmax(soft_to_hard_pred), dice = mean over classes 1..3 of dc.  Returns python floats + arrays."""
    with torch.no_grad():
        p = {k: v.clone() for k, v in params.items()}
        logits, verts = ON.seg_forward(p, torch.as_tensor(x), cfg, training=False)
        l1, l2 = OL.seg_loss_sigmoid(logits, torch.as_tensor(y_onehot))
        l3 = OL.batch_nn_loss(verts, torch.as_tensor(z)) if (d4 and cfg.pointnet) else None
    loss = float(l1 + l2 + (l3 if l3 is not None else 0.0))
    pred = OM.argmax_labels(logits.numpy())
    gt = OM.argmax_labels(np.asarray(y_onehot))
    dc = OM.label_dice(pred, gt, cfg.n_class)
    return {"loss": loss, "vert_loss": float(l3) if l3 is not None else -1.0, "dice": float(dc[1:4].mean()),
            "dice_per_class": dc, "labels": pred, "logits": logits.numpy()}
