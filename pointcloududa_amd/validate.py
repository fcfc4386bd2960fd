"""Validation / inference forward of the segmenter on the HIP kernels.

Mirrors ``valid_model_with_one_dataset`` (``src/train_mscmrseg.py:53-99``): eval-mode forward (BatchNorm folded
from its running statistics), BCE + Jaccard (+ point NN) loss, hard labels, per-class Dice -- with every
per-batch quantity kept on the device (one synchronisation per data set instead of three per batch).
Hausdorff / average-surface distances (medpy, CPU) are out of scope."""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

from .utils import loss as L
from .utils import metric as M


@torch.no_grad()
def predict_labels(seg_model, x: torch.Tensor) -> torch.Tensor:
    """``evaluate_segmentation``'s core (``evaluate_mscmrseg.py:132-145``): eval forward -> uint8 label map."""
    was_training = seg_model.training
    seg_model.eval()
    try:
        out = seg_model(x)
        logits = out[0] if isinstance(out, tuple) else out
        return M.argmax_labels(logits)
    finally:
        seg_model.train(was_training)


@torch.no_grad()
def valid_batch(seg_model, x: torch.Tensor, y_onehot_u8: torch.Tensor, z: Optional[torch.Tensor] = None,
                d4: bool = True, variant: str = "mscmrseg", softmax: bool = True) -> Dict[str, torch.Tensor]:
    """One iteration of the reference's validation loop; ``seg_model`` must be in eval mode.  Returns device scalars.
    ``variant="mscmrseg"`` (``train_mscmrseg.py:67-92``): loss = BCE + Jaccard + point NN loss (l1 + l2 + l3), vert_loss
    (l3 or -1), dice = mean of classes 1..3.  ``variant="mmwhs"`` (``train_mmwhs.py:65-90``): l1 = double-softmax CE
    (``softmax``) or BCE, loss = l1 + l2 WITHOUT the point term (:82), dice = mean of classes 1..4 (``metrics2``)."""
    prediction, _, vert_s = seg_model(x)
    ms = variant == "mscmrseg"
    l1, l2 = L.seg_loss(prediction, y_onehot_u8, "sigmoid" if (ms or not softmax) else "softmax")
    loss = l1 + l2
    if d4 and vert_s is not None and z is not None:
        vert = L.batch_NN_loss(vert_s, z)
        if ms:
            loss = loss + vert
    else:
        vert = torch.full((), -1.0, dtype=torch.float32, device=x.device)
    c = prediction.shape[1]
    dc = M.label_dice(M.argmax_labels(prediction), M.argmax_labels(y_onehot_u8), c)
    return {"loss": loss, "vert_loss": vert, "dice": dc[1:(4 if ms else 5)].mean(), "dice_per_class": dc}


def valid_model_with_one_dataset(seg_model, batches: Iterable, d4: bool = True, variant: str = "mscmrseg",
                                 softmax: bool = True) -> Dict[str, float]:
    """``train_mscmrseg.py:53-99`` without the Hausdorff option: means over the batches of dice / loss /
    valid_vert_loss.  ``batches`` yields ``(x, y_onehot_u8, z)`` device tensors."""
    was_training = seg_model.training
    seg_model.eval()
    acc = {"dice": [], "loss": [], "vert_loss": []}
    try:
        for x, y, z in batches:
            r = valid_batch(seg_model, x, y, z, d4, variant, softmax)
            for k in acc:
                acc[k].append(r[k])
    finally:
        seg_model.train(was_training)
    if not acc["dice"]:
        return {"dice": float("nan"), "loss": float("nan"), "valid_vert_loss": float("nan")}
    means = torch.stack([torch.stack(acc[k]).mean() for k in ("dice", "loss", "vert_loss")]).tolist()   # one sync
    return {"dice": means[0], "loss": means[1], "valid_vert_loss": means[2]}
