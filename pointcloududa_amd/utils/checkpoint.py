"""Checkpoint I/O in the reference's on-disk format.

The reference saves ``{'epoch', 'model_state_dict', 'optimizer_state_dict'}`` through
``ModelCheckPointCallback.step`` (``src/utils/callbacks.py:61-94``: best model, renamed with ``.Scr<score>`` after
the last epoch, and optionally the last model) and loads either that dictionary or a bare state dict
(``src/train_mmwhs.py:536-583``).  The modules of this package keep the reference's ``state_dict`` keys, so model
weights interchange directly; the flat fused optimisers export / import ``torch.optim`` state dicts."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch


def _opt_state(optimizer):
    if optimizer is None:
        return None
    return optimizer.torch_state_dict() if hasattr(optimizer, "torch_state_dict") else optimizer.state_dict()


def save_checkpoint(path: str, epoch: int, model, optimizer=None) -> None:
    """One file in the reference's layout (``callbacks.py:78-80``); tensors are moved to the host."""
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    torch.save({"epoch": int(epoch), "model_state_dict": sd, "optimizer_state_dict": _opt_state(optimizer)}, path)


def load_checkpoint(path_or_obj, model, optimizer=None, strict: bool = True) -> str:
    """``train_mmwhs.py:538-551``: a checkpoint dictionary (weights + optionally optimiser state; a failing
    optimiser load is ignored like in the reference) or a bare state dict.  Returns which one it was."""
    ck = torch.load(path_or_obj, map_location="cpu") if isinstance(path_or_obj, (str, os.PathLike)) else path_or_obj
    if isinstance(ck, dict) and "model_state_dict" in ck:
        model.load_state_dict(ck["model_state_dict"], strict=strict)
        osd = ck.get("optimizer_state_dict")
        if optimizer is not None and osd is not None:
            try:
                (optimizer.load_torch_state_dict if hasattr(optimizer, "load_torch_state_dict")
                 else optimizer.load_state_dict)(osd)
            except Exception:   # noqa: BLE001 -- the reference swallows this too
                pass
        if hasattr(model, "_wgen"):
            model._wgen += 1
        return "dict"
    model.load_state_dict(ck, strict=strict)
    if hasattr(model, "_wgen"):
        model._wgen += 1
    return "single state"


class ModelCheckPointCallback:
    """Same constructor and ``step`` behaviour as ``utils/callbacks.py:15-94`` (state-dict mode): keep the best
    model by ``monitor`` (``mode`` max/min, always saved at epoch 1), rename it to ``<base>.Scr<score><ext>`` after
    the last epoch, optionally save the last model."""

    def __init__(self, mode: str = "min", model_name: str = "../weights/model_checkpoint.pt",
                 best_model_name: Optional[str] = None, save_best: bool = True, entire_model: bool = False,
                 save_last_model: bool = False, n_epochs: int = 200):
        assert mode in ("max", "min"), "mode can only be 'min' or 'max'"
        if entire_model:
            raise NotImplementedError("entire_model=True pickles the module object; only state dicts are supported")
        self.mode = mode
        self.best_result = np.inf if mode == "min" else -np.inf
        self.model_name = model_name
        self.best_model_name = best_model_name if best_model_name is not None else model_name
        self.best_model_name_base, self.ext = os.path.splitext(self.best_model_name)
        self.save_last_model, self.n_epochs, self.epoch, self._save_best = save_last_model, n_epochs, 0, save_best

    def step(self, monitor, model, epoch, optimizer=None):
        if self._save_best:
            better = monitor > self.best_result if self.mode == "max" else monitor < self.best_result
            if epoch == 1 or better:
                self.best_result, self.epoch = monitor, epoch
                save_checkpoint(self.best_model_name, epoch, model, optimizer)
            if epoch == self.n_epochs:
                os.rename(self.best_model_name, "{}{}{}{}".format(self.best_model_name_base, ".Scr",
                                                                 np.around(self.best_result, 3), self.ext))
        if self.save_last_model and epoch == self.n_epochs:
            save_checkpoint(self.model_name, epoch, model, optimizer)
