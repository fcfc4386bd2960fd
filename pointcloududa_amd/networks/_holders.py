"""Parameter holders that keep the reference's module tree (and therefore its state_dict keys).

They subclass the torch containers only for parameter registration / initialisation /
(de)serialisation.  Their own ``forward`` is never the ATen op: the parent network runs the HIP
kernels over their parameters in one fused autograd node, and calling a holder on its own
raises (there is no CPU or ATen fallback in this package).
"""
from __future__ import annotations

import torch
from torch import nn


def _no_standalone(self, *a, **k):
    raise RuntimeError("%s is a parameter holder: run the enclosing network (its forward launches the HIP kernels)"
                       % type(self).__name__)


class Conv2d(nn.Conv2d):
    forward = _no_standalone


class Conv1d(nn.Conv1d):
    forward = _no_standalone


class Linear(nn.Linear):
    forward = _no_standalone


class BatchNorm2d(nn.BatchNorm2d):
    forward = _no_standalone


class BatchNorm1d(nn.BatchNorm1d):
    forward = _no_standalone


class InstanceNorm1d(nn.InstanceNorm1d):
    forward = _no_standalone


class LeakyReLU(nn.LeakyReLU):
    forward = _no_standalone


class Marker(nn.Module):
    """parameter-free placeholder (UpsamplingNearest2d / MaxPool2d / Dropout positions in a Sequential)"""

    def __init__(self, what: str):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

    forward = _no_standalone


def collect(module: nn.Module):
    """(names, tensors) of all parameters then all buffers, in registration order."""
    names, tensors = [], []
    for k, v in module.named_parameters():
        names.append(k); tensors.append(v)
    for k, v in module.named_buffers():
        names.append(k); tensors.append(v)
    return names, tensors


def ensure_grad(p: torch.Tensor) -> torch.Tensor:
    """The buffer gradients are accumulated into (created zero-filled on first use)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad
