"""Flat parameter / gradient buffers and the fused optimiser steps.

MI355X-first layout: every network keeps ALL its parameters in one contiguous fp32 buffer and all
its gradients in another (288 GB of HBM: nothing is ever sharded or offloaded).  The backward
kernels accumulate straight into the gradient buffer, ``zero_grad`` is one memset, the optimiser is
one kernel launch, and data parallelism is one RCCL all-reduce per network and step.

Adam(betas=(0.9, 0.99)) for the segmenter and SGD(momentum, weight_decay=5e-4) for the
discriminators follow train_mscmrseg.py:427-455 / train_mmwhs.py:453-489.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import kernels as K

_ALIGN = 64   # elements: every parameter starts on a 256-B boundary


def flatten_module(module: nn.Module):
    """Re-home the module's parameters (and their .grad) as views of two flat buffers.
    Returns (flat_param, flat_grad).  Idempotent."""
    if getattr(module, "_flat_param", None) is not None:
        return module._flat_param, module._flat_grad
    params = [p for p in module.parameters()]
    if not params:
        raise ValueError("module has no parameters")
    dev = params[0].device
    offs, total = [], 0
    for p in params:
        if p.dtype != torch.float32:
            raise TypeError("flatten_module: fp32 parameters only")
        offs.append(total)
        total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    grad = torch.zeros(total, dtype=torch.float32, device=dev)
    for p, o in zip(params, offs):
        n = p.numel()
        flat[o:o + n].copy_(p.data.reshape(-1))
        p.data = flat[o:o + n].view(p.shape)
        p.grad = grad[o:o + n].view(p.shape)
    tracked = [b for k, b in module.named_buffers() if k.endswith("num_batches_tracked")]
    if tracked:
        ft = torch.zeros(len(tracked), dtype=torch.long, device=dev)
        for i, b in enumerate(tracked):
            ft[i] = b
            b.data = ft[i]
        module._flat_tracked = ft
    module._flat_param, module._flat_grad = flat, grad
    return flat, grad


_collectives_override = None     # None: automatic; False: off (bench.py's "step without communication" leg)


def set_collectives(mode) -> None:
    """``False``: run the step without its gradient all-reduces even in a multi-rank group (measurement only: bench.py's
    ``comm_exposed_ms`` = step time with the collectives minus step time without; every rank must switch together).
    ``None``: automatic again."""
    global _collectives_override
    _collectives_override = mode


def _collectives_on(group=None) -> bool:
    """True when gradients have to be all-reduced: an initialised process group of more than one rank.
    ``PCUDA_FORCE_COLLECTIVES=1`` keeps the collectives in a one-rank group too (to exercise the RCCL path -- stream
    ordering, async handles -- on a single-GPU machine)."""
    import os
    import torch.distributed as dist
    if _collectives_override is False or not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("PCUDA_FORCE_COLLECTIVES") == "1"


class _FlatOptimizer:
    def __init__(self, module: nn.Module, lr: float):
        self.module = module
        self.p, self.g = flatten_module(module)
        self.lr = float(lr)
        self.steps = 0

    def zero_grad(self):
        self.g.zero_()

    # ---- torch.optim-format state (what the reference's checkpoints hold: callbacks.py:61-94)
    def _slices(self):
        """(offset, numel, shape) of every parameter inside the flat buffers, in module.parameters() order --
        the order torch.optim indexes its state by when built from ``model.parameters()``"""
        out, total = [], 0
        for p in self.module.parameters():
            out.append((total, p.numel(), tuple(p.shape)))
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        return out

    def all_reduce_grads(self, group=None):
        import torch.distributed as dist
        if _collectives_on(group):
            dist.all_reduce(self.g, op=dist.ReduceOp.SUM, group=group)
            return 1.0 / dist.get_world_size(group)
        return 1.0

    def all_reduce_grads_async(self, group=None, lo=0, hi=None):
        """Start the all-reduce of the flat gradient (or of its slice [lo, hi)) and return ``(work, scale)``;
        ``finish_all_reduce(work)`` must run before ``step(scale)``.  Lets the caller put independent kernels under
        the collective: RCCL runs on its own stream, ordered after everything already enqueued on the current one."""
        import torch.distributed as dist
        if _collectives_on(group):
            g = self.g if (lo == 0 and hi is None) else self.g[lo:hi]
            if g.numel() == 0:
                return None, 1.0 / dist.get_world_size(group)
            work = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True)
            return work, 1.0 / dist.get_world_size(group)
        return None, 1.0

    def split_after(self, prefix: str) -> int:
        """Offset in the flat buffers at which the parameters whose names start with ``prefix`` end, if they form
        the HEAD of the buffer (registration order); 0 otherwise.  Used to all-reduce the rest of the gradient while
        the backward pass is still producing the head's (the segmenter's encoder is first in the buffer and last in
        the backward pass)."""
        off, head_end, seen_other = 0, 0, False
        for name, p in self.module.named_parameters():
            n = (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            if name.startswith(prefix):
                if seen_other:
                    return 0
                head_end = off + n
            else:
                seen_other = True
            off += n
        return head_end if seen_other else 0

    @staticmethod
    def finish_all_reduce(work):
        """``work``: one handle or a list of them (None entries allowed)"""
        for w in (work if isinstance(work, (list, tuple)) else [work]):
            if w is not None:
                w.wait()      # the current stream waits for the collective; the host does not block on NCCL


class FusedAdam(_FlatOptimizer):
    def __init__(self, module, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, skip_prefixes=None):
        super().__init__(module, lr)
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        # ``skip_prefixes``: names of the parameters whose ``.grad`` stays None in the reference (torch.optim.Adam holds no
        # state for them); None = unknown, decided from the second moments when the state is exported
        self.skip_prefixes = None if skip_prefixes is None else tuple(skip_prefixes)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        # the step count lives on the device (the kernel increments it): a captured graph of the step replays right
        self.step_t = torch.zeros(1, dtype=torch.int32, device=self.p.device)

    def step(self, grad_scale: float = 1.0):
        self.steps += 1   # host-side count of step() calls (not advanced by graph replays; see state_dict)
        self.module._wgen = getattr(self.module, "_wgen", 0) + 1   # packed conv weights are stale now
        K.adam_step_dev(self.p, self.g, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                        self.step_t, grad_scale)
        K.repack_owner(self.module)   # every packed conv-weight layout of the network, one launch behind the update

    def state_dict(self):
        return {"steps": int(self.step_t.item()), "lr": self.lr, "exp_avg": self.m, "exp_avg_sq": self.v}

    def torch_state_dict(self):
        """The state in ``torch.optim.Adam.state_dict()`` form (loadable by the reference's optimiser)."""
        step = int(self.step_t.item())
        state = {}
        if step > 0:
            sl = self._slices()
            # torch.optim.Adam holds no state for a parameter whose .grad was always None (the reference's never-used
            # encoder.conv1_1; the point head without a loss on it): here that is a second moment that is still all zero
            if self.skip_prefixes is not None:
                # by NAME: a parameter that did receive gradients which happened to be exactly zero keeps its state (and
                # its step count) in torch.optim.Adam too
                used = [not (self.skip_prefixes and k.startswith(self.skip_prefixes)) for k, _ in self.module.named_parameters()]
            else:
                used = torch.stack([self.v[o:o + n].max() for o, n, _ in sl]).gt(0).tolist()        # one transfer
            for i, (o, n, shp) in enumerate(sl):
                if not used[i]:
                    continue
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": self.m[o:o + n].view(shp).clone(),
                            "exp_avg_sq": self.v[o:o + n].view(shp).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self._slices())))}
        return {"state": state, "param_groups": [group]}

    def load_torch_state_dict(self, sd):
        """Inverse of ``torch_state_dict``: accepts what ``torch.optim.Adam.state_dict()`` (any torch >= 1.4) holds."""
        g = sd["param_groups"][0]
        self.lr = float(g["lr"]); self.betas = tuple(g["betas"]); self.eps = float(g["eps"])
        self.wd = float(g.get("weight_decay", 0.0))
        self.m.zero_(); self.v.zero_()
        step = 0
        for i, (o, n, shp) in enumerate(self._slices()):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is None:
                continue
            self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            step = max(step, int(float(st["step"])))
        self.step_t.fill_(step)
        self.steps = step


class FusedSGD(_FlatOptimizer):
    def __init__(self, module, lr=2.5e-5, momentum=0.99, weight_decay=0.0005, skip_prefixes=()):
        super().__init__(module, lr)
        self.momentum, self.wd = momentum, weight_decay
        self.buf = torch.zeros_like(self.p) if momentum != 0 else None
        # ``skip_prefixes``: parameters that never receive a gradient in the reference (``.grad is None``): torch.optim
        # skips them altogether -- no weight decay, no momentum buffer.  The flat buffer is updated in the contiguous
        # ranges between them.
        self.ranges = [(0, self.p.numel())]
        if skip_prefixes:
            self.ranges, lo, off = [], 0, 0
            for name, p in module.named_parameters():
                n = (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
                if name.startswith(tuple(skip_prefixes)):
                    if off > lo:
                        self.ranges.append((lo, off))
                    lo = off + n
                off += n
            if off > lo:
                self.ranges.append((lo, off))

    def step(self, grad_scale: float = 1.0):
        self.module._wgen = getattr(self.module, "_wgen", 0) + 1
        for lo, hi in self.ranges:
            K.sgd_step(self.p[lo:hi], self.g[lo:hi], None if self.buf is None else self.buf[lo:hi], self.lr, self.momentum,
                       self.wd, self.steps == 0, grad_scale)
        self.steps += 1
        K.repack_owner(self.module)

    def state_dict(self):
        return {"steps": self.steps, "lr": self.lr, "momentum_buffer": self.buf}

    def torch_state_dict(self):
        """The state in ``torch.optim.SGD.state_dict()`` form."""
        state = {}
        if self.buf is not None and self.steps > 0:
            for i, (o, n, shp) in enumerate(self._slices()):
                if not any(lo <= o and o + n <= hi for lo, hi in self.ranges):
                    continue          # a skipped parameter (.grad is None in the reference): torch.optim.SGD holds no state for it
                state[i] = {"momentum_buffer": self.buf[o:o + n].view(shp).clone()}
        group = {"lr": self.lr, "momentum": self.momentum, "dampening": 0, "weight_decay": self.wd, "nesterov": False,
                 "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                 "params": list(range(len(self._slices())))}
        return {"state": state, "param_groups": [group]}

    def load_torch_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr = float(g["lr"]); self.momentum = float(g.get("momentum", 0.0)); self.wd = float(g.get("weight_decay", 0.0))
        loaded = False
        if self.buf is not None:
            self.buf.zero_()
            for i, (o, n, shp) in enumerate(self._slices()):
                st = sd["state"].get(i, sd["state"].get(str(i)))
                if st is not None and st.get("momentum_buffer") is not None and \
                        any(lo <= o and o + n <= hi for lo, hi in self.ranges):      # (skipped ranges stay zero)
                    self.buf[o:o + n].copy_(st["momentum_buffer"].reshape(-1))
                    loaded = True
        self.steps = 1 if loaded else 0   # "first step" only decides whether the momentum buffer is initialised
