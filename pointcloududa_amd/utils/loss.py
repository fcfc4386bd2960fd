"""HIP-backed losses with the reference's function names (``src/utils/loss.py``) plus the fused
forms the train step uses.

* ``jaccard_loss(true, logits, eps, activation)`` / ``batch_NN_loss(x, y)``: the reference
  signatures (loss.py:5-37, 40-76), backed by the fused kernels.
* ``seg_loss(logits, onehot_u8, mode)``: BCE + Jaccard (train_mscmrseg.py:202-203) or the
  "double softmax" cross-entropy + Jaccard (train_mmwhs.py:212-218) in one pass each way.
* ``entropy_map(logits, mode, normalise)``: -p log(p + 1e-7) [/ log C] (train_mscmrseg.py:222;
  train_mmwhs.py:224,242), optionally also returning p.
* ``bce_logits_const(d_out, label, weight)``: weight * BCE-with-logits against a constant map
  (train_mscmrseg.py:224-226), with the discriminator accuracy as a by-product.
"""
from __future__ import annotations

import math

import torch

from .. import kernels as K


def _zero(dev):
    return torch.zeros((), dtype=torch.float32, device=dev)


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, onehot, mode):
        logits = logits.contiguous()
        out2, ws = K.seg_loss_fwd(logits, onehot, mode)
        ctx.logits, ctx.onehot, ctx.mode, ctx.ws = logits, onehot, mode, ws
        ctx.set_materialize_grads(False)
        return out2[0], out2[1]

    @staticmethod
    def backward(ctx, g_main, g_jac):
        dev = ctx.logits.device
        g_main = _zero(dev) if g_main is None else g_main.contiguous()
        g_jac = _zero(dev) if g_jac is None else g_jac.contiguous()
        d = K.seg_loss_bwd(ctx.logits, ctx.onehot, ctx.mode, ctx.ws, g_main, g_jac)
        return d, None, None


def seg_loss(logits, onehot_u8, mode="sigmoid"):
    """-> (bce_or_ce, jaccard) scalars.  onehot_u8: uint8 one-hot [B,C,H,W] (utils.py:25-29 layout)."""
    return _SegLossFn.apply(logits, onehot_u8, mode)


class _JaccardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, true, eps):
        loss, ws, truth = K.jaccard_fwd(probs, true, eps)
        ctx.truth, ctx.ws, ctx.eps, ctx.shape = truth, ws, eps, tuple(probs.shape)
        ctx.set_materialize_grads(False)
        return loss

    @staticmethod
    def backward(ctx, gout):
        if gout is None:
            return None, None, None
        return K.jaccard_bwd(ctx.truth, ctx.shape, ctx.eps, ctx.ws, gout.contiguous()), None, None


def jaccard_loss(true, logits, eps=1e-7, activation=True):
    """Reference signature and semantics (loss.py:5-37).  ``logits``: [B,C,H,W]; with ``activation=False`` (how
    train_mscmrseg.py:203 and train_mmwhs.py:218 call it) they already are probabilities, with ``activation=True``
    they go through a softmax over the channels first; ``true``: one-hot [B,C,H,W] (float or uint8).  Differentiable
    w.r.t. ``logits`` (through whatever produced them: the result is an ordinary autograd node on the HIP kernels).
    The reference's C == 1 branch pairs [sigmoid, 1 - sigmoid] with ``true`` BROADCAST over both channels (the
    one-hot it builds at loss.py:15-19 is overwritten at :27): reproduced as executed.
    The train step itself uses the fused ``seg_loss`` below (one pass over the logits for both loss terms)."""
    num_classes = logits.shape[1]
    if num_classes == 1:
        pos = entropy_map(logits, "sigmoid", False, want_prob=True)[1]
        probas = torch.cat([pos, 1 - pos], dim=1)                                     # loss.py:20-22
        return _JaccardFn.apply(probas, true.float().expand_as(probas).contiguous(), float(eps))   # :27
    if activation:
        probas = entropy_map(logits, "softmax", False, want_prob=True)[1]
    else:
        probas = logits
    return _JaccardFn.apply(probas, true, float(eps))


class _EntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, mode, norm, want_prob, want_mean):
        logits = logits.contiguous()
        ent, prob = K.entropy_fwd(logits, mode, norm, want_prob)
        ctx.logits, ctx.mode, ctx.norm = logits, mode, norm
        ctx.set_materialize_grads(False)
        outs = [ent]
        if want_prob:
            outs.append(prob)
        if want_mean:     # torch.mean(torch.sum(map, dim=1)): train_mmwhs.py:225,243
            outs.append(K.sum_all(ent, 1.0 / (logits.shape[0] * (logits.numel() // (logits.shape[0] * logits.shape[1])))))
        ctx.layout = (want_prob, want_mean)
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, dent, *rest):
        want_prob, want_mean = ctx.layout
        rest = list(rest)
        dprob = rest.pop(0) if want_prob else None
        dmean = rest.pop(0) if want_mean else None
        if dent is None and dprob is None and dmean is None:
            return None, None, None, None, None
        d = K.entropy_bwd(ctx.logits, ctx.mode, ctx.norm, dent, dprob, dmean=dmean)
        return d, None, None, None, None


def entropy_map(logits, mode="sigmoid", normalise=False, want_prob=False, want_mean=False):
    """-> ent [, prob] [, mean]: ``mean`` = torch.mean(torch.sum(ent, dim=1)), the entropy loss term of the MM-WHS loop
    (train_mmwhs.py:225-230,243-247), differentiable like the map itself"""
    norm = 1.0 / math.log(logits.shape[1]) if normalise else 1.0
    return _EntropyFn.apply(logits, mode, norm, want_prob, want_mean)


class _EntropyTapFn(torch.autograd.Function):
    """(logits, entropy(logits)) with ONE gradient join: d_logits = d_tap + J^T d_ent is formed by the
    entropy kernel's accumulate store instead of a separate add (d1 sees the raw logits and d2 their
    entropy map in train_mscmrseg.py:222-241)."""

    @staticmethod
    def forward(ctx, logits, mode, norm):
        logits = logits.contiguous()
        ent, _ = K.entropy_fwd(logits, mode, norm, False)
        ctx.logits, ctx.mode, ctx.norm = logits, mode, norm
        ctx.set_materialize_grads(False)
        return logits.view_as(logits), ent

    @staticmethod
    def backward(ctx, dtap, dent):
        if dent is None:
            return dtap, None, None
        if dtap is None:
            return K.entropy_bwd(ctx.logits, ctx.mode, ctx.norm, dent, None), None, None
        dtap = dtap.contiguous()
        K.entropy_bwd(ctx.logits, ctx.mode, ctx.norm, dent, None, out=dtap, accumulate=True)
        return dtap, None, None


def logits_and_entropy(logits, mode="sigmoid", normalise=False):
    norm = 1.0 / math.log(logits.shape[1]) if normalise else 1.0
    return _EntropyTapFn.apply(logits, mode, norm)


class _BceConstFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, label, weight, want_acc):
        x = x.contiguous()
        loss, acc = K.bce_const_fwd(x, label, want_acc)
        ctx.x, ctx.label, ctx.weight = x, label, weight
        ctx.set_materialize_grads(False)
        if weight != 1.0:
            # the scalar is rescaled on the host side of the graph by the backward's gscale; the
            # forward value is reported unscaled next to it
            pass
        if want_acc:
            ctx.mark_non_differentiable(acc)
            return loss, acc
        return loss

    @staticmethod
    def backward(ctx, gout, gacc=None):
        if gout is None:
            return None, None, None, None
        return K.bce_const_bwd(ctx.x, ctx.label, gout.contiguous(), ctx.weight), None, None, None


def bce_logits_const(d_out, label, weight=1.0, want_acc=False):
    """mean BCE-with-logits of ``d_out`` against the constant ``label``.  The returned scalar is the
    UNWEIGHTED loss; ``weight`` (the reference's ``args.dr``) scales its gradient, i.e. backpropagating
    the returned value with grad 1 equals backpropagating ``weight * loss``."""
    return _BceConstFn.apply(d_out, float(label), float(weight), want_acc)


class _NNLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x, y = x.contiguous(), y.contiguous()
        loss, idx, val = K.nn_loss_fwd(x, y)
        ctx.x, ctx.y, ctx.idx, ctx.val = x, y, idx, val
        ctx.set_materialize_grads(False)
        return loss

    @staticmethod
    def backward(ctx, gout):
        if gout is None:
            return None, None
        return K.nn_loss_bwd(ctx.x, ctx.y, ctx.idx, ctx.val, gout.contiguous()), None


def batch_NN_loss(x, y):
    """loss.py:40-76: symmetric nearest-neighbour (Chamfer, L2 root) distance between [B,N,3] clouds.
    Differentiable w.r.t. ``x`` (the predicted vertices); ``y`` is data."""
    return _NNLossFn.apply(x, y)
