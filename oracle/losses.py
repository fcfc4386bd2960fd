"""CPU restatement of the loss arithmetic on the hot path (oracle; test-only).

Reference (relative to /root/reference/src):
  utils/loss.py:5-37    jaccard_loss                  -> jaccard_loss
  utils/loss.py:40-76   batch_NN_loss                 -> batch_nn_loss
  train_mscmrseg.py:222,265     entropy map (sigmoid, un-normalised)   -> entropy_map(mode="sigmoid")
  train_mmwhs.py:224,240-242    entropy map (softmax|sigmoid, /log C)  -> entropy_map(..., normalise=True)
  train_mscmrseg.py:202-203     BCELoss(sigmoid(o), y) + jaccard       -> seg_loss_sigmoid
  train_mmwhs.py:212-218        cross_entropy(softmax(o), argmax y) + jaccard(softmax(o)) -> seg_loss_softmax
  train_mscmrseg.py:224-226     F.binary_cross_entropy_with_logits(D_out, const) -> bce_logits_const
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

SMOOTH = 1e-7          # train_mscmrseg.py:164


def jaccard_loss(true, probs, eps: float = 1e-7):
    """loss.py:27-37 with ``activation=False`` and C > 1 (the only way the scripts call it):
    1 - mean_c( sum(p*y) / (sum(p+y) - sum(p*y) + eps) ), sums over batch and space."""
    true = true.to(probs.dtype)
    dims = (0,) + tuple(range(2, true.ndim))
    inter = torch.sum(probs * true, dims)
    card = torch.sum(probs + true, dims)
    return 1.0 - (inter / (card - inter + eps)).mean()


def jaccard_loss_ref_signature(true, logits, eps: float = 1e-7, activation: bool = True):
    """loss.py:5-37 with every branch: C > 1 with ``activation`` (softmax over channels) or without; C == 1:
    probabilities [sigmoid, 1 - sigmoid].  NB the one-hot the C == 1 branch builds (:15-19) is dead code: line 27,
    ``true_1_hot = true.type(probas.type())``, replaces it with ``true`` itself, which then BROADCASTS over the two
    probability channels -- restated as the reference executes, not as it reads."""
    c = logits.shape[1]
    if c == 1:
        pos = torch.sigmoid(logits)
        probs = torch.cat([pos, 1 - pos], dim=1)                                           # :20-22
        true = true.to(probs.dtype).expand_as(probs)                                       # :27 + broadcasting in :31,33
    else:
        probs = F.softmax(logits, dim=1) if activation else logits                         # :24
    return jaccard_loss(true, probs, eps)


def batch_nn_loss(x, y):
    """loss.py:40-76.  x,y: [B,N,3].  P[b,i,j] = |x_i|^2 + |y_j|^2 - 2 x_i.y_j (three bmm's,
    so NOT clamped at 0), d = sqrt(P + 1e-5); mean_i min_j d  +  mean_j min_i d, averaged over B.

    The reference builds the second matrix with a second call (``batch_pairwise_dist(y, x)``),
    which is the transpose of the first; restated the same way to keep rounding identical.
    """
    def pdist(a, b):
        aa = (a * a).sum(-1)                # diag(a a^T)   (loss.py:56,60)
        bb = (b * b).sum(-1)                # diag(b b^T)
        ab = torch.bmm(a, b.transpose(2, 1))
        return aa[:, :, None] + bb[:, None, :] - 2 * ab

    bs, n, _ = x.shape
    d1 = torch.sqrt(pdist(x, y) + 0.00001).min(dim=2)[0]
    d2 = torch.sqrt(pdist(y, x) + 0.00001).min(dim=2)[0]
    a = d1.sum(1) / n
    b = d2.sum(1) / n
    return a.sum() / bs + b.sum() / bs


def entropy_map(logits, mode: str = "sigmoid", normalise: bool = False):
    """-p * log(p + 1e-7) per channel (NOT summed over classes).
    mode 'sigmoid' un-normalised = train_mscmrseg.py:222; 'softmax'/'sigmoid' with
    ``normalise`` (divide by log C) = train_mmwhs.py:224,242."""
    p = torch.sigmoid(logits) if mode == "sigmoid" else F.softmax(logits, dim=1)
    e = -1.0 * p * torch.log(p + SMOOTH)
    if normalise:
        e = e / math.log(logits.shape[1])
    return e


def seg_loss_sigmoid(logits, onehot):
    """train_mscmrseg.py:202-203 -> (BCE mean, jaccard)."""
    p = torch.sigmoid(logits)
    y = onehot.to(p.dtype)
    return F.binary_cross_entropy(p, y), jaccard_loss(y, p)


def seg_loss_softmax(logits, onehot):
    """train_mmwhs.py:212-214,218: the reference feeds *probabilities* into
    F.cross_entropy, i.e. log_softmax is applied on top of softmax ("double softmax")."""
    p = F.softmax(logits, dim=1)
    y = onehot.to(p.dtype)
    ce = F.cross_entropy(p, torch.argmax(onehot, dim=1).long())
    return ce, jaccard_loss(y, p)


def bce_logits_const(d_out, label: float):
    """mean BCE-with-logits against a constant target map (train_mscmrseg.py:224-226)."""
    return F.binary_cross_entropy_with_logits(d_out, torch.full_like(d_out, float(label)))
