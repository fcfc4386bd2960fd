"""Fused loss / entropy / metric kernels against the golden vectors produced by the REFERENCE
(tests/golden/losses.npz, written by oracle/make_golden.py) and against the oracle on larger inputs."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, rel_err

pytestmark = pytest.mark.gpu


def _inputs(seed=7):
    rng = np.random.default_rng(seed)      # same stream as oracle/make_golden.py:gold_losses
    b, c, hw = 2, 4, 32
    logits = torch.from_numpy(rng.normal(0, 2, (b, c, hw, hw)).astype(np.float32))
    lab = rng.integers(0, c, (b, hw, hw))
    onehot = torch.from_numpy(np.moveaxis(np.eye(c, dtype=np.uint8)[lab], -1, 1).copy())
    x = torch.from_numpy(rng.random((3, 300, 3), dtype=np.float32))
    y = torch.from_numpy(rng.random((3, 300, 3), dtype=np.float32))
    return rng, logits, onehot, x, y


def test_losses_against_reference_golden(dev):
    from pointcloududa_amd.utils import loss as L
    g = np.load(os.path.join(GOLD, "losses.npz"))
    rng, logits, onehot, x, y = _inputs(int(g["seed"]))
    lo, oh = logits.to(dev), onehot.to(dev)
    # BCE + Jaccard (train_mscmrseg.py:202-203)
    l = lo.clone().requires_grad_(True)
    bce, jac = L.seg_loss(l, oh, "sigmoid")
    torch.autograd.backward([bce, jac], [torch.ones((), device=dev)] * 2)
    assert abs(float(bce) - float(g["bce"])) < 1e-5 and abs(float(jac) - float(g["jac"])) < 1e-5
    assert rel_err(l.grad, g["dlogits_sig"]) < 1e-4
    # double-softmax CE + Jaccard (train_mmwhs.py:212-218)
    l = lo.clone().requires_grad_(True)
    ce, jac = L.seg_loss(l, oh, "softmax")
    torch.autograd.backward([ce, jac], [torch.ones((), device=dev)] * 2)
    assert abs(float(ce) - float(g["ce"])) < 1e-5 and abs(float(jac) - float(g["jac_sm"])) < 1e-5
    assert rel_err(l.grad, g["dlogits_sm"]) < 1e-4
    # entropy maps
    w = torch.from_numpy(g["ent_w"]).to(dev)
    for name, mode, norm in (("ent_sig", "sigmoid", False), ("ent_sm_n", "softmax", True), ("ent_sig_n", "sigmoid", True)):
        l = lo.clone().requires_grad_(True)
        e = L.entropy_map(l, mode, norm)
        e.backward(w)
        assert rel_err(e, g[name]) < 1e-5, name
        assert rel_err(l.grad, g[name + "_grad"]) < 1e-4, name
    # nearest-neighbour point loss (loss.py:40-76)
    xd = x.to(dev).requires_grad_(True)
    nn = L.batch_NN_loss(xd, y.to(dev))
    nn.backward()
    assert abs(float(nn) - float(g["nn"])) < 1e-5
    assert rel_err(xd.grad, g["nn_dx"]) < 1e-3
    # constant-target domain loss
    rng2 = np.random.default_rng(int(g["seed"]))
    d = None
    # regenerate d exactly as the generator did: it is drawn after everything above
    _r, *_ = _inputs(int(g["seed"]))
    _ = _r.normal(0, 1, logits.shape)           # ent_w draw
    d = torch.from_numpy(_r.normal(0, 1, (2, 1, 9, 9)).astype(np.float32)).to(dev)
    for lbl in (0, 1):
        dd = d.clone().requires_grad_(True)
        l, acc = L.bce_logits_const(dd, float(lbl), 1.0, want_acc=True)
        l.backward()
        assert abs(float(l) - float(g["bce_const_%d" % lbl])) < 1e-5
        assert rel_err(dd.grad, g["bce_const_%d_grad" % lbl]) < 1e-4
        assert abs(float(acc) - float((torch.sigmoid(d) >= 0.5).float().mean())) < 1e-6


def test_losses_against_oracle_large(dev):
    """larger, class-imbalanced inputs incl. saturated logits (|o| ~ 30: the BCE log clamp at -100
    and the 1e-12 clamp of its backward are exercised)"""
    from oracle import losses as OL
    from oracle import metrics as OM
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.utils import loss as L
    rng = np.random.default_rng(11)
    b, c, hw = 3, 5, 96
    logits = torch.from_numpy(rng.normal(0, 6, (b, c, hw, hw)).astype(np.float32))
    logits[0, 0, :4] = 40.0
    logits[0, 1, :4] = -40.0
    lab = rng.integers(0, c, (b, hw, hw)); lab[:, :30] = 0
    onehot = torch.from_numpy(np.moveaxis(np.eye(c, dtype=np.uint8)[lab], -1, 1).copy())
    for mode, fn in (("sigmoid", OL.seg_loss_sigmoid), ("softmax", OL.seg_loss_softmax)):
        lr = logits.clone().requires_grad_(True)
        m_ref, j_ref = fn(lr, onehot)
        (m_ref + 0.7 * j_ref).backward()
        l = logits.to(dev).requires_grad_(True)
        m, j = L.seg_loss(l, onehot.to(dev), mode)
        torch.autograd.backward([m, j], [torch.ones((), device=dev), torch.full((), 0.7, device=dev)])
        assert abs(float(m) - float(m_ref)) < 2e-5 * max(1, abs(float(m_ref))), mode
        assert abs(float(j) - float(j_ref)) < 2e-5, mode
        assert rel_err(l.grad, lr.grad) < 2e-4, mode
    # tap + entropy with one fused gradient join
    lr = logits.clone().requires_grad_(True)
    e_ref = OL.entropy_map(lr, "sigmoid", False)
    w1, w2 = torch.randn_like(logits), torch.randn_like(logits)
    ((e_ref * w1).sum() + (lr * w2).sum()).backward()
    l = logits.to(dev).requires_grad_(True)
    tap, ent = L.logits_and_entropy(l, "sigmoid", False)
    torch.autograd.backward([ent, tap], [w1.to(dev), w2.to(dev)])
    assert rel_err(l.grad, lr.grad) < 1e-4
    # entropy + probabilities (mmwhs: d1 sees probabilities)
    lr = logits.clone().requires_grad_(True)
    p_ref = torch.softmax(lr, 1)
    e_ref = OL.entropy_map(lr, "softmax", True)
    ((e_ref * w1).sum() + (p_ref * w2).sum()).backward()
    l = logits.to(dev).requires_grad_(True)
    ent, prob = L.entropy_map(l, "softmax", True, want_prob=True)
    torch.autograd.backward([ent, prob], [w1.to(dev), w2.to(dev)])
    assert rel_err(prob, p_ref) < 1e-5 and rel_err(ent, e_ref) < 1e-5
    assert rel_err(l.grad, lr.grad) < 1e-4
    # Dice metric on the device == numpy restatement of utils.py:32-40 + metric.py:5-36
    hard = OM.soft_to_hard_pred(logits.numpy(), 1)
    d_ref = OM.dice_coef_multilabel(onehot.numpy(), hard, c)
    d = K.dice_metric(logits.to(dev), onehot.to(dev))
    assert abs(float(d) - d_ref) < 1e-6


@pytest.mark.parametrize("b,npts", [(3, 300), (2, 37), (1, 1), (2, 64), (1, 1024)])
def test_nearest_neighbour_loss_blocks_of_points(dev, b, npts):
    """batch_NN_loss (loss.py:40-76) with the search spread over blocks of 64 points and four lanes per point: values,
    nearest indices and gradient against the dense torch form, for clouds that are no multiple of the block (and of 4)"""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.utils import loss as L
    rng = np.random.default_rng(b * 1000 + npts)
    x = torch.from_numpy(rng.normal(0, 1, (b, npts, 3)).astype(np.float32))
    y = torch.from_numpy(rng.normal(0, 1, (b, npts, 3)).astype(np.float32))
    xr = x.clone().requires_grad_(True)
    r_x, r_y = (xr * xr).sum(2), (y * y).sum(2)
    P = r_x[:, :, None] + r_y[:, None, :] - 2 * torch.bmm(xr, y.transpose(1, 2))
    D = torch.sqrt(P + 0.00001)
    want = (D.min(2)[0].mean(1) + D.min(1)[0].mean(1)).mean()
    want.backward()
    loss, idx, val = K.nn_loss_fwd(x.to(dev), y.to(dev))
    assert abs(float(loss) - float(want)) < 1e-5 * max(1.0, abs(float(want)))
    assert torch.equal(idx[0].cpu().long(), D.detach().argmin(2)) and torch.equal(idx[1].cpu().long(), D.detach().argmin(1))
    xd = x.to(dev).requires_grad_(True)
    L.batch_NN_loss(xd, y.to(dev)).backward()
    assert rel_err(xd.grad, xr.grad) < 1e-3
