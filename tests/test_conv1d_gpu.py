"""The exact-fp32 MFMA k=1 Conv1d kernels of PointNetCls (csrc/conv1d_f32.hip) against float64 torch-CPU references
of torch.nn.Conv1d(cin, cout, 1) (PointNetCls.py:26-28,76-78,116-131) and its two gradients.  The kernels compute a
k-ordered fp32 fma chain (no operand rounding): the bound is a few fp32 ulps of the accumulated magnitude, 2e-6 of the
tensor's scale -- three orders of magnitude inside what the bf16x3 path these layers used to run on could hold."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

SHAPES = [  # (b, cin, cout, l): PointNetCls's layers, the ext variant's narrow ones, ragged everything
    (32, 3, 64, 300), (16, 64, 128, 300), (12, 128, 1024, 300), (16, 3, 8, 300), (6, 8, 64, 300), (4, 512, 1024, 300),
    (5, 64, 64, 300), (3, 70, 130, 37), (2, 1, 1, 5), (7, 33, 65, 301),
]


@pytest.mark.parametrize("b,cin,cout,l", SHAPES)
def test_conv1d_k1_forward_stats_dgrad_wgrad(dev, b, cin, cout, l):
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(b * 1000 + cin + cout + l)
    x = torch.from_numpy(rng.standard_normal((b, cin, l)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32))
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((b, cout, l)).astype(np.float32))
    xd, wd, bd, dyd = x.double(), w.double(), bias.double(), dy.double()
    y_ref = torch.einsum("oc,bcl->bol", wd, xd) + bd[None, :, None]
    dx_ref = torch.einsum("oc,bol->bcl", wd, dyd)
    dw_ref = torch.einsum("bol,bcl->oc", dyd, xd)
    db_ref = dyd.sum((0, 2))

    X, W, B, DY = x.to(dev), w.to(dev), bias.to(dev), dy.to(dev)
    y, part, nt = K.conv1d_fwd(X, W, B, want_stats=True)
    assert rel_err(y, y_ref) < 2e-6
    # BatchNorm partial sums of the epilogue: per-tile (sum, sum of squares) add up to the statistics of y
    assert part.shape == (nt, cout, 2)
    tot = part.double().sum(0).cpu()
    assert rel_err(tot[:, 0], y_ref.sum((0, 2))) < 1e-5
    assert rel_err(tot[:, 1], (y_ref * y_ref).sum((0, 2))) < 1e-5
    y2, none, _ = K.conv1d_fwd(X, W, None)
    assert none is None and rel_err(y2, y_ref - bd[None, :, None]) < 2e-6

    assert rel_err(K.conv1d_dgrad(DY, W), dx_ref) < 2e-6

    dw = torch.full((cout, cin), 0.5, dtype=torch.float32, device=dev)
    db = torch.full((cout,), -0.25, dtype=torch.float32, device=dev)
    K.conv1d_wgrad(X, DY, dw, db, accumulate=True)
    assert rel_err(dw, dw_ref + 0.5) < 4e-6
    assert rel_err(db, db_ref - 0.25) < 4e-6
    dw2 = torch.empty_like(dw)
    K.conv1d_wgrad(X, DY, dw2, None, accumulate=False)
    assert rel_err(dw2, dw_ref) < 4e-6
    # deterministic: fixed-order split-K
    dw3 = torch.empty_like(dw)
    K.conv1d_wgrad(X, DY, dw3, None, accumulate=False)
    assert torch.equal(dw2, dw3)
