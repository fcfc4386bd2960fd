"""The row-streaming 3x3 kernel of the 32 -> 32-channel layers (csrc/conv_rs.hip, round 6: every wave walks down a 32-pixel
column strip, rows in a wave-private LDS ring, no barriers) against plain PyTorch-CPU fp32 references of the same convolutions
(unet.py:23,27,116,122 at the full-resolution level), through the C ABI.  Every case ASSERTS that the dispatcher chose the
kernel (``pcuda_last_kernel``).  Tolerance: bf16x3 carries ~2^-17 relative error per product -> 1e-4 of the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _small_maps_on_the_kernel(monkeypatch):
    """the dispatcher gives the kernel maps of at least 64 rows and launches with enough (image, strip, row segment) items for
    the chip; these cases are smaller"""
    monkeypatch.setenv("PCUDA_RS_MIN_ROWS", "2")
    monkeypatch.setenv("PCUDA_RS_MIN_ITEMS", "0")


# (n, h, w, bias, slope)
CASES = [
    (2, 64, 64, True, 0.01),       # two strips, rows split into segments (prologue / halo rows of a neighbouring segment)
    (3, 37, 96, True, 0.01),       # three strips, a row count that does not divide by the unroll of four
    (1, 2, 32, False, 0.2),        # the smallest map: one strip, two rows (top and bottom padding in the same window)
    (2, 19, 32, True, 1.0),        # one strip: both halo columns outside the image
    (5, 128, 128, True, 0.01),     # more items than one wave round of a small launch
]


def _is_rs(K):
    return "conv3rs" in K.last_kernel()


@pytest.mark.parametrize("case", CASES)
def test_forward_with_statistics_and_dgrad(dev, case):
    from pointcloududa_amd import kernels as K
    n, h, w_, bias, slope = case
    rng = np.random.default_rng(hash(case) & 0xffff)
    x = torch.from_numpy(rng.normal(0, 1, (n, 32, h, w_)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, (32, 32, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 0.1, (32,)).astype(np.float32)) if bias else None
    xr = x.clone().requires_grad_(True)
    z = F.conv2d(xr, w, b, padding=1)
    y_ref = F.leaky_relu(z, slope) if slope != 1.0 else z
    gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
    z.backward(gz)
    op = K.ConvOp(32, 32, 3, pad=1)
    fb = K.fallback_count()
    y, part, nt = op.forward(x.to(dev), w.to(dev), None if b is None else b.to(dev), slope, h, w_, want_stats=True)
    assert _is_rs(K) and K.fallback_count() == fb, K.last_kernel()
    assert rel_err(y, y_ref) < 1e-4
    assert nt % (n * (w_ // 32)) == 0          # items = (image, strip, row segment)
    s = part[:nt].double().sum(0).cpu()
    assert rel_err(s[:, 0], y_ref.double().sum((0, 2, 3))) < 1e-3
    assert rel_err(s[:, 1], (y_ref.double() ** 2).sum((0, 2, 3))) < 1e-3
    # the partial sums are sums of the STORED values
    assert rel_err(s[:, 0], y.double().sum((0, 2, 3)).cpu()) < 1e-5
    # without statistics: the same bits
    y2, _, _ = op.forward(x.to(dev), w.to(dev), None if b is None else b.to(dev), slope, h, w_)
    assert _is_rs(K) and torch.equal(y2, y)
    # data gradient, plain and accumulating
    dx = op.dgrad(gz.to(dev), w.to(dev), h, w_)
    assert _is_rs(K), K.last_kernel()
    assert rel_err(dx, xr.grad) < 1e-4
    base = torch.from_numpy(rng.normal(0, 1, x.shape).astype(np.float32)).to(dev)
    dxa = base.clone()
    op.dgrad(gz.to(dev), w.to(dev), h, w_, dx=dxa, accumulate=True)
    assert _is_rs(K), K.last_kernel()
    assert rel_err(dxa - base, xr.grad) < 1e-4


def test_affine_on_load_and_what_stays_on_the_ordinary_kernels(dev):
    """the lazy-BatchNorm affine (zero padding applied AFTER the affine: a shift far from zero); a zero-copy concat of 16 + 16
    channels and a gradient split over two destinations are not this kernel's (one source, one destination): they run on the
    ordinary kernels, with the same results"""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    rng = np.random.default_rng(12)
    n, h, w_ = 3, 24, 64
    a = torch.from_numpy(rng.normal(0, 1, (n, 16, h, w_)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 1, (n, 16, h, w_)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, (32, 32, 3, 3)).astype(np.float32))
    bias = torch.from_numpy(rng.normal(0, 0.1, (32,)).astype(np.float32))
    op = K.ConvOp(32, 32, 3, pad=1)
    full = torch.cat([a, b], 1)
    sc2 = torch.from_numpy(rng.normal(1, 0.3, (32,)).astype(np.float32))
    sf2 = torch.from_numpy(rng.normal(-0.6, 0.3, (32,)).astype(np.float32))
    y_ref2 = F.leaky_relu(F.conv2d(full * sc2[None, :, None, None] + sf2[None, :, None, None], w, bias, padding=1), 0.01)
    y2, part, nt = op.forward(TA(full.to(dev), sc2.to(dev), sf2.to(dev)), w.to(dev), bias.to(dev), 0.01, h, w_, want_stats=True)
    assert _is_rs(K), K.last_kernel()
    assert rel_err(y2, y_ref2) < 1e-4
    assert rel_err(part[:nt].double().sum(0).cpu()[:, 0], y_ref2.double().sum((0, 2, 3))) < 1e-3
    # two sources (no statistics: the partial sums of such a launch would be sized by this kernel's items)
    sc = torch.from_numpy(rng.normal(1, 0.3, (16,)).astype(np.float32))
    sf = torch.from_numpy(rng.normal(0.7, 0.3, (16,)).astype(np.float32))
    xin = torch.cat([a * sc[None, :, None, None] + sf[None, :, None, None], b], 1)
    y_ref = F.leaky_relu(F.conv2d(xin, w, bias, padding=1), 0.01)
    y, _, _ = op.forward(TA(a.to(dev), sc.to(dev), sf.to(dev)), w.to(dev), bias.to(dev), 0.01, h, w_, x2=b.to(dev))
    assert not _is_rs(K), K.last_kernel()
    assert rel_err(y, y_ref) < 1e-4
    # the gradient to two tensors
    gz = torch.from_numpy(rng.normal(0, 1, (n, 32, h, w_)).astype(np.float32))
    ref = F.conv_transpose2d(gz, w, padding=1)
    d1 = torch.empty((n, 8, h, w_), device=dev)
    d2 = torch.empty((n, 24, h, w_), device=dev)
    op.dgrad(gz.to(dev), w.to(dev), h, w_, dx=d1, dx2=d2)
    assert not _is_rs(K), K.last_kernel()
    assert rel_err(d1, ref[:, :8]) < 1e-4 and rel_err(d2, ref[:, 8:]) < 1e-4


@pytest.mark.parametrize("acc", [False, True])
def test_dgrad_with_fused_bn_backward_reduce(dev, acc):
    """conv -> LeakyReLU -> BN -> conv (unet.py:23-30): the second convolution's data gradient with the first BatchNorm's
    backward reduce in its epilogue: (sum g, sum g * (a - mean) * invstd) per channel against the sums over the reference
    gradient (fp64)"""
    from pointcloududa_amd import kernels as K
    n, h, w_ = 3, 40, 64
    rng = np.random.default_rng(5 + acc)
    a = torch.from_numpy(rng.normal(0, 1, (n, 32, h, w_)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.05, (32, 32, 3, 3)).astype(np.float32))
    gz = torch.from_numpy(rng.normal(0, 1, (n, 32, h, w_)).astype(np.float32))
    mean, invstd = a.mean((0, 2, 3)), 1.0 / torch.sqrt(a.var((0, 2, 3), unbiased=False) + 1e-5)
    g_ref = F.conv_transpose2d(gz, w, padding=1).double()
    base = torch.from_numpy(rng.normal(0, 1, (n, 32, h, w_)).astype(np.float32))
    if acc:
        g_ref = g_ref + base.double()
    ahat = (a.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    s1, s2 = g_ref.sum((0, 2, 3)), (g_ref * ahat).sum((0, 2, 3))
    st = K.BNState()
    st.mean, st.invstd = mean.to(dev), invstd.to(dev)
    op = K.ConvOp(32, 32, 3, pad=1)
    dx = base.to(dev).clone() if acc else None
    dx, red = op.dgrad(gz.to(dev), w.to(dev), h, w_, dx=dx, accumulate=acc, bnred=(a.to(dev), st))
    assert "conv3rs+bnred" in K.last_kernel(), K.last_kernel()
    assert red is not None
    part, nt = red
    got = part[:nt].double().sum(0).cpu()
    assert rel_err(dx, g_ref.float()) < 1e-4
    assert rel_err(got[:, 0], s1) < 1e-3 and rel_err(got[:, 1], s2) < 1e-3


def test_reproducible_and_no_fallback(dev):
    """the same bits on a second run (fixed summation order everywhere), no fallback launch, and the ordinary kernels' result
    within bf16x3 rounding of this kernel's (PCUDA_RS=0 is read once per process: compared through the reference instead)"""
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(21)
    n, h, w_ = 4, 64, 64
    x = torch.from_numpy(rng.normal(0, 1, (n, 32, h, w_)).astype(np.float32)).to(dev)
    w = torch.from_numpy(rng.normal(0, 0.1, (32, 32, 3, 3)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.normal(0, 0.1, (32,)).astype(np.float32)).to(dev)
    op = K.ConvOp(32, 32, 3, pad=1)
    fb = K.fallback_count()
    y1, p1, nt = op.forward(x, w, b, 0.01, h, w_, want_stats=True)
    assert _is_rs(K)
    y2, p2, _ = op.forward(x, w, b, 0.01, h, w_, want_stats=True)
    assert torch.equal(y1, y2) and torch.equal(p1[:nt], p2[:nt]) and K.fallback_count() == fb
    # every sample is computed independently of its neighbours in the batch
    y3, _, _ = op.forward(x[1:3].contiguous(), w, b, 0.01, h, w_)
    assert _is_rs(K) and torch.equal(y3, y1[1:3])


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_geometries(dev, seed):
    """random (batch, rows, strips): every row-segment split the plan can produce (segments of 16 .. 2 x 16 - 1 rows, remainders of
    the unroll), forward with the affine + statistics and the accumulating data gradient with the BatchNorm-backward reduce"""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    rng = np.random.default_rng(100 + seed)
    n, h, w_ = int(rng.integers(1, 7)), int(rng.integers(2, 150)), 32 * int(rng.integers(1, 5))
    x = torch.from_numpy(rng.normal(0, 1, (n, 32, h, w_)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, (32, 32, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 0.1, (32,)).astype(np.float32))
    sc = torch.from_numpy(rng.normal(1, 0.3, (32,)).astype(np.float32))
    sf = torch.from_numpy(rng.normal(0.4, 0.3, (32,)).astype(np.float32))
    y_ref = F.leaky_relu(F.conv2d(x * sc[None, :, None, None] + sf[None, :, None, None], w, b, padding=1), 0.01)
    op = K.ConvOp(32, 32, 3, pad=1)
    y, part, nt = op.forward(TA(x.to(dev), sc.to(dev), sf.to(dev)), w.to(dev), b.to(dev), 0.01, h, w_, want_stats=True)
    assert _is_rs(K), K.last_kernel()
    assert rel_err(y, y_ref) < 1e-4, (n, h, w_)
    s = part[:nt].double().sum(0).cpu()
    assert rel_err(s[:, 0], y_ref.double().sum((0, 2, 3))) < 1e-3 and rel_err(s[:, 1], (y_ref.double() ** 2).sum((0, 2, 3))) < 1e-3
    gz = torch.from_numpy(rng.normal(0, 1, (n, 32, h, w_)).astype(np.float32))
    a = torch.from_numpy(rng.normal(0, 1, (n, 32, h, w_)).astype(np.float32))
    mean, invstd = a.mean((0, 2, 3)), 1.0 / torch.sqrt(a.var((0, 2, 3), unbiased=False) + 1e-5)
    base = torch.from_numpy(rng.normal(0, 1, (n, 32, h, w_)).astype(np.float32))
    g_ref = F.conv_transpose2d(gz, w, padding=1).double() + base.double()
    ahat = (a.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    st = K.BNState()
    st.mean, st.invstd = mean.to(dev), invstd.to(dev)
    dx, red = op.dgrad(gz.to(dev), w.to(dev), h, w_, dx=base.to(dev).clone(), accumulate=True, bnred=(a.to(dev), st))
    assert "conv3rs+bnred" in K.last_kernel(), K.last_kernel()
    assert rel_err(dx, g_ref.float()) < 1e-4, (n, h, w_)
    got = red[0][:red[1]].double().sum(0).cpu()
    assert rel_err(got[:, 0], g_ref.sum((0, 2, 3))) < 1e-3 and rel_err(got[:, 1], (g_ref * ahat).sum((0, 2, 3))) < 1e-3
