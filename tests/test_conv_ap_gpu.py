"""The anti-phase 3x3 kernel (csrc/conv_ap_impl.h, round 6: one 512-thread workgroup per CU, two groups in anti-phase) against
plain PyTorch-CPU fp32 references of the same convolutions (unet.py:23,27,116,122), through the C ABI.  Every case ASSERTS
that the dispatcher chose the kernel (``pcuda_last_kernel``): a result from the ordinary kernels would prove nothing here.
Tolerance: bf16x3 carries ~2^-17 relative error per product -> 1e-4 of the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _small_maps_on_the_kernel(monkeypatch):
    """the dispatcher gives the kernel only launches with enough (tile pair, co tile) items for the chip; these cases are small"""
    monkeypatch.setenv("PCUDA_AP_MIN_ITEMS", "0")

# (n, cin, cout, h, w, bias, slope): forward rows = cout (a multiple of 64), reduction = cin (a multiple of 16), whole
# 32 x 8-pixel tiles, an even number of them; the data gradient of the same layer has rows = cin, reduction = cout
CASES = [
    (2, 64, 64, 32, 32, True, 0.01),      # one co tile, four chunks, one item per workgroup
    (2, 32, 128, 16, 64, True, 0.01),     # two co tiles, two tile columns
    (4, 128, 64, 24, 32, True, 1.0),      # eight chunks, three tile rows
    (6, 64, 128, 40, 96, True, 0.01),     # three tile columns, more items than ... one per workgroup
    (2, 16, 64, 8, 64, False, 0.2),       # ONE chunk per tile: an epilogue in every memory segment
    (34, 64, 64, 64, 64, True, 0.01),     # 544 items: workgroups walk several items (one weight chunk after the other tile's last)
    (3, 256, 192, 16, 32, True, 0.01),    # three co tiles, 16 chunks
]


def _is_ap(K):
    return "conv3ap" in K.last_kernel()


@pytest.mark.parametrize("case", CASES)
def test_forward_with_statistics_and_dgrad(dev, case):
    from pointcloududa_amd import kernels as K
    n, cin, cout, h, w_, bias, slope = case
    rng = np.random.default_rng(hash(case) & 0xffff)
    x = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w_)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32)) if bias else None
    xr = x.clone().requires_grad_(True)
    z = F.conv2d(xr, w, b, padding=1)
    y_ref = F.leaky_relu(z, slope) if slope != 1.0 else z
    gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
    z.backward(gz)
    op = K.ConvOp(cin, cout, 3, pad=1)
    fb = K.fallback_count()
    y, part, nt = op.forward(x.to(dev), w.to(dev), None if b is None else b.to(dev), slope, h, w_, want_stats=True)
    assert _is_ap(K) and K.fallback_count() == fb, K.last_kernel()
    assert rel_err(y, y_ref) < 1e-4
    assert nt == n * (h // 8) * (w_ // 32)
    s = part[:nt].double().sum(0).cpu()
    assert rel_err(s[:, 0], y_ref.double().sum((0, 2, 3))) < 1e-3
    assert rel_err(s[:, 1], (y_ref.double() ** 2).sum((0, 2, 3))) < 1e-3
    # per-tile partial sums are sums of the STORED values of that tile
    yt = y.double().reshape(n, cout, h // 8, 8, w_ // 32, 32)
    t1 = yt.sum((3, 5)).permute(0, 2, 3, 1).reshape(nt, cout)
    assert rel_err(part[:nt, :, 0].double(), t1) < 1e-5
    # without statistics
    y2, _, _ = op.forward(x.to(dev), w.to(dev), None if b is None else b.to(dev), slope, h, w_)
    assert _is_ap(K) and torch.equal(y2, y)
    # data gradient: rows = cin must be a multiple of 64 for the kernel to take it
    dx = op.dgrad(gz.to(dev), w.to(dev), h, w_)
    assert _is_ap(K) == (cin % 64 == 0), K.last_kernel()
    assert rel_err(dx, xr.grad) < 1e-4


def test_two_sources_affine_on_load_and_split_gradient(dev):
    """zero-copy concat (both sources a multiple of 16 channels), the lazy-BatchNorm affine on the first, zero padding applied
    AFTER the affine; the data gradient split over two destinations, plain and accumulating"""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    rng = np.random.default_rng(11)
    n, c1, c2, cout, h, w_ = 3, 48, 80, 128, 16, 64
    a = torch.from_numpy(rng.normal(0, 1, (n, c1, h, w_)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 1, (n, c2, h, w_)).astype(np.float32))
    sc = torch.from_numpy(rng.normal(1, 0.3, (c1,)).astype(np.float32))
    sf = torch.from_numpy(rng.normal(0, 0.3, (c1,)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, (cout, c1 + c2, 3, 3)).astype(np.float32))
    bias = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32))
    xin = torch.cat([a * sc[None, :, None, None] + sf[None, :, None, None], b], 1).requires_grad_(True)
    z = F.conv2d(xin, w, bias, padding=1)
    y_ref = F.leaky_relu(z, 0.01)
    gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
    z.backward(gz)
    op = K.ConvOp(c1 + c2, cout, 3, pad=1)
    src = TA(a.to(dev), sc.to(dev), sf.to(dev))
    y, part, nt = op.forward(src, w.to(dev), bias.to(dev), 0.01, h, w_, x2=b.to(dev), want_stats=True)
    assert _is_ap(K), K.last_kernel()
    assert rel_err(y, y_ref) < 1e-4
    assert rel_err(part[:nt].double().sum(0).cpu()[:, 0], y_ref.double().sum((0, 2, 3))) < 1e-3
    # the gradient of the concatenated input goes to two tensors (rows = 128, first destination 64 channels)
    op2 = K.ConvOp(128, 64, 3, pad=1)
    w2 = torch.from_numpy(rng.normal(0, 0.1, (64, 128, 3, 3)).astype(np.float32))
    gz2 = torch.from_numpy(rng.normal(0, 1, (n, 64, h, w_)).astype(np.float32))
    ref = F.conv_transpose2d(gz2, w2, padding=1)
    d1 = torch.empty((n, 64, h, w_), device=dev)
    d2 = torch.empty((n, 64, h, w_), device=dev)
    op2.dgrad(gz2.to(dev), w2.to(dev), h, w_, dx=d1, dx2=d2)
    assert _is_ap(K), K.last_kernel()
    assert rel_err(d1, ref[:, :64]) < 1e-4 and rel_err(d2, ref[:, 64:]) < 1e-4
    base = torch.from_numpy(rng.normal(0, 1, (n, 64, h, w_)).astype(np.float32)).to(dev)
    d1b, d2b = base.clone(), torch.zeros_like(d2)
    op2.dgrad(gz2.to(dev), w2.to(dev), h, w_, dx=d1b, dx2=d2b, accumulate=True)
    assert _is_ap(K), K.last_kernel()
    assert rel_err(d1b - base, ref[:, :64]) < 1e-4 and rel_err(d2b, ref[:, 64:]) < 1e-4


@pytest.mark.parametrize("acc", [False, True])
def test_dgrad_with_fused_bn_backward_reduce(dev, acc):
    """conv -> LeakyReLU -> BN -> conv (unet.py:23-30): the second convolution's data gradient with the first BatchNorm's
    backward reduce in its epilogue, on the anti-phase kernel: (sum g, sum g * (a - mean) * invstd) per channel against the
    sums over the reference gradient (fp64)"""
    from pointcloududa_amd import kernels as K
    n, cin, cout, h, w_ = 4, 64, 128, 32, 64
    rng = np.random.default_rng(3 + acc)
    a = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w_)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.05, (cout, cin, 3, 3)).astype(np.float32))
    gz = torch.from_numpy(rng.normal(0, 1, (n, cout, h, w_)).astype(np.float32))
    mean, invstd = a.mean((0, 2, 3)), 1.0 / torch.sqrt(a.var((0, 2, 3), unbiased=False) + 1e-5)
    g_ref = F.conv_transpose2d(gz, w, padding=1).double()
    base = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w_)).astype(np.float32))
    if acc:
        g_ref = g_ref + base.double()
    ahat = (a.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    s1, s2 = g_ref.sum((0, 2, 3)), (g_ref * ahat).sum((0, 2, 3))
    st = K.BNState()
    st.mean, st.invstd = mean.to(dev), invstd.to(dev)
    op = K.ConvOp(cin, cout, 3, pad=1)
    dx = base.to(dev).clone() if acc else None
    dx, red = op.dgrad(gz.to(dev), w.to(dev), h, w_, dx=dx, accumulate=acc, bnred=(a.to(dev), st))
    assert "conv3ap+bnred" in K.last_kernel(), K.last_kernel()
    assert red is not None
    part, nt = red
    got = part[:nt].double().sum(0).cpu()
    assert rel_err(dx, g_ref.float()) < 1e-4
    assert rel_err(got[:, 0], s1) < 1e-4 and rel_err(got[:, 1], s2) < 1e-4


def test_results_do_not_depend_on_the_launch_and_match_the_ordinary_kernel(dev, monkeypatch):
    """bit-reproducible across launches; PCUDA_AP=0 (a fresh process would be needed to flip the cached switch, so here:)
    the same layer on a map the kernel does NOT take (30 rows) runs on the ordinary kernel and agrees to the tolerance"""
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(9)
    n, cin, cout, w_ = 2, 64, 64, 32
    w = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32)).to(dev)
    op = K.ConvOp(cin, cout, 3, pad=1)
    x = torch.from_numpy(rng.normal(0, 1, (n, cin, 32, w_)).astype(np.float32)).to(dev)
    y1, p1, _ = op.forward(x, w, b, 0.01, 32, w_, want_stats=True)
    assert _is_ap(K)
    y2, p2, _ = op.forward(x, w, b, 0.01, 32, w_, want_stats=True)
    assert torch.equal(y1, y2) and torch.equal(p1, p2)
    x30 = x[:, :, :30].contiguous()
    y30, _, _ = op.forward(x30, w, b, 0.01, 30, w_)
    assert not _is_ap(K)
    ref = F.leaky_relu(F.conv2d(x30.cpu(), w.cpu(), b.cpu(), padding=1), 0.01)
    assert rel_err(y30, ref) < 1e-4


@pytest.mark.parametrize("n,cin,cout,h,w_", [(4, 128, 64, 32, 64), (2, 256, 128, 16, 32), (6, 64, 64, 24, 96)])
def test_forward_through_the_nearest_x2_fold_and_its_folded_data_gradient(dev, n, cin, cout, h, w_):
    """up-convolution (unet.py:111-112): the forward reads the STORED half-resolution input through the nearest x2 fold -- the
    kernel's LDS tile holds the stored pixels, its fragment addresses do the fold --; the data gradient is written at the
    stored resolution (the 2x2 sum in the epilogue), with and without the BatchNorm-backward reduce of the layer in front"""
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(cin + h)
    x = torch.from_numpy(rng.normal(0, 1, (n, cin, h // 2, w_ // 2)).astype(np.float32))
    wt = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32))
    xr = x.clone().requires_grad_(True)
    z = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), wt, b, padding=1)
    gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
    z.backward(gz)
    op = K.ConvOp(cin, cout, 3, pad=1, in_up=True)
    y, part, nt = op.forward(x.to(dev), wt.to(dev), b.to(dev), 1.0, h, w_, want_stats=True)
    assert _is_ap(K), K.last_kernel()
    assert rel_err(y, z) < 1e-4
    assert rel_err(part[:nt].double().sum(0).cpu()[:, 0], z.double().sum((0, 2, 3))) < 1e-3
    got = op.dgrad_fold(gz.to(dev), wt.to(dev), h, w_)
    want_k = "conv3ap+fold" if cin % 64 == 0 else "igemm"
    assert want_k in K.last_kernel(), K.last_kernel()
    assert rel_err(got, xr.grad) < 1e-4
    a = torch.from_numpy(rng.normal(0, 1, (n, cin, h // 2, w_ // 2)).astype(np.float32)).to(dev)
    st = K.BNState()
    st.mean = torch.from_numpy(rng.normal(0, 0.3, (cin,)).astype(np.float32)).to(dev)
    st.invstd = torch.from_numpy(rng.uniform(0.5, 2.0, (cin,)).astype(np.float32)).to(dev)
    got2, red = op.dgrad_fold(gz.to(dev), wt.to(dev), h, w_, bnred=(a, st))
    assert "conv3ap+fold" in K.last_kernel(), K.last_kernel()
    assert torch.equal(got2, got) and red is not None
    part, nt = red
    gd = got.double()
    tot = part[:nt].double().sum(0)
    want2 = (gd * ((a.double() - st.mean.double()[None, :, None, None]) * st.invstd.double()[None, :, None, None])).sum((0, 2, 3))
    assert rel_err(tot[:, 0], gd.sum((0, 2, 3))) < 1e-4 and rel_err(tot[:, 1], want2) < 1e-4
