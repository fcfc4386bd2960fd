"""Direct (vector-ALU) kernels of the degenerate layers (csrc/conv_direct.hip): the segmenter's 1-channel first
convolution (unet.py:23; forward with BatchNorm partial sums, weight gradient) and the 1x1 classifier
(unet.py:178; forward with a lazy-BatchNorm input, data gradient into a split destination), each against a plain
PyTorch-CPU fp32 reference of the same op, through the same C-ABI entry points as every other layer.  The weights
are the packed hi + lo bf16 pair (2^-17 relative), the arithmetic fp32 FMA: tolerance 1e-4 of the output scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 32, 64, 64), (3, 8, 36, 20), (2, 64, 16, 256), (1, 5, 10, 12)])
@pytest.mark.parametrize("stats", [True, False])
def test_first_layer_forward_and_wgrad(dev, shape, stats):
    from pointcloududa_amd import kernels as K
    n, cout, h, w_ = shape
    rng = np.random.default_rng(n * 1000 + cout + h)
    x = torch.from_numpy(rng.uniform(0, 1, (n, 1, h, w_)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, (cout, 1, 3, 3)).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32)).requires_grad_(True)
    z = F.conv2d(x, w, b, padding=1)
    y_ref = F.leaky_relu(z, 0.01)
    gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
    z.backward(gz)
    op = K.ConvOp(1, cout, 3, pad=1)
    y, part, nt = op.forward(x.to(dev), w.detach().to(dev), b.detach().to(dev), 0.01, h, w_, want_stats=stats)
    assert rel_err(y, y_ref) < 1e-4
    if stats:
        s = part[:nt].double().sum(0).cpu()
        assert rel_err(s[:, 0], y_ref.double().sum((0, 2, 3))) < 1e-4
        assert rel_err(s[:, 1], (y_ref.double() ** 2).sum((0, 2, 3))) < 1e-4
    dw = torch.full((cout, 1, 3, 3), 7.0, device=dev)
    db = torch.full((cout,), 7.0, device=dev)
    op.wgrad(x.to(dev), gz.to(dev), dw, db, h, w_, accumulate=False)
    assert rel_err(dw, w.grad) < 1e-4 and rel_err(db, b.grad) < 1e-4
    op.wgrad(x.to(dev), gz.to(dev), dw, db, h, w_, accumulate=True)
    assert rel_err(dw, 2 * w.grad) < 1e-4 and rel_err(db, 2 * b.grad) < 1e-4


@pytest.mark.parametrize("shape", [(2, 32, 4, 64, 64), (2, 48, 5, 12, 20), (1, 64, 8, 8, 8), (2, 16, 1, 16, 16)])
def test_classifier_forward_and_dgrad(dev, shape):
    """input = cat(lazy-BatchNorm tensor, plain tensor) as two sources; dgrad accumulates into a split destination"""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    n, cin, cout, h, w_ = shape
    c1 = cin // 2 if cin >= 32 else cin
    rng = np.random.default_rng(cin * 100 + cout)
    a = torch.from_numpy(rng.normal(0, 1, (n, c1, h, w_)).astype(np.float32))
    sc = torch.from_numpy(rng.normal(1, 0.2, (c1,)).astype(np.float32))
    sf = torch.from_numpy(rng.normal(0, 0.2, (c1,)).astype(np.float32))
    parts = [a * sc[None, :, None, None] + sf[None, :, None, None]]
    b2 = None
    if c1 < cin:
        b2 = torch.from_numpy(rng.normal(0, 1, (n, cin - c1, h, w_)).astype(np.float32))
        parts.append(b2)
    xin = torch.cat(parts, 1).requires_grad_(True)
    w = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 1, 1)).astype(np.float32))
    bias = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32))
    z = F.conv2d(xin, w, bias)
    gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
    z.backward(gz)
    op = K.ConvOp(cin, cout, 1)
    src = TA(a.to(dev), sc.to(dev), sf.to(dev))
    y, _, _ = op.forward(src, w.to(dev), bias.to(dev), 1.0, h, w_, x2=None if b2 is None else b2.to(dev))
    assert rel_err(y, z) < 1e-4
    y2, _, _ = op.forward(src, w.to(dev), bias.to(dev), 0.2, h, w_, x2=None if b2 is None else b2.to(dev))
    assert rel_err(y2, F.leaky_relu(z, 0.2)) < 1e-4
    d1 = torch.ones((n, c1, h, w_), device=dev)
    d2 = torch.ones((n, cin - c1, h, w_), device=dev) if c1 < cin else None
    op.dgrad(gz.to(dev), w.to(dev), h, w_, dx=d1, dx2=d2, accumulate=True)
    got = d1 if d2 is None else torch.cat([d1, d2], 1)
    assert rel_err(got - 1.0, xin.grad) < 1e-4
    op.dgrad(gz.to(dev), w.to(dev), h, w_, dx=d1, dx2=d2, accumulate=False)
    got = d1 if d2 is None else torch.cat([d1, d2], 1)
    assert rel_err(got, xin.grad) < 1e-4


@pytest.mark.parametrize("shape", [(2, 4, 64, 64, 64), (3, 5, 64, 32, 48), (2, 4, 32, 256, 256), (2, 1, 64, 16, 24)])
def test_discriminator_first_layer_dgrad_and_wgrad(dev, shape):
    """cin <= 5 -> 64 channels, 4x4, stride 2, pad 2 (GAN.py:97): data gradient (with and without accumulation) and
    weight gradient (cin <= 4) on the direct kernels, against torch."""
    from pointcloududa_amd import kernels as K
    n, cin, cout, h, w_ = shape
    rng = np.random.default_rng(cin * 1000 + h)
    x = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w_)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy(rng.normal(0, 0.05, (cout, cin, 4, 4)).astype(np.float32)).requires_grad_(True)
    z = F.conv2d(x, w, None, stride=2, padding=2)
    gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
    z.backward(gz)
    op = K.ConvOp(cin, cout, 4, stride=2, pad=2)
    wd, gzd = w.detach().to(dev), gz.to(dev)
    y, _, _ = op.forward(x.detach().to(dev), wd, None, 1.0, h, w_)
    assert rel_err(y, z) < 1e-4
    dx = op.dgrad(gzd, wd, h, w_)
    assert rel_err(dx, x.grad) < 1e-4
    base = torch.full((n, cin, h, w_), 0.5, device=dev)
    op.dgrad(gzd, wd, h, w_, dx=base, accumulate=True)
    assert rel_err(base - 0.5, x.grad) < 1e-4
    dw = torch.full((cout, cin, 4, 4), 3.0, device=dev)
    op.wgrad(x.detach().to(dev), gzd, dw, None, h, w_, accumulate=False)
    assert rel_err(dw, w.grad) < 1e-4
    op.wgrad(x.detach().to(dev), gzd, dw, None, h, w_, accumulate=True)
    assert rel_err(dw, 2 * w.grad) < 1e-4


@pytest.mark.parametrize("shape", [(4, 512, 17, 17), (3, 256, 9, 9), (2, 72, 17, 13)])
def test_discriminator_last_layer_forward(dev, shape):
    """cin -> 1 channel, 4x4, stride 2, pad 2 on a small map (GAN.py:101): the direct forward kernel against torch
    (ragged last 16-channel chunk included); its gradients stay on the MFMA kernels and are checked there."""
    from pointcloududa_amd import kernels as K
    n, cin, h, w_ = shape
    rng = np.random.default_rng(cin + h)
    x = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w_)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy(rng.normal(0, 0.05, (1, cin, 4, 4)).astype(np.float32)).requires_grad_(True)
    z = F.conv2d(x, w, None, stride=2, padding=2)
    gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
    z.backward(gz)
    op = K.ConvOp(cin, 1, 4, stride=2, pad=2)
    y, _, _ = op.forward(x.detach().to(dev), w.detach().to(dev), None, 1.0, h, w_)
    assert rel_err(y, z) < 1e-4
    y2, _, _ = op.forward(x.detach().to(dev), w.detach().to(dev), None, 0.2, h, w_)
    assert rel_err(y2, F.leaky_relu(z, 0.2)) < 1e-4
    assert rel_err(op.dgrad(gz.to(dev), w.detach().to(dev), h, w_), x.grad) < 1e-4
    dw = torch.zeros((1, cin, 4, 4), device=dev)
    op.wgrad(x.detach().to(dev), gz.to(dev), dw, None, h, w_, accumulate=False)
    assert rel_err(dw, w.grad) < 1e-4


# (enough tiles for the two-workgroups-per-CU kernel that carries the transposed epilogue: > 320 tile x row-tile items)
@pytest.mark.parametrize("shape", [(2, 32, 32, 256, 256, False), (6, 48, 64, 128, 128, True), (2, 16, 8, 64, 64, False)])
def test_dgrad_with_fused_bn_backward_reduce(dev, shape):
    """conv -> LeakyReLU -> BN -> conv (unet.py:23-30): the second convolution's data gradient with the first BatchNorm's
    backward reduce in its epilogue: (sum g, sum g * (a - mean) * invstd) per channel against the sums over the reference
    gradient (fp64), plain and accumulating into an existing gradient."""
    from pointcloududa_amd import kernels as K
    n, cin, cout, h, w_, acc = shape
    rng = np.random.default_rng(cin + cout)
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
    assert rel_err(dx, g_ref.float()) < 1e-4
    if h < 128 and red is None:      # few tiles: where the one-workgroup-per-CU kernel takes the layer the caller reduces separately
        return                       # (round 4's plan rules give this geometry to the transposed-epilogue kernel: checked below)
    assert red is not None, "this geometry runs on the transposed-epilogue kernel"
    part, nt = red
    got = part[:nt].double().sum(0).cpu()
    assert rel_err(dx, g_ref.float()) < 1e-4
    assert rel_err(got[:, 0], s1) < 1e-4 and rel_err(got[:, 1], s2) < 1e-4


def test_decoder_reduce_fusions(dev):
    """the two producers of a decoder block's incoming gradient carry its second BatchNorm's backward-reduce partials:
    the 2x2 fold behind an up-convolution (pointwise kernel) and the 1x1 classifier's direct dgrad"""
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(77)
    n, c, h, w_ = 3, 6, 24, 40
    gy = torch.from_numpy(rng.normal(0, 1, (n, c, 2 * h, 2 * w_)).astype(np.float32))
    a = torch.from_numpy(rng.normal(0, 1, (n, c, h, w_)).astype(np.float32))
    mean, invstd = a.mean((0, 2, 3)), 1.0 / torch.sqrt(a.var((0, 2, 3), unbiased=False) + 1e-5)
    st = K.BNState()
    st.mean, st.invstd = mean.to(dev), invstd.to(dev)
    g_ref = F.avg_pool2d(gy.double(), 2) * 4
    ahat = (a.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    dx, red = K.upsample2_bwd(gy.to(dev), bnred=(a.to(dev), st))
    part, nt = red
    got = part[:nt].double().sum(0).cpu()
    assert rel_err(dx, g_ref.float()) < 1e-6
    assert rel_err(got[:, 0], g_ref.sum((0, 2, 3))) < 1e-5 and rel_err(got[:, 1], (g_ref * ahat).sum((0, 2, 3))) < 1e-5
    # classifier: 32 -> 4, 1x1
    n, cin, cout, h, w_ = 2, 32, 4, 48, 64
    a = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w_)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 1, 1)).astype(np.float32))
    gz = torch.from_numpy(rng.normal(0, 1, (n, cout, h, w_)).astype(np.float32))
    mean, invstd = a.mean((0, 2, 3)), 1.0 / torch.sqrt(a.var((0, 2, 3), unbiased=False) + 1e-5)
    st.mean, st.invstd = mean.to(dev), invstd.to(dev)
    g_ref = F.conv_transpose2d(gz, w).double()
    ahat = (a.double() - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    op = K.ConvOp(cin, cout, 1)
    dx, red = op.dgrad(gz.to(dev), w.to(dev), h, w_, bnred=(a.to(dev), st))
    assert red is not None
    part, nt = red
    got = part[:nt].double().sum(0).cpu()
    assert rel_err(dx, g_ref.float()) < 1e-4
    assert rel_err(got[:, 0], g_ref.sum((0, 2, 3))) < 1e-4 and rel_err(got[:, 1], (g_ref * ahat).sum((0, 2, 3))) < 1e-4


@pytest.mark.parametrize("prec,tol", [("bf16x3", 1e-4), ("bf16", 2e-2)])
@pytest.mark.parametrize("n,cin,h,w_", [(2, 4, 64, 64), (3, 5, 34, 72), (1, 1, 6, 8), (2, 3, 130, 256), (2, 2, 18, 40)])
def test_discriminator_first_layer_forward_reads_taps_from_the_image(dev, prec, tol, n, cin, h, w_):
    """GAN.py:97 (<= 5 maps -> 64 channels, 4x4 / stride 2 / pad 2, bias, LeakyReLU 0.2) on d1_fwd_kernel -- one MFMA k-step
    per input channel, the 16 taps read straight from the image -- against the CPU reference: every channel count the
    kernel is instantiated for, maps whose last 32-pixel tile is partial, borders on all four sides; and the library
    says it takes these geometries (pcuda_conv2d_d1_forward_ok), so that no unfolded tensor is built for them"""
    from pointcloududa_amd import kernels as K
    K.set_precision(prec)
    try:
        rng = np.random.default_rng(cin * 100 + h)
        x = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w_)).astype(np.float32))
        w = torch.from_numpy(rng.normal(0, 0.1, (64, cin, 4, 4)).astype(np.float32))
        b = torch.from_numpy(rng.normal(0, 0.1, (64,)).astype(np.float32))
        ref = F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=2), 0.2)
        op = K.ConvOp(cin, 64, 4, stride=2, pad=2)
        assert K.d1_forward_direct(op, n, h, w_)
        y, _, _ = op.forward(x.to(dev), w.to(dev), b.to(dev), 0.2, h, w_)
        assert K.last_kernel().startswith("direct d1 fwd (mfma)"), K.last_kernel()      # (not the unfold + 1x1 route)
        assert y.shape == ref.shape and rel_err(y, ref) < tol
        y2, _, _ = op.forward(x.to(dev), w.to(dev), None, 1.0, h, w_)            # no bias, no activation
        assert rel_err(y2, F.conv2d(x, w, None, stride=2, padding=2)) < tol
    finally:
        K.set_precision("bf16x3")
