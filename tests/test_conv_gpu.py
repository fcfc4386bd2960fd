"""MFMA implicit-GEMM convolution (forward, dgrad = transposed conv, wgrad) against a plain
PyTorch-CPU fp32 reference of the same op, through the C ABI.  Tolerances: the bf16x3 (parity)
mode carries ~2^-17 relative error per product -> 1e-4 of the output scale; the bf16 (throughput)
mode 2^-9 per product -> 2e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = {"bf16x3": 1e-4, "bf16": 2e-2}

# (n, cin, cout, h, w, k, stride, pad, dil, in_up, bias, slope)
CASES = [
    (2, 32, 32, 64, 64, 3, 1, 1, 1, False, True, 0.01),     # encoder 3x3
    (2, 1, 32, 64, 64, 3, 1, 1, 1, False, True, 0.01),      # first layer, 1 input channel
    (2, 3, 8, 32, 32, 3, 1, 1, 1, False, True, 0.01),       # reduced config
    (2, 48, 64, 32, 32, 3, 1, 1, 1, False, True, 1.0),      # ragged channel chunk, no activation
    (2, 96, 64, 32, 32, 1, 1, 0, 1, False, True, 0.01),     # 1x1 after concat
    (2, 32, 4, 64, 64, 1, 1, 0, 1, False, True, 1.0),       # classifier
    (2, 64, 64, 16, 16, 3, 1, 2, 2, False, True, 0.01),     # dilated bottleneck d=2
    (2, 64, 64, 16, 16, 3, 1, 4, 4, False, True, 0.01),     # d=4 (clamped LDS tile)
    (2, 64, 128, 16, 16, 3, 1, 8, 8, False, True, 0.01),    # d=8
    (2, 32, 64, 4, 4, 3, 1, 8, 8, False, True, 0.01),       # d=8 on a 4x4 map (reduced config): all halo
    (2, 32, 32, 8, 8, 3, 1, 4, 4, False, True, 0.01),
    (2, 64, 300, 16, 16, 6, 1, 0, 1, False, True, 0.01),    # 6x6 valid point head
    (2, 4, 64, 64, 64, 4, 2, 2, 1, False, False, 0.2),      # discriminator conv1
    (2, 64, 128, 33, 33, 4, 2, 2, 1, False, False, 0.2),    # discriminator, odd size
    (2, 128, 1, 9, 9, 4, 2, 2, 1, False, False, 1.0),       # discriminator conv5
    (2, 64, 32, 17, 17, 3, 2, 1, 1, False, False, 0.2),     # ext discriminator 3x3 s2 p1
    (2, 64, 32, 32, 32, 3, 1, 1, 1, True, True, 1.0),       # decoder: nearest x2 folded into the conv
    (2, 4, 64, 16, 24, 4, 2, 2, 1, False, False, 0.2),      # stride-2 halo tile taller than the image, rows of 4k pixels
    (2, 8, 32, 12, 16, 3, 1, 1, 1, False, True, 0.01),      # 3x3 on a map shorter than one tile
    # shapes the fixed-geometry weight-gradient kernel takes (csrc/conv_wgrad3.hip, w3_eligible: 3x3 / stride 1 / pad 1, cout >= 128
    # and a multiple of 64, cin a multiple of 32, rows of 32k pixels, 4k rows): two tiles across, one tile across, one tile in
    # all, three chunks x three co-tile pairs, and the 224x224 network's 32-multiple-free maps that must NOT take it (28, 56)
    (2, 64, 128, 32, 64, 3, 1, 1, 1, False, True, 0.01),
    (2, 128, 256, 8, 32, 3, 1, 1, 1, False, True, 0.01),
    (1, 32, 128, 4, 32, 3, 1, 1, 1, False, True, 1.0),
    (3, 96, 192, 12, 96, 3, 1, 1, 1, False, True, 0.01),
    (2, 128, 128, 28, 28, 3, 1, 1, 1, False, True, 0.01),
    (2, 64, 64, 56, 56, 3, 1, 1, 1, False, True, 0.01),
    (2, 256, 512, 14, 14, 3, 1, 8, 8, False, True, 0.01),   # 224x224 bottleneck: dilation 8 on a 14-wide map
    (2, 64, 300, 14, 14, 6, 1, 0, 1, False, True, 0.01),    # 224x224 point head: 6x6 valid on 14x14 -> 9x9
    (2, 64, 128, 57, 57, 4, 2, 2, 1, False, False, 0.2),    # 224x224 discriminator maps
    (2, 64, 128, 29, 29, 4, 2, 2, 1, False, False, 0.2),
    (3, 3, 64, 1, 300, 1, 1, 0, 1, False, True, 1.0),       # PointNet conv1d(k=1): H = 1
    (3, 128, 1024, 1, 300, 1, 1, 0, 1, False, True, 1.0),
]


def _ref(x, w, b, k, s, p, d, up, slope):
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x, w, b, stride=s, padding=p, dilation=d)
    return F.leaky_relu(y, slope) if slope != 1.0 else y


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(dev, case, prec):
    from pointcloududa_amd import kernels as K
    n, cin, cout, h, w_, k, s, p, d, up, bias, slope = case
    K.set_precision(prec)
    try:
        rng = np.random.default_rng(hash(case) & 0xffff)
        sh, sw = (h // 2, w_ // 2) if up else (h, w_)
        x = torch.from_numpy(rng.normal(0, 1, (n, cin, sh, sw)).astype(np.float32))
        w = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, k, k)).astype(np.float32))
        b = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32)) if bias else None
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True) if bias else None
        z = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest") if up else xr, wr, br, stride=s, padding=p,
                     dilation=d)
        y_ref = F.leaky_relu(z, slope) if slope != 1.0 else z
        gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))     # gradient w.r.t. the conv output
        z.backward(gz)

        op = K.ConvOp(cin, cout, k, stride=s, pad=p, dil=d, in_up=up)
        xd, wd = x.to(dev), w.to(dev)
        bd = b.to(dev) if bias else None
        y, part, nt = op.forward(xd, wd, bd, slope, h, w_, want_stats=True)
        assert rel_err(y, y_ref) < TOL[prec], "forward"
        # BatchNorm partial sums from the epilogue: sum and sum of squares per channel
        s1 = part[:nt].double().sum(0).cpu()
        assert rel_err(s1[:, 0], y_ref.double().sum((0, 2, 3))) < max(TOL[prec], 1e-4) * 10, "bn sum"
        assert rel_err(s1[:, 1], (y_ref.double() ** 2).sum((0, 2, 3))) < max(TOL[prec], 1e-4) * 10, "bn sumsq"

        gzd = gz.to(dev)
        dx = op.dgrad(gzd, wd, h, w_)
        dx_ref = xr.grad
        if up:   # dgrad is w.r.t. the upsampled input; fold 2x2 like the autograd of nearest upsampling
            dx = K.upsample2_bwd(dx)
        assert rel_err(dx, dx_ref) < TOL[prec], "dgrad"

        dw = torch.zeros_like(wd)
        db = torch.zeros(cout, device=dev) if bias else None
        op.wgrad(xd, gzd, dw, db, h, w_, accumulate=False)
        assert rel_err(dw, wr.grad) < TOL[prec], "wgrad"
        if bias:
            assert rel_err(db, br.grad) < 1e-4, "bias grad"
        # accumulate semantics
        op.wgrad(xd, gzd, dw, db, h, w_, accumulate=True)
        assert rel_err(dw, 2 * wr.grad) < TOL[prec], "wgrad accumulate"
    finally:
        K.set_precision("bf16x3")


def test_conv_concat_affine_split(dev):
    """two-source input (zero-copy cat), per-channel affine on load (lazy BatchNorm), and a
    gradient split across two destinations"""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    rng = np.random.default_rng(5)
    n, c1, c2, cout, h, w_ = 2, 40, 24, 48, 32, 32
    a = torch.from_numpy(rng.normal(0, 1, (n, c1, h, w_)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 1, (n, c2, h, w_)).astype(np.float32))
    sc = torch.from_numpy(rng.normal(1, 0.2, (c1,)).astype(np.float32))
    sf = torch.from_numpy(rng.normal(0, 0.2, (c1,)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, (cout, c1 + c2, 3, 3)).astype(np.float32))
    bias = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32))
    xin = torch.cat([a * sc[None, :, None, None] + sf[None, :, None, None], b], 1).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    z = F.conv2d(xin, wr, bias, padding=1)
    gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
    z.backward(gz)
    op = K.ConvOp(c1 + c2, cout, 3, pad=1)
    ad, bd = a.to(dev), b.to(dev)
    src = TA(ad, sc.to(dev), sf.to(dev))
    y, _, _ = op.forward(src, w.to(dev), bias.to(dev), 1.0, h, w_, x2=bd)
    assert rel_err(y, z) < 1e-4
    d1 = torch.empty((n, c1, h, w_), device=dev)
    d2 = torch.empty((n, c2, h, w_), device=dev)
    op.dgrad(gz.to(dev), w.to(dev), h, w_, dx=d1, dx2=d2)
    assert rel_err(d1, xin.grad[:, :c1]) < 1e-4 and rel_err(d2, xin.grad[:, c1:]) < 1e-4
    dw = torch.zeros_like(w, device=dev)
    op.wgrad(src, gz.to(dev), dw, None, h, w_, x2=bd, accumulate=False)
    assert rel_err(dw, wr.grad) < 1e-4
    # accumulate into an existing gradient
    base = torch.from_numpy(rng.normal(0, 1, (n, c1, h, w_)).astype(np.float32)).to(dev)
    d1b = base.clone()
    op.dgrad(gz.to(dev), w.to(dev), h, w_, dx=d1b, dx2=d2, accumulate=True)
    assert rel_err(d1b - base, xin.grad[:, :c1]) < 1e-4


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("c1,c2,cout,h,w_", [(64, 64, 128, 16, 32), (32, 96, 192, 8, 64), (128, 0, 128, 12, 32)])
def test_wgrad3_two_sources_and_affine_on_load(dev, prec, c1, c2, cout, h, w_):
    """The fixed-geometry weight-gradient kernel's staging path with everything it can be handed at once: two sources
    (zero-copy concat, both a multiple of 32 channels), the lazy-BatchNorm affine on the first, a bias gradient,
    accumulation -- against the fp32 CPU gradient of the same convolution."""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    K.set_precision(prec)
    try:
        rng = np.random.default_rng(c1 * 7 + c2 + w_)
        n = 2
        a = torch.from_numpy(rng.normal(0, 1, (n, c1, h, w_)).astype(np.float32))
        sc = torch.from_numpy(rng.normal(1, 0.3, (c1,)).astype(np.float32))
        sf = torch.from_numpy(rng.normal(0, 0.3, (c1,)).astype(np.float32))
        parts = [a * sc[None, :, None, None] + sf[None, :, None, None]]
        b = None
        if c2:
            b = torch.from_numpy(rng.normal(0, 1, (n, c2, h, w_)).astype(np.float32))
            parts.append(b)
        xin = torch.cat(parts, 1)
        wr = torch.from_numpy(rng.normal(0, 0.1, (cout, c1 + c2, 3, 3)).astype(np.float32)).requires_grad_(True)
        br = torch.zeros(cout, requires_grad=True)
        z = F.conv2d(xin, wr, br, padding=1)
        gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
        z.backward(gz)
        op = K.ConvOp(c1 + c2, cout, 3, pad=1)
        src = TA(a.to(dev), sc.to(dev), sf.to(dev))
        dw = torch.full(wr.shape, float("nan"), device=dev)
        db = torch.full((cout,), float("nan"), device=dev)
        fb = K.fallback_count()
        op.wgrad(src, gz.to(dev), dw, db, h, w_, x2=(b.to(dev) if c2 else None), accumulate=False)
        # (the result below is only evidence for the fixed-geometry kernel if the dispatcher chose it: round-4 review, weak 3)
        assert K.last_kernel().startswith("wgrad3 ") and K.fallback_count() == fb, K.last_kernel()
        assert rel_err(dw, wr.grad) < TOL[prec] and rel_err(db, br.grad) < 1e-4
        op.wgrad(src, gz.to(dev), dw, db, h, w_, x2=(b.to(dev) if c2 else None), accumulate=True)
        assert rel_err(dw, 2 * wr.grad) < TOL[prec] and rel_err(db, 2 * br.grad) < 1e-4
    finally:
        K.set_precision("bf16x3")


@pytest.mark.parametrize("cin,cout,hw,k,s,p", [(32, 32, 256, 3, 1, 1), (64, 128, 129, 4, 2, 2), (256, 256, 32, 3, 1, 1)])
def test_conv_full_batch_properties(dev, cin, cout, hw, k, s, p, monkeypatch):
    """BASELINE's full batch (32) at the benchmark's layer shapes, through properties that need no CPU reference:
    samples are independent (the batch result of sample i IS the single-sample result, bit for bit, forward and
    dgrad: tiles never mix samples), the weight gradient of the batch is the sum of the halves' (split-K over
    pixels and samples: fp32 summation order only), and one small CPU-checked sample pins the absolute values."""
    from pointcloududa_amd import kernels as K
    K.set_precision("bf16x3")
    # (round 6: the dispatcher gives the anti-phase kernel only launches with enough work items for the chip -- the batch of 32
    #  of 256 -> 256 at 32x32, not its single samples.  Sample independence is a property of EACH kernel: both sizes on that one.)
    monkeypatch.setenv("PCUDA_AP_MIN_ITEMS", "0")
    monkeypatch.setenv("PCUDA_RS_MIN_ITEMS", "0")       # (the same for the row-streaming kernel of 32 -> 32 at 256x256)
    torch.manual_seed(5)
    n = 32
    op = K.ConvOp(cin, cout, k, stride=s, pad=p)
    x = torch.randn(n, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    y, _, _ = op.forward(x, w, b, 0.2, hw, hw)
    oh = y.shape[2]
    for i in (0, 17, 31):
        yi, _, _ = op.forward(x[i:i + 1].contiguous(), w, b, 0.2, hw, hw)
        assert torch.equal(yi[0], y[i]), i
    ref = F.leaky_relu(F.conv2d(x[3:4].cpu(), w.cpu(), b.cpu(), stride=s, padding=p), 0.2)
    assert rel_err(y[3:4], ref) < 1e-4
    gz = torch.randn(n, cout, oh, oh, device=dev)
    dx = op.dgrad(gz, w, hw, hw)
    d17 = op.dgrad(gz[17:18].contiguous(), w, hw, hw)
    assert torch.equal(d17[0], dx[17])
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    op.wgrad(x, gz, dw, db, hw, hw)
    dwh, dbh = torch.zeros_like(w), torch.zeros_like(b)
    op.wgrad(x[:16].contiguous(), gz[:16].contiguous(), dwh, dbh, hw, hw)
    op.wgrad(x[16:].contiguous(), gz[16:].contiguous(), dwh, dbh, hw, hw)      # accumulates
    assert rel_err(dw, dwh) < 2e-5 and rel_err(db, dbh) < 2e-5
    # <dy, conv(x)> = <dgrad(dy), x>: the forward (without bias / activation) and dgrad kernels are adjoint
    y_lin, _, _ = op.forward(x, w, None, 1.0, hw, hw)
    lhs, rhs = float((gz.double() * y_lin.double()).sum()), float((dx.double() * x.double()).sum())
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), abs(rhs), 1.0)


def test_deferred_split_k_reduces_are_bit_identical(dev, monkeypatch):
    """pcuda_conv2d_wgrad_partial + pcuda_wgrad_reduce_batch (several layers' reduces in one launch) against the
    one-launch-per-layer form: same arithmetic, same order -> identical bits, with and without accumulation."""
    from pointcloududa_amd import kernels as K
    monkeypatch.setattr(K, "_batch_reduce", True)
    g = torch.Generator(device="cpu").manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    layers = [(K.ConvOp(8, 16, 3, pad=1), 2, 64), (K.ConvOp(16, 8, 3, pad=1), 2, 64), (K.ConvOp(4, 64, 4, stride=2, pad=2), 2, 32),
              (K.ConvOp(64, 40, 1), 3, 24), (K.ConvOp(1, 4, 3, pad=1), 2, 64)]
    cases = []
    for op, n, hw in layers:
        oh, ow = op.out_hw(hw, hw)
        cases.append((op, rn(n, op.cin, hw, hw), rn(n, op.cout, oh, ow), hw))
    ref = []
    for op, x, dy, hw in cases:
        dw, db = torch.ones(op.cout, op.cin, op.k, op.k, device=dev), torch.ones(op.cout, device=dev)
        op.wgrad(x, dy, dw, db, hw, hw, accumulate=True)
        ref.append((dw, db))
    got = []
    with K.deferred_wgrad_reduces():
        for op, x, dy, hw in cases:
            dw, db = torch.ones(op.cout, op.cin, op.k, op.k, device=dev), torch.ones(op.cout, device=dev)
            op.wgrad(x, dy, dw, db, hw, hw, accumulate=True)
            got.append((dw, db))
        op, x, dy, hw = cases[0]
        op.wgrad(x, dy, got[0][0], got[0][1], hw, hw, accumulate=True)     # same buffers again: flushes the first in between
    op, x, dy, hw = cases[0]
    op.wgrad(x, dy, ref[0][0], ref[0][1], hw, hw, accumulate=True)
    for (a, b), (c, d) in zip(ref, got):
        assert torch.equal(a, c) and torch.equal(b, d)


# (n, cin, cout, h, w): k = 4, stride 2, pad 2 (GAN.py:97-105).  The data gradient of these layers runs one launch per
# ROW parity with rows (channel, column parity) -- csrc/conv_host.h, dgrad_pair_ok -- so every shape of the pairing is
# exercised here: odd and even maps (the odd class one column short), a single output column of the odd class missing
# altogether (w = 1), ragged channel counts (the last co-tile half empty), cout = 1 (one reduction channel), the
# benchmark's discriminator sizes.
PAIR_CASES = [(2, 64, 128, 129, 129), (2, 128, 256, 65, 65), (2, 256, 512, 33, 33), (2, 512, 1, 17, 17), (3, 8, 16, 16, 16),
              (2, 24, 40, 11, 14), (2, 96, 32, 20, 37), (1, 16, 8, 5, 1), (2, 40, 24, 2, 2), (2, 72, 64, 31, 64)]


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", PAIR_CASES)
def test_stride2_dgrad_paired_column_classes(dev, case, prec):
    from pointcloududa_amd import kernels as K
    n, cin, cout, h, w_ = case
    K.set_precision(prec)
    try:
        g = torch.Generator().manual_seed(hash(case) & 0xffff)
        oh, ow = h // 2 + 1, w_ // 2 + 1
        w = 0.1 * torch.randn(cout, cin, 4, 4, generator=g)
        dy = torch.randn(n, cout, oh, ow, generator=g)
        ref = F.grad.conv2d_input((n, cin, h, w_), w, dy, stride=2, padding=2)
        op = K.ConvOp(cin, cout, 4, stride=2, pad=2)
        dx = torch.full((n, cin, h, w_), float("nan"), device=dev)          # every element must be written
        op.dgrad(dy.to(dev), w.to(dev), h, w_, dx=dx)
        assert rel_err(dx, ref) < TOL[prec]
        base = torch.randn(n, cin, h, w_, generator=g).to(dev)              # accumulate into an existing gradient
        acc = base.clone()
        op.dgrad(dy.to(dev), w.to(dev), h, w_, dx=acc, accumulate=True)
        assert rel_err(acc - base, ref) < max(TOL[prec], 2e-4)
        if cin % 16 == 0 and cin >= 16:                                     # the gradient split over two destinations
            c1 = cin // 2
            d1 = torch.full((n, c1, h, w_), float("nan"), device=dev)
            d2 = torch.full((n, cin - c1, h, w_), float("nan"), device=dev)
            op.dgrad(dy.to(dev), w.to(dev), h, w_, dx=d1, dx2=d2)
            assert rel_err(torch.cat([d1, d2], 1), ref) < TOL[prec]
    finally:
        K.set_precision("bf16x3")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_table_repack_matches_the_per_layer_pack(dev, prec):
    """The one-launch repack of a network (kernels.repack_owner: what every optimiser step replays) writes the same packed
    images, byte for byte, as the per-layer pack -- forward layouts, the four-class and the paired dgrad layouts
    (discriminator: k = 4 / stride 2 / pad 2), 1x1 / 3x3 / dilated / 6x6 layers (segmenter)."""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.networks import Segmentation_model_Point, UncertaintyDiscriminator
    K.set_precision(prec)
    try:
        torch.manual_seed(3)
        nets = [(UncertaintyDiscriminator(4).to(dev).train(), torch.randn(2, 4, 64, 64, device=dev)),
                (Segmentation_model_Point(filters=8, in_channels=1, n_class=4, pointnet=True, fc_inch=1).to(dev).train(),
                 torch.randn(2, 1, 96, 96, device=dev))]
        for net, x in nets:
            out = net(x.requires_grad_(True))
            (out[0] if isinstance(out, tuple) else out).sum().backward()          # forward + dgrad images exist now
            ops = [op for op in K._conv_ops if op.owner is net and op._last_fwd is not None]
            assert len(ops) >= 5
            with torch.no_grad():
                for q in net.parameters():
                    q.mul_(1.25).add_(0.01)                                         # (in place: same storage)
            net._wgen = getattr(net, "_wgen", 0) + 1
            assert K.repack_owner(net)
            torch.cuda.synchronize()
            got = [(op, {k: v[3].clone() for k, v in op._pk.items() if k[1] == K._precision}) for op in ops]
            for op, packed in got:
                w, g = op._last_fwd
                assert packed, "no packed image"
                for (kind, _), img in packed.items():
                    fresh = torch.zeros_like(img)
                    lib = K.L.lib()
                    fn = lib.pcuda_conv2d_pack_fwd if kind == "fwd" else lib.pcuda_conv2d_pack_dgrad
                    K.check(fn(K.C.byref(g), K._precision, w.data_ptr(), fresh.data_ptr(), K._stream()), "pack")
                    torch.cuda.synchronize()
                    assert torch.equal(img.view(torch.uint8), fresh.view(torch.uint8)), (kind, op.cin, op.cout, op.k)
    finally:
        K.set_precision("bf16x3")


def test_pruned_build_falls_back_to_the_generic_kernels(dev):
    """The default build holds the template instantiations the benchmark configurations and the network-level tests reach
    (csrc/variants.h, built_variants.h); any other geometry must run -- correctly -- on the generic kernels and say so.
    A 3x3 layer on a 20 x 52 map (ragged tiles, dword staging) is in no configuration: forward, dgrad and wgrad against the
    CPU reference, and the fallback counter moves unless this is a FULL build."""
    from pointcloududa_amd import kernels as K
    lib = K.L.lib()
    n, cin, cout, h, w_ = 2, 48, 80, 20, 52
    rng = np.random.default_rng(77)
    x = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w_)).astype(np.float32))
    w = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    z = F.conv2d(xr, wr, None, padding=1)
    gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
    z.backward(gz)
    before = lib.pcuda_fallback_count()
    op = K.ConvOp(cin, cout, 3, pad=1)
    y, _, _ = op.forward(x.to(dev), w.to(dev), None, 1.0, h, w_)
    dx = op.dgrad(gz.to(dev), w.to(dev), h, w_)
    dw = torch.zeros_like(w, device=dev)
    op.wgrad(x.to(dev), gz.to(dev), dw, None, h, w_, accumulate=False)
    assert rel_err(y, z) < 1e-4 and rel_err(dx, xr.grad) < 1e-4 and rel_err(dw, wr.grad) < 1e-4
    print("fallback launches for this geometry:", lib.pcuda_fallback_count() - before)


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("n,cin,cout,h,w", [(8, 64, 32, 128, 128), (8, 128, 64, 64, 96), (12, 256, 128, 32, 64), (2, 16, 8, 20, 24)])
def test_dgrad_with_the_2x2_fold_in_its_epilogue(dev, prec, n, cin, cout, h, w):
    """The data gradient of an up-convolution (nearest x2 folded into the forward's addressing, unet.py:111-112) written at
    the stored resolution -- dgrad + upsample2_bwd in one kernel -- against the two-kernel form and the CPU reference, with
    and without the BatchNorm-backward reduce riding along; the last geometry has no folding plan (fallback).  (Batch sizes:
    the first dispatch assertion of round 5 showed the 128- and 256-row cases at n = 4 / 3 on the eight-wave kernel -- too few
    tiles for the pipelined one -- i.e. on the two-kernel fallback; they now have the > 320 (tile, co-tile) items the folding
    kernel's plan needs.)"""
    from pointcloududa_amd import kernels as K
    K.set_precision(prec)
    try:
        rng = np.random.default_rng(cin + h)
        wt = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32))
        gz = torch.from_numpy(rng.normal(0, 1, (n, cout, h, w)).astype(np.float32))
        x = torch.zeros(n, cin, h // 2, w // 2, requires_grad=True)
        F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, None, padding=1).backward(gz)
        op = K.ConvOp(cin, cout, 3, pad=1, in_up=True)
        got = op.dgrad_fold(gz.to(dev), wt.to(dev), h, w)
        if (h, w) != (20, 24):      # (the last geometry has no folding plan: dgrad + upsample2_bwd)
            # (round 6: in bf16x3 mode the anti-phase kernel takes these data gradients -- rows = cin a multiple of 64, whole
            #  32 x 8 tiles --, with the same fold in its epilogue)
            want = "| conv3ap+fold" if prec == "bf16x3" else "| igemm_pipe+fold"
            assert K.last_kernel().endswith(want), K.last_kernel()
        two = K.upsample2_bwd(op.dgrad(gz.to(dev), wt.to(dev), h, w))
        assert rel_err(got, x.grad) < TOL[prec] and rel_err(got, two) < 1e-5
        # with the reduce: partial sums of (g, g * a_hat) over the folded gradient
        a = torch.from_numpy(rng.normal(0, 1, (n, cin, h // 2, w // 2)).astype(np.float32)).to(dev)
        st = K.BNState()
        st.mean = torch.from_numpy(rng.normal(0, 0.3, (cin,)).astype(np.float32)).to(dev)
        st.invstd = torch.from_numpy(rng.uniform(0.5, 2.0, (cin,)).astype(np.float32)).to(dev)
        got2, red = op.dgrad_fold(gz.to(dev), wt.to(dev), h, w, bnred=(a, st))
        assert torch.equal(got2, got)
        gd = got.double()
        want1 = gd.sum((0, 2, 3))
        want2 = (gd * ((a.double() - st.mean.double()[None, :, None, None]) * st.invstd.double()[None, :, None, None])).sum((0, 2, 3))
        if red is not None:
            part, nt = red
            tot = part[:nt].double().sum(0)
            assert rel_err(tot[:, 0], want1) < 1e-4 and rel_err(tot[:, 1], want2) < 1e-4
    finally:
        K.set_precision("bf16x3")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("n,cin,cout,h,w,k,stride,pad,fused", [
    (4, 64, 128, 129, 129, 4, 2, 2, True),      # the discriminators' layers (GAN.py:97-108), odd maps, paired column classes
    (3, 128, 256, 65, 65, 4, 2, 2, True),
    (2, 256, 512, 33, 33, 4, 2, 2, True),
    (2, 64, 128, 113, 113, 4, 2, 2, True),      # the 224 x 224 input's second layer
    (2, 32, 48, 40, 36, 4, 2, 1, True),         # even maps, another padding
    (2, 40, 24, 21, 19, 3, 1, 1, True),         # stride 1 on rows that are no multiple of 4: plain epilogue
    (2, 64, 64, 32, 32, 3, 1, 1, True),         # few tiles: the eight-wave kernel (plain epilogue); in bf16x3 mode the
                                                # anti-phase kernel takes this layer's data gradient (round 6): not fused there
    (88, 64, 64, 32, 32, 3, 1, 1, False),       # transposed-epilogue plan: not taken, the two-kernel form runs
])
def test_dgrad_with_the_leaky_relu_backward_in_its_epilogue(dev, prec, n, cin, cout, h, w, k, stride, pad, fused):
    """dgrad * (a > 0 ? 1 : slope) in one kernel (pcuda_conv2d_dgrad_lrelu) equals pcuda_conv2d_dgrad followed by
    pcuda_lrelu_bwd BIT FOR BIT (the same accumulators, one more multiply) and the CPU reference within the precision's
    tolerance; plans the fused epilogue does not cover report UNSUPPORTED and the wrapper runs the two kernels."""
    import ctypes as C
    from pointcloududa_amd import _lib as L, kernels as K
    K.set_precision(prec)
    if (prec == "bf16x3" and k == 3 and stride == 1 and pad == 1 and cin % 64 == 0 and cout % 16 == 0 and h % 8 == 0 and
            w % 32 == 0 and (n * (h // 8) * (w // 32)) % 2 == 0 and n * (h // 8) * (w // 32) // 2 * (cin // 64) >= 192):
        fused = False    # csrc/conv_ap.hip takes the plain data gradient of this layer: dgrad_lrelu reports UNSUPPORTED
    try:
        rng = np.random.default_rng(cin + h + k)
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        wt = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, k, k)).astype(np.float32))
        gz = torch.from_numpy(rng.normal(0, 1, (n, cout, oh, ow)).astype(np.float32))
        a = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w)).astype(np.float32))
        x = torch.zeros(n, cin, h, w, requires_grad=True)
        F.conv2d(x, wt, None, stride=stride, padding=pad).backward(gz)
        want = torch.where(a > 0, x.grad, 0.2 * x.grad)
        op = K.ConvOp(cin, cout, k, stride=stride, pad=pad)
        two = K.lrelu_bwd(op.dgrad(gz.to(dev), wt.to(dev), h, w), a.to(dev), 0.2)
        got = op.dgrad_lrelu(gz.to(dev), wt.to(dev), h, w, a.to(dev), 0.2)
        assert torch.equal(got, two)
        assert rel_err(got, want) < TOL[prec]
        # which path ran: the entry itself on the same operands
        g = op.geom(n, h, w)
        dz = torch.empty(n, cin, h, w, device=dev)
        ad, gd = a.to(dev), gz.to(dev)
        pk = op._packed("dgrad", wt.to(dev), g)
        src, dst = K.make_src(gd), K.make_dst(dz)
        rc = L.lib().pcuda_conv2d_dgrad_lrelu(C.byref(g), K._precision, C.byref(src), pk.data_ptr(), C.byref(dst), ad.data_ptr(),
                                              ad.stride(0), ad.stride(1), 0.2, K._stream())
        assert rc == (0 if fused else L.PCUDA_E_UNSUPPORTED)
        if fused:
            assert torch.equal(dz, two)
    finally:
        K.set_precision("bf16x3")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("n,c1,c2,cout,h,w", [
    (2, 64, 32, 64, 32, 32),        # encoder conv1_2: cat(y, res_prev) -> 1x1 (unet.py:41-47), three 32-channel blocks
    (3, 128, 64, 128, 16, 16),      # two co tiles x two ci tiles
    (2, 48, 32, 40, 12, 8),         # ragged on both sides: 80 input channels (source boundary inside a block), 40 outputs
    (4, 32, 0, 4, 64, 64),          # the classifier: 4 output rows of a 32-row block
    (1, 20, 0, 24, 4, 4),           # one 16-pixel step in all
])
def test_pointwise_layer_weight_gradient_from_global_rows(dev, prec, n, c1, c2, cout, h, w):
    """1x1 / stride-1 layers on whole 16-pixel steps take the NT-GEMM-over-pixels kernel (csrc/conv_wgrad1.hip: operand fragments
    straight from global memory): against the CPU reference with two sources, the lazy-BatchNorm affine on the first one,
    the bias gradient, and accumulate semantics"""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    K.set_precision(prec)
    try:
        rng = np.random.default_rng(c1 + cout + h)
        cin = c1 + c2
        a = torch.from_numpy(rng.normal(0, 1, (n, c1, h, w)).astype(np.float32))
        b = torch.from_numpy(rng.normal(0, 1, (n, c2, h, w)).astype(np.float32)) if c2 else None
        sc = torch.from_numpy(rng.normal(1, 0.2, (c1,)).astype(np.float32))
        sf = torch.from_numpy(rng.normal(0, 0.2, (c1,)).astype(np.float32))
        wt = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 1, 1)).astype(np.float32)).requires_grad_(True)
        bias = torch.zeros(cout, requires_grad=True)
        xa = a * sc[None, :, None, None] + sf[None, :, None, None]
        xin = torch.cat([xa, b], 1) if c2 else xa
        z = F.conv2d(xin, wt, bias)
        gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
        z.backward(gz)
        op = K.ConvOp(cin, cout, 1)
        dw, db = torch.zeros(cout, cin, 1, 1, device=dev), torch.zeros(cout, device=dev)
        src = TA(a.to(dev), sc.to(dev), sf.to(dev))
        fb = K.fallback_count()
        op.wgrad(src, gz.to(dev), dw, db, h, w, x2=b.to(dev) if c2 else None, accumulate=False)
        assert K.last_kernel().startswith("wgrad1 ") and K.fallback_count() == fb, K.last_kernel()      # (the NT-GEMM kernel ran)
        assert rel_err(dw, wt.grad) < TOL[prec] and rel_err(db, bias.grad) < 1e-4
        op.wgrad(src, gz.to(dev), dw, db, h, w, x2=b.to(dev) if c2 else None, accumulate=True)
        assert rel_err(dw, 2 * wt.grad) < TOL[prec] and rel_err(db, 2 * bias.grad) < 1e-4
        # the same bits on a second run (fixed summation order)
        dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
        op.wgrad(src, gz.to(dev), dw2, db2, h, w, x2=b.to(dev) if c2 else None, accumulate=False)
        op.wgrad(src, gz.to(dev), dw2, db2, h, w, x2=b.to(dev) if c2 else None, accumulate=True)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
    finally:
        K.set_precision("bf16x3")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("n,cin,cout,h,w", [
    (2, 64, 32, 64, 64),            # up-convolution of decoder block 1 (unet.py:85: Upsample(x2) -> 3x3): 64 -> 32, stored 32x32
    (2, 128, 64, 32, 64),           # 128 -> 64: four input-channel blocks x two output-channel blocks, two strips
    (3, 48, 40, 12, 96),            # ragged channel blocks, six stored rows (row counts that do not divide by the unroll), three strips
])
def test_register_window_weight_gradient_of_up_convolution(dev, prec, n, cin, cout, h, w):
    """The up-convolutions' weight gradient on the LDS-free kernel (csrc/conv_wgrad3r.hip, UP): the input is read at its STORED
    resolution (a quarter of the upsampled tensor) and doubled in registers: against the CPU reference (nearest x2 -> 3x3) with
    the lazy-BatchNorm affine (zero padding AFTER the affine), bias gradient, accumulate semantics, bit-reproducibility"""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    K.set_precision(prec)
    try:
        rng = np.random.default_rng(cin + cout + h)
        a = torch.from_numpy(rng.normal(0, 1, (n, cin, h // 2, w // 2)).astype(np.float32))
        sc = torch.from_numpy(rng.normal(1, 0.2, (cin,)).astype(np.float32))
        sf = torch.from_numpy(rng.normal(0.5, 0.2, (cin,)).astype(np.float32))
        wt = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32)).requires_grad_(True)
        bias = torch.zeros(cout, requires_grad=True)
        xa = a * sc[None, :, None, None] + sf[None, :, None, None]
        z = F.conv2d(F.interpolate(xa, scale_factor=2, mode="nearest"), wt, bias, padding=1)
        gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
        z.backward(gz)
        op = K.ConvOp(cin, cout, 3, pad=1, in_up=True)
        dw, db = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
        src = TA(a.to(dev), sc.to(dev), sf.to(dev))
        fb = K.fallback_count()
        op.wgrad(src, gz.to(dev), dw, db, h, w, accumulate=False)
        assert K.last_kernel().startswith("wgrad3r ") and " up1 " in K.last_kernel() and K.fallback_count() == fb, K.last_kernel()
        assert rel_err(dw, wt.grad) < TOL[prec] and rel_err(db, bias.grad) < 1e-4
        op.wgrad(src, gz.to(dev), dw, db, h, w, accumulate=True)
        assert rel_err(dw, 2 * wt.grad) < TOL[prec] and rel_err(db, 2 * bias.grad) < 1e-4
        dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
        op.wgrad(src, gz.to(dev), dw2, db2, h, w, accumulate=False)
        op.wgrad(src, gz.to(dev), dw2, db2, h, w, accumulate=True)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
    finally:
        K.set_precision("bf16x3")


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("n,c1,c2,cout,h,w", [
    (2, 32, 32, 32, 64, 64),        # decoder block 1: cat(skip, up) 64 -> 32 (unet.py:116), the shape the kernel runs by default
    (2, 32, 0, 64, 64, 32),         # encoder block 2: 32 -> 64; one 32-pixel strip
    (3, 16, 8, 40, 68, 96),         # ragged channel blocks, rows that do not divide by the unroll of 4, three strips
])
def test_register_window_weight_gradient(dev, prec, n, c1, c2, cout, h, w):
    """3x3 layers with unequal channel counts <= 64 on maps of >= 64 rows take the LDS-free weight-gradient kernel
    (csrc/conv_wgrad3r.hip: operand rows straight from global memory, the nine taps as register shifts of a three-row window):
    against the CPU reference with two sources, the lazy-BatchNorm affine (zero padding AFTER the affine: borders matter),
    bias gradient, accumulate semantics, and bit-reproducibility"""
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    K.set_precision(prec)
    try:
        rng = np.random.default_rng(c1 + cout + h)
        cin = c1 + c2
        a = torch.from_numpy(rng.normal(0, 1, (n, c1, h, w)).astype(np.float32))
        b = torch.from_numpy(rng.normal(0, 1, (n, c2, h, w)).astype(np.float32)) if c2 else None
        sc = torch.from_numpy(rng.normal(1, 0.2, (c1,)).astype(np.float32))
        sf = torch.from_numpy(rng.normal(0.5, 0.2, (c1,)).astype(np.float32))        # a shift far from 0: padding must stay 0
        wt = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32)).requires_grad_(True)
        bias = torch.zeros(cout, requires_grad=True)
        xa = a * sc[None, :, None, None] + sf[None, :, None, None]
        xin = torch.cat([xa, b], 1) if c2 else xa
        z = F.conv2d(xin, wt, bias, padding=1)
        gz = torch.from_numpy(rng.normal(0, 1, z.shape).astype(np.float32))
        z.backward(gz)
        op = K.ConvOp(cin, cout, 3, pad=1)
        dw, db = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
        src = TA(a.to(dev), sc.to(dev), sf.to(dev))
        x2 = b.to(dev) if c2 else None
        fb = K.fallback_count()
        op.wgrad(src, gz.to(dev), dw, db, h, w, x2=x2, accumulate=False)
        assert K.last_kernel().startswith("wgrad3r ") and K.fallback_count() == fb, K.last_kernel()     # (the register-window kernel ran)
        assert rel_err(dw, wt.grad) < TOL[prec] and rel_err(db, bias.grad) < 1e-4
        op.wgrad(src, gz.to(dev), dw, db, h, w, x2=x2, accumulate=True)
        assert rel_err(dw, 2 * wt.grad) < TOL[prec] and rel_err(db, 2 * bias.grad) < 1e-4
        dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
        op.wgrad(src, gz.to(dev), dw2, db2, h, w, x2=x2, accumulate=False)
        op.wgrad(src, gz.to(dev), dw2, db2, h, w, x2=x2, accumulate=True)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
    finally:
        K.set_precision("bf16x3")
