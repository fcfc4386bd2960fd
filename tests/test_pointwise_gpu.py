"""HBM-bound kernels (BatchNorm train fwd/bwd fused with LeakyReLU, pooling, upsample backward,
dense ops, optimisers) against plain PyTorch-CPU fp32 references, through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _rand(rng, *shape, scale=1.0):
    return torch.from_numpy(rng.normal(0, scale, shape).astype(np.float32))


@pytest.mark.parametrize("shape", [(3, 8, 32, 32), (2, 5, 70, 66), (4, 16, 300), (6, 32)])
@pytest.mark.parametrize("post_relu", [False, True])
def test_bn_train_forward_backward(dev, shape, post_relu):
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(1)
    c = shape[1]
    z = _rand(rng, *shape)
    gamma, beta = _rand(rng, c) * 0.2 + 1, _rand(rng, c) * 0.2
    rm0, rv0 = _rand(rng, c) * 0.1, torch.rand(c) + 0.5
    gy, gy2 = _rand(rng, *shape), _rand(rng, *shape)
    slope = 0.01
    # reference: [z -> a = lrelu(z) -> y = BN(a)]  or  [a -> BN -> relu]
    zr = z.clone().requires_grad_(True)
    g_r, b_r = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    a_ref = zr if post_relu else F.leaky_relu(zr, slope)
    y_ref = F.batch_norm(a_ref, rm, rv, g_r, b_r, True, 0.1, 1e-5)
    if post_relu:
        y_ref = F.relu(y_ref)
    y_ref.backward(gy + gy2)

    a = (z if post_relu else F.leaky_relu(z, slope)).to(dev)
    part, nt, cnt = K.bn_stats(a)
    rmd, rvd = rm0.to(dev), rv0.to(dev)
    st = K.bn_finalize(part, nt, cnt, gamma.to(dev), beta.to(dev), rmd, rvd)
    y = K.bn_apply(a, st, relu=post_relu)
    assert rel_err(y, y_ref) < 2e-5
    assert rel_err(rmd, rm) < 1e-5 and rel_err(rvd, rv) < 1e-5
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    dz = K.bn_backward(gy.to(dev), a, st, gamma.to(dev), dg, db, dy2=gy2.to(dev), post_relu=post_relu,
                       act_slope=slope, accumulate=False)
    assert rel_err(dz, zr.grad) < 5e-5
    assert rel_err(dg, g_r.grad) < 5e-5 and rel_err(db, b_r.grad) < 5e-5


def test_maxpool_upsample_add(dev):
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    rng = np.random.default_rng(2)
    x = _rand(rng, 2, 6, 32, 48)
    sc, sf = _rand(rng, 6) * 0.5 + (-0.2), _rand(rng, 6)          # negative scales too: max does not commute
    xr = (x * sc[None, :, None, None] + sf[None, :, None, None]).requires_grad_(True)
    y_ref = F.max_pool2d(xr, 2)
    g1, g2 = _rand(rng, *y_ref.shape), _rand(rng, *y_ref.shape)
    y_ref.backward(g1 + g2)
    y, idx = K.maxpool2_fwd(TA(x.to(dev), sc.to(dev), sf.to(dev)))
    assert rel_err(y, y_ref) < 1e-6
    dx = K.maxpool2_bwd(g1.to(dev), idx, 32, 48, dy2=g2.to(dev))
    assert rel_err(dx, xr.grad) < 1e-6
    u = _rand(rng, 2, 3, 8, 10).requires_grad_(True)
    up = F.interpolate(u, scale_factor=2, mode="nearest")
    gu = _rand(rng, *up.shape)
    up.backward(gu)
    assert rel_err(K.upsample2_bwd(gu.to(dev)), u.grad) < 1e-6
    ts = [_rand(rng, 1000) for _ in range(4)]
    assert rel_err(K.add_n([t.to(dev) for t in ts]), ts[0] + ts[1] + ts[2] + ts[3]) < 1e-6
    assert rel_err(K.add_n([t.to(dev) for t in ts[:2]]), ts[0] + ts[1]) < 1e-6
    assert rel_err(K.mul(ts[0].to(dev), ts[1].to(dev)), ts[0] * ts[1]) < 1e-6
    a = _rand(rng, 2, 5, 16, 16)
    gy = _rand(rng, 2, 5, 16, 16)
    assert rel_err(K.lrelu_bwd(gy.to(dev), a.to(dev), 0.2), torch.where(a > 0, gy, 0.2 * gy)) < 1e-6
    # odd planes (the discriminators' 129x129 / 65x65 / 33x33 maps): dense tensors go through the flat float4 path,
    # a total that is no multiple of 4 through the scalar one, a second gradient source through either
    for shp in ((4, 3, 33, 33), (4, 8, 17, 17), (1, 3, 9, 9)):
        a2, g2, g3 = _rand(rng, *shp), _rand(rng, *shp), _rand(rng, *shp)
        assert rel_err(K.lrelu_bwd(g2.to(dev), a2.to(dev), 0.2), torch.where(a2 > 0, g2, 0.2 * g2)) < 1e-6
        assert rel_err(K.lrelu_bwd(g2.to(dev), a2.to(dev), 0.2, dy2=g3.to(dev)),
                       torch.where(a2 > 0, g2 + g3, 0.2 * (g2 + g3))) < 1e-6
    db = torch.zeros(5, device=dev)
    K.channel_sum(gy.to(dev), db, accumulate=False)
    assert rel_err(db, gy.sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("shape", [(2, 8, 32, 48), (3, 5, 22, 18), (2, 32, 112, 112), (1, 3, 6, 10)])
@pytest.mark.parametrize("shares", [(False, False), (True, False), (True, True)])
def test_backward_kernels_read_the_pool_scatter_in_place(dev, shape, shares):
    """lrelu_bwd / bn_backward with the gradient arriving through a 2x2 max-pool (pcuda_*_pooled) equal
    maxpool2_bwd followed by the plain kernel BIT FOR BIT: the same sums in the same order, the 4x tensor never written;
    rows that are no multiple of 4 wide take the scalar kernels"""
    from pointcloududa_amd import kernels as K
    two, skip = shares
    n, c, h, w = shape
    rng = np.random.default_rng(40 + h)
    a = _rand(rng, n, c, h, w).to(dev)
    _, idx = K.maxpool2_fwd(_rand(rng, n, c, h, w).to(dev))
    g = _rand(rng, n, c, h // 2, w // 2).to(dev)
    g2 = _rand(rng, n, c, h // 2, w // 2).to(dev) if two else None
    dy = _rand(rng, n, c, h, w).to(dev) if skip else None
    scat = K.maxpool2_bwd(g, idx, h, w, dy2=g2)
    ref = K.lrelu_bwd(scat, a, 0.2, dy2=dy)
    got = K.lrelu_bwd_pooled(g, idx, a, 0.2, g2=g2, dy=dy)
    assert torch.equal(got, ref)
    # (against the definition as well, not only against the sibling kernel)
    want = F.max_unpool2d((g + g2 if two else g).cpu(), (idx.cpu().long() // 2 + 2 * torch.arange(h // 2)[:, None]) * w +
                          idx.cpu().long() % 2 + 2 * torch.arange(w // 2)[None, :], 2, output_size=(h, w))
    want = want + dy.cpu() if skip else want
    assert rel_err(got, torch.where(a.cpu() > 0, want, 0.2 * want)) < 1e-6
    part, nt, cnt = K.bn_stats(a)
    gamma = (_rand(rng, c) * 0.2 + 1).to(dev)
    st = K.bn_finalize(part, nt, cnt, gamma, torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.ones(c, device=dev))
    out = []
    for pooled in (False, True):
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        if pooled:
            dz = K.bn_backward_pooled(g, idx, a, st, gamma, dg, db, g2=g2, dy=dy, act_slope=0.2, accumulate=False)
        else:
            dz = K.bn_backward(scat, a, st, gamma, dg, db, dy2=dy, act_slope=0.2, accumulate=False)
        out.append((dz, dg, db))
    for x, y in zip(*out):      # (the scalar kernels add a plane's elements in another order than the float4 ones)
        assert torch.equal(x, y) if w % 4 == 0 else rel_err(x, y) < 1e-5


def test_dense_ops(dev):
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(3)
    for m, k, n in [(32, 1024, 512), (5, 121, 3), (600, 9, 3), (7, 256, 9)]:
        x, w, b = _rand(rng, m, k), _rand(rng, n, k, scale=0.1), _rand(rng, n)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y_ref = F.linear(xr, wr, br)
        g = _rand(rng, m, n)
        y_ref.backward(g)
        assert rel_err(K.linear_fwd(x.to(dev), w.to(dev), b.to(dev)), y_ref) < 1e-5
        assert rel_err(K.linear_bwd_x(g.to(dev), w.to(dev)), xr.grad) < 1e-5
        dw, db = torch.zeros(n, k, device=dev), torch.zeros(n, device=dev)
        K.linear_bwd_w(g.to(dev), x.to(dev), dw, db, accumulate=False)
        assert rel_err(dw, wr.grad) < 1e-5 and rel_err(db, br.grad) < 1e-5
    t, x = _rand(rng, 4, 3, 3), _rand(rng, 4, 3, 300)
    assert rel_err(K.bmm(t.to(dev), x.to(dev), ta=True), torch.bmm(t.transpose(1, 2), x)) < 1e-5
    assert rel_err(K.bmm(t.to(dev), x.to(dev)), torch.bmm(t, x)) < 1e-5
    assert rel_err(K.bmm(x.to(dev), x.to(dev), tb=True), torch.bmm(x, x.transpose(1, 2))) < 1e-5
    t64, x64 = _rand(rng, 3, 64, 64), _rand(rng, 3, 64, 300)
    assert rel_err(K.bmm(t64.to(dev), x64.to(dev), ta=True), torch.bmm(t64.transpose(1, 2), x64)) < 1e-5
    h = _rand(rng, 3, 70, 300)
    h[0, 0, 5] = h[0, 0, 200] = 9.0                                # tie: first occurrence wins
    v, idx = K.max_points_fwd(h.to(dev))
    vr, ir = h.max(dim=2)
    assert torch.equal(v.cpu(), vr) and int(idx[0, 0]) == 5
    g = _rand(rng, 3, 70)
    ref = torch.zeros_like(h).scatter_(2, idx.cpu().long()[..., None], g[..., None])
    assert rel_err(K.max_points_bwd(g.to(dev), idx, 300), ref) < 1e-7


def test_fused_optimisers(dev):
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(4)
    p0, steps = _rand(rng, 5000), 3
    grads = [_rand(rng, 5000) for _ in range(steps)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.9, 0.99))
    p, m, v = p0.to(dev), torch.zeros(5000, device=dev), torch.zeros(5000, device=dev)
    for i, g in enumerate(grads):
        pr.grad = g.clone(); opt.step()
        K.adam_step(p, g.to(dev), m, v, 1e-3, 0.9, 0.99, 1e-8, 0.0, i + 1)
        assert rel_err(p, pr.detach()) < 1e-6
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=2.5e-5, momentum=0.99, weight_decay=0.0005)
    p, buf = p0.to(dev), torch.zeros(5000, device=dev)
    for i, g in enumerate(grads):
        pr.grad = g.clone(); opt.step()
        K.sgd_step(p, g.to(dev), buf, 2.5e-5, 0.99, 0.0005, i == 0)
        assert rel_err(p, pr.detach()) < 1e-6
    # grad_scale = 1/world (data-parallel mean of summed gradients)
    p2, m2, v2 = p0.to(dev), torch.zeros(5000, device=dev), torch.zeros(5000, device=dev)
    K.adam_step(p2, (grads[0] * 4).to(dev), m2, v2, 1e-3, 0.9, 0.99, 1e-8, 0.0, 1, grad_scale=0.25)
    p3, m3, v3 = p0.to(dev), torch.zeros(5000, device=dev), torch.zeros(5000, device=dev)
    K.adam_step(p3, grads[0].to(dev), m3, v3, 1e-3, 0.9, 0.99, 1e-8, 0.0, 1)
    assert rel_err(p2, p3) < 1e-6


@pytest.mark.parametrize("c,h,w,k,s,p,d", [(4, 64, 64, 4, 2, 2, 1), (5, 33, 47, 4, 2, 2, 1), (1, 20, 20, 3, 1, 1, 1),
                                            (3, 17, 19, 3, 2, 1, 2)])
def test_unfold_taps_is_exactly_im2col(dev, c, h, w, k, s, p, d):
    """pcuda_unfold_taps against F.unfold (same (c, ky, kx) channel order): a copy, so bit-exact, on a strided input"""
    from pointcloududa_amd import kernels as K
    big = torch.randn(3, c + 2, h, w, device=dev)
    x = big[:, 1:c + 1]                                        # plane strides differ from the dense ones
    u = K.unfold_taps(x, k, s, p, d)
    ref = F.unfold(x.contiguous(), k, dilation=d, padding=p, stride=s).view(3, c * k * k, u.shape[2], u.shape[3])
    assert torch.equal(u, ref)


def test_discriminator_first_layer_folded_equals_direct(dev, monkeypatch):
    """GAN.py:96 layer as unfold + 1x1 MFMA layer against the same layer on the 16-tap kernel: same products, other
    summation order -> outputs and input gradients within 1e-4 (weight gradients share one kernel)"""
    from pointcloududa_amd.networks import UncertaintyDiscriminator
    from pointcloududa_amd.utils import loss as L
    torch.manual_seed(3)
    a = UncertaintyDiscriminator(in_channel=4).to(dev)
    monkeypatch.setenv("PCUDA_NOFOLD", "1")
    b = UncertaintyDiscriminator(in_channel=4).to(dev)
    monkeypatch.delenv("PCUDA_NOFOLD")
    assert a._fold1 is not None and b._fold1 is None
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 4, 96, 96, device=dev)
    outs = []
    for m in (a, b):
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        L.bce_logits_const(y, 0.0).backward()
        outs.append((y.detach(), xi.grad, [p.grad.clone() for p in m.parameters()]))
    assert rel_err(outs[0][0], outs[1][0]) < 1e-4
    assert rel_err(outs[0][1], outs[1][1]) < 1e-4
    for ga, gb in zip(outs[0][2], outs[1][2]):
        assert rel_err(ga, gb) < 1e-4
