"""Record-form activations (csrc/conv_rec.hip): layout conversions and the LDS-DMA 3x3 convolution against a plain
PyTorch-CPU fp32 reference of the same op (bf16x3 products: 1e-4 of the output scale), through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c,h,w", [(2, 32, 8, 32), (3, 64, 16, 64), (1, 40, 8, 32)])
def test_record_round_trip_and_affine(dev, n, c, h, w):
    from pointcloududa_amd import kernels as K
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(n, c, h, w, generator=g) * 3.0
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rec = K.rec_from_nchw(x.to(dev))
    assert rec.shape == (n, (c + 31) // 32, h, w, 128)
    back = K.rec_to_nchw(rec, c).cpu()
    assert float((back - x).abs().max()) <= 2.0 ** -16 * float(x.abs().max())          # hi + lo: 16 mantissa bits
    rec2 = K.rec_from_nchw(x.to(dev), sc.to(dev), sh.to(dev))
    want = x * sc[None, :, None, None] + sh[None, :, None, None]
    assert rel_err(K.rec_to_nchw(rec2, c), want) < 2e-5
    # the record IS (hi | lo): 32 bf16 high parts, then the 32 residuals, per pixel and chunk
    r16 = rec.cpu().view(torch.int16).view(n, (c + 31) // 32, h, w, 64)
    hi = (r16[..., :32].to(torch.int32) << 16).view(torch.float32)
    x0 = x[:, :32].permute(0, 2, 3, 1)
    assert torch.equal(hi[:, 0], x0.to(torch.bfloat16).to(torch.float32))


@pytest.mark.parametrize("n,cin,cout,h,w,bias,slope", [
    (2, 32, 32, 16, 32, True, 0.01),      # one chunk, one co-tile, one tile across: every tile touches the border
    (2, 32, 32, 32, 96, True, 0.01),      # interior tiles (fast DMA path) next to border tiles
    (1, 64, 32, 8, 32, False, 1.0),       # two chunks, no bias, no activation
    (2, 64, 64, 24, 64, True, 0.01),      # two chunks x two co-tiles (weights re-staged per stage)
    (3, 96, 64, 8, 64, True, 0.2),
])
def test_rconv3_forward_vs_cpu_reference(dev, n, cin, cout, h, w, bias, slope):
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(cin * 7 + cout + h)
    x = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w)).astype(np.float32))
    wt = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32)) if bias else None
    z = F.conv2d(x, wt, b, padding=1)
    ref = F.leaky_relu(z, slope) if slope != 1.0 else z
    xr = K.rec_from_nchw(x.to(dev))
    wp = K.rconv3_pack(wt.to(dev))
    yr, stats, nt = K.rconv3_forward(xr, wp, None if b is None else b.to(dev), slope, cout, want_stats=True)
    y = K.rec_to_nchw(yr, cout)
    assert rel_err(y, ref) < 1e-4
    s = stats.double().sum(0).cpu()
    assert rel_err(s[:, 0], ref.double().sum((0, 2, 3))) < 1e-3 and rel_err(s[:, 1], (ref.double() ** 2).sum((0, 2, 3))) < 1e-3
    # without the statistics: the same records, bit for bit
    yr2, none, _ = K.rconv3_forward(xr, wp, None if b is None else b.to(dev), slope, cout, want_stats=False)
    assert none is None and torch.equal(yr2, yr)
    # the same layer on the NCHW kernel (same hi / lo operands, fp32 accumulation in another order)
    op = K.ConvOp(cin, cout, 3, pad=1)
    y0, _, _ = op.forward(x.to(dev), wt.to(dev), None if b is None else b.to(dev), slope, h, w)
    assert rel_err(y, y0) < 2e-5


def test_rconv3_folded_batchnorm_pad_records(dev):
    """BatchNorm of the producing layer folded into the consumer: scale into the packed weights, shift into the bias, and
    the record of -shift / scale as what the convolution reads outside the image (it cancels the shift term there
    exactly as zero padding of the normalised tensor does)."""
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(3)
    n, cin, cout, h, w = 2, 64, 32, 16, 64
    a = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w)).astype(np.float32))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, (cin,)).astype(np.float32)) * torch.from_numpy(rng.choice([-1.0, 1.0], cin).astype(np.float32))
    sh = torch.from_numpy(rng.normal(0, 0.5, (cin,)).astype(np.float32))
    wt = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32))
    ref = F.leaky_relu(F.conv2d(a * sc[None, :, None, None] + sh[None, :, None, None], wt, b, padding=1), 0.01)
    bias2 = b + (wt * sh[None, :, None, None]).sum((1, 2, 3))                 # the shift through every tap
    padv = (-sh / sc).view(1, cin, 1, 1)
    pad_rec = K.rec_from_nchw(padv.to(dev)).view(-1, 128)                     # [cin / 32][128 B]
    yr, _, _ = K.rconv3_forward(K.rec_from_nchw(a.to(dev)), K.rconv3_pack(wt.to(dev), sc.to(dev)), bias2.to(dev), 0.01, cout,
                                pad_records=pad_rec)
    assert rel_err(K.rec_to_nchw(yr, cout), ref) < 2e-4


# (maps large enough for the four-wave pipelined plan: the eight-wave kernel of the few-tile layers does not stage records)
@pytest.mark.parametrize("n,cin,cout,h,w,up", [(8, 32, 32, 128, 128, False), (6, 64, 64, 128, 96, False), (8, 128, 64, 64, 128, False),
                                                (8, 64, 32, 128, 128, True)])
def test_igemm_record_staging_matches_the_nchw_source(dev, n, cin, cout, h, w, up):
    """igemm_pipe_kernel staging its input from a record tensor (pcuda_src::rec) instead of fp32 NCHW: the LDS image is the
    same hi / lo records, so forward (with statistics) and dgrad equal the NCHW-source launch up to the summation order of the plan; a pad
    record replaces the zero padding; the nearest-x2 fold reads the half-resolution records."""
    from pointcloududa_amd import kernels as K
    rng = np.random.default_rng(cin + cout + h)
    sh, sw = (h // 2, w // 2) if up else (h, w)
    x = torch.from_numpy(rng.normal(0, 1, (n, cin, sh, sw)).astype(np.float32)).to(dev)
    wt = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32)).to(dev)
    op = K.ConvOp(cin, cout, 3, pad=1, in_up=up)
    y0, p0, nt0 = op.forward(x, wt, b, 0.01, h, w, want_stats=True)
    y1, p1, nt1 = op.forward(K.Rec(K.rec_from_nchw(x)), wt, b, 0.01, h, w, want_stats=True)
    # (bit-identical when both launches take the same plan; a few-tile layer's NCHW launch runs on the eight-wave kernel,
    # another summation order)
    assert rel_err(y1, y0) < 1e-5
    assert rel_err(p1[:nt1].double().sum(0), p0[:nt0].double().sum(0)) < 1e-5
    if not up:
        gz = torch.from_numpy(rng.normal(0, 1, (n, cout, h, w)).astype(np.float32)).to(dev)
        d0 = op.dgrad(gz, wt, h, w)
        d1 = op.dgrad(K.Rec(K.rec_from_nchw(gz)), wt, h, w)
        assert rel_err(d1, d0) < 1e-5
        # a pad record: what the convolution reads outside the image (here: 1.5 in every channel)
        padv = torch.full((1, cin, 1, 1), 1.5, device=dev)
        pad = K.rec_from_nchw(padv).view(-1, 128)
        y2, _, _ = op.forward(K.Rec(K.rec_from_nchw(x), pad=pad), wt, None, 1.0, h, w)
        ref = F.conv2d(F.pad(x.cpu(), (1, 1, 1, 1), value=1.5), wt.cpu(), None)
        assert rel_err(y2, ref) < 1e-4
