"""Whole-network parity: the HIP-backed modules against golden vectors produced by the REFERENCE
modules (tests/golden/*.npz from oracle/make_golden.py).  Parameters are regenerated from the same
numpy seeds and load_state_dict()-ed, so state_dict key compatibility is exercised too.

Tolerances (bf16x3 parity mode): 1e-3 of the tensor's scale, the north-star figure; observed errors
are one to two orders of magnitude below that."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, rel_err

pytestmark = pytest.mark.gpu


def _load(mod, params, dev):
    mod.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
    return mod.to(dev).train()


def _strided(t, n=4096):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n]


def _check_grads(mod, g, tol, tol_elem=None):
    """per-parameter gradient norms (and samples) against the reference's.  Parameters whose reference
    gradient is rounding noise (a bias feeding a BatchNorm has an exactly-zero true gradient) are only
    required to be equally negligible: the floor is 1e-3 of the model's total gradient norm.

    Where the golden file carries the reference's OWN gradient spread (``gspread/``, ``gnspread/``: the reference
    re-run with 2^-17 relative noise on every convolution output, oracle/make_golden.py), the bound of a parameter is
    max(2e-2, 3 x that spread) instead of the flat ``tol`` / ``tol_elem``: tight where the reference is
    well-conditioned, and no tighter than the reference can reproduce itself where it is not."""
    worst = 0.0
    tol_elem = tol if tol_elem is None else tol_elem
    total = sum(float(g[k]) ** 2 for k in g.files if k.startswith("gnorm/")) ** 0.5
    floor = 1e-3 * total
    for k, p in mod.named_parameters():
        if "gnone/" + k in g:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        if "gnorm/" + k not in g:
            continue
        ref_norm = float(g["gnorm/" + k])
        got_norm = float(p.grad.double().norm())
        # (capped: a parameter whose reference spread is 20 % is not excused beyond 25 % -- a real backward bug confined to
        # such a layer must not pass; the shared-routing twins of test_backward_exact_gpu.py hold every one of them to 1e-4)
        tn = max(2e-2, min(3.0 * float(g["gnspread/" + k]), 0.25)) if "gnspread/" + k in g else tol
        te = max(2e-2, min(3.0 * float(g["gspread/" + k]), 0.25)) if "gspread/" + k in g else tol_elem
        assert abs(got_norm - ref_norm) <= tn * ref_norm + floor, (k, got_norm, ref_norm, tn)
        if ref_norm < 10 * floor:
            continue
        if "g/" + k in g:
            e = rel_err(p.grad, g["g/" + k]); worst = max(worst, e / te)
            assert e < te, (k, e, te)
        elif "gs/" + k in g:
            e = rel_err(_strided(p.grad, len(g["gs/" + k])), g["gs/" + k]); worst = max(worst, e / te)
            assert e < te, (k, e, te)
    return worst


@pytest.mark.parametrize("tag,cfg_kw,softmax", [
    ("seg_small", dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9), False),
    ("seg_small_3ch_nopoint", dict(filters=8, in_channels=3, n_class=5, pointnet=False), True),
    ("seg_full256", dict(filters=32, in_channels=1, n_class=4, pointnet=True, fc_inch=121), False),
    # stand-in for BASELINE config 5's input size (no DeepLab exists in the reference): the reference's segmenter at 512x512
    ("seg_full512", dict(filters=32, in_channels=1, n_class=4, pointnet=True, fc_inch=729), False),
    # the reference's real MS-CMRSeg shape (train_mscmrseg.py:412-414): 224x224x3, constructor defaults in_channels=3, fc_inch=81:
    # 14x14 bottleneck (dilation 8 on a 14-wide map), 28 / 56 / 112-wide levels, 9x9 head output
    ("seg_full224", dict(filters=32, in_channels=3, n_class=4, pointnet=True, fc_inch=81), False),
    # extpn=True: two extra 3x3 convolutions in front of the point head (unet.py:81-83,90-92)
    ("seg_small_extpn", dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9, extpn=True), False),
    # batchnorm=False (unet.py:25,29): conv -> LeakyReLU -> conv -> LeakyReLU blocks; a constructor variant the reference
    # offers and its scripts never use
    ("seg_small_nobn", dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9, batchnorm=False), False),
])
def test_segmenter_vs_reference_golden(dev, tag, cfg_kw, softmax):
    """forward + backward under the reference's own supervised loss (train_mscmrseg.py:202-209 /
    train_mmwhs.py:212-218).  Outputs and loss: 1e-3 (north-star); observed ~1e-4.
    Gradients: per parameter max(2e-2, 3 x the reference's own spread under 2^-17 relative noise on every convolution output) -- see _check_grads;
    the golden files record spreads of up to 20 % (encoder4, 16x16 level).  The reference network's gradients are discontinuous
    in its activations (max-pool argmax, LeakyReLU sign): scripts/gradient_conditioning.py shows that
    1e-6 relative noise on the FIRST conv output of the fp32 oracle already moves some weight gradients
    by 1-2 % (one routing flip at the 8x8 level is 1/128 of a sum), so no two fp32-class
    implementations agree better than ~1e-2 there.  Kernel-level gradient exactness (1e-4 / 1e-5 on
    identical inputs) is covered by test_conv_gpu.py, test_pointwise_gpu.py and
    test_block_backward_exact below."""
    from oracle import nets as ON
    from oracle.synth import synth_batch
    from pointcloududa_amd.networks import Segmentation_model_Point
    from pointcloududa_amd.utils import loss as L
    g = np.load(os.path.join(GOLD, tag + ".npz"))
    seed, b, hw = int(g["seed"]), int(g["b"]), int(g["hw"])
    cfg = ON.SegCfg(**cfg_kw)
    params = ON.make_params(ON.seg_param_shapes(cfg), seed)
    model = _load(Segmentation_model_Point(**cfg_kw), params, dev)
    assert list(model.state_dict().keys()) == list(params.keys())
    img, mask, vert, _, _ = synth_batch(b, cfg.in_channels, cfg.n_class, hw, seed=seed + 1)
    x = torch.from_numpy(img).to(dev).requires_grad_(True)
    logits, none, verts = model(x)
    assert none is None
    one = torch.ones((), device=dev)
    l_main, l_jac = L.seg_loss(logits, torch.from_numpy(mask).to(dev), "softmax" if softmax else "sigmoid")
    seeds, grads, total = [l_main, l_jac], [one, one], float(l_main.detach()) + float(l_jac.detach())
    if cfg.pointnet:
        l_pt = L.batch_NN_loss(verts, torch.from_numpy(vert).to(dev))
        seeds.append(l_pt); grads.append(one); total += float(l_pt.detach())
    torch.autograd.backward(seeds, grads)
    assert abs(total - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    tdx = max(2e-2, 3.0 * float(g["dxspread"]))
    if "logits" in g:
        assert rel_err(logits, g["logits"]) < 1e-3
        assert rel_err(x.grad, g["dx"]) < tdx
    else:
        assert rel_err(_strided(logits), g["logits_s"]) < 1e-3
        assert rel_err(_strided(x.grad), g["dx_s"]) < tdx
    if cfg.pointnet:
        assert rel_err(verts, g["verts"]) < 1e-3
    _check_grads(model, g, 5e-2, 1e-1)
    sd = model.state_dict()
    for k in params:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(sd[k], g["bn/" + k]) < 1e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == 1


@pytest.mark.parametrize("tag,inch,ext,hw", [("disc_small", 4, False, 64), ("disc_ext_small", 5, True, 128)])
def test_discriminator_vs_reference_golden(dev, tag, inch, ext, hw):
    from oracle import nets as ON
    from pointcloududa_amd.networks import UncertaintyDiscriminator
    from pointcloududa_amd.utils import loss as L
    g = np.load(os.path.join(GOLD, tag + ".npz"))
    seed = int(g["seed"])
    params = ON.make_params(ON.disc_param_shapes(inch, ext), seed, std=0.02)
    model = _load(UncertaintyDiscriminator(in_channel=inch, ext=ext), params, dev)
    rng = np.random.default_rng(seed + 1)
    x = torch.from_numpy(rng.normal(0, 1, (2, inch, hw, hw)).astype(np.float32)).to(dev).requires_grad_(True)
    d = model(x)
    loss = L.bce_logits_const(d, 1.0)
    loss.backward()
    assert rel_err(d, g["out"]) < 1e-3
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
    assert rel_err(_strided(x.grad), g["dx_s"]) < 1e-2
    assert abs(float(x.grad.double().norm()) - float(g["dx_norm"])) < 1e-2 * float(g["dx_norm"])
    _check_grads(model, g, 1e-2)
    # frozen discriminator (adversarial phase): input gradient only, no parameter gradients
    model.zero_grad(set_to_none=True)
    model.requires_grad_(False)
    x2 = x.detach().clone().requires_grad_(True)
    L.bce_logits_const(model(x2), 1.0).backward()
    assert rel_err(_strided(x2.grad), g["dx_s"]) < 1e-2
    assert all(p.grad is None for p in model.parameters())


@pytest.mark.parametrize("tag,ft,ext", [("pncls", False, False), ("pncls_ft_ext", True, True)])
def test_pointnet_cls_vs_reference_golden(dev, tag, ft, ext):
    from oracle import nets as ON
    from pointcloududa_amd.networks import PointNetCls
    from pointcloududa_amd.utils import loss as L
    g = np.load(os.path.join(GOLD, tag + ".npz"))
    seed, b = int(g["seed"]), int(g["b"])
    params = ON.make_params(ON.pointnet_cls_param_shapes(ft, ext=ext), seed)
    model = _load(PointNetCls(feature_transform=ft, ext=ext, drop=0.0), params, dev)
    assert list(model.state_dict().keys()) == list(params.keys())
    rng = np.random.default_rng(seed + 1)
    x = torch.from_numpy(rng.random((b, 3, 300), dtype=np.float32)).to(dev).requires_grad_(True)
    y, trans, trans_feat = model(x)
    loss = L.bce_logits_const(y, 0.0)
    loss.backward()
    # Outputs at the north-star bar, flat: 1e-3 of the tensor's scale.  (Rounds 1-2 ran the k=1 Conv1d layers on the
    # bf16x3 image convolution; BatchNorm1d over 12-16 clouds right behind a max over 300 points turned its 2^-17 noise
    # into 0.5-2.4e-3 of y, and the test had to fall back on the reference's own spread.  They now run in exact fp32
    # on the matrix cores, csrc/conv1d_f32.hip.)  The input gradient and the per-parameter gradients go through ReLU /
    # argmax routing and keep max(2e-2, 3 x the reference's own spread); test_backward_exact_gpu.py holds them to 1e-4
    # with the routing shared.
    sp = lambda k: float(g["spread/" + k])
    e_y, e_t, e_l = rel_err(y, g["y"]), rel_err(trans, g["trans"]), abs(float(loss.detach()) - float(g["loss"]))
    e_tf = rel_err(_strided(trans_feat), g["trans_feat_s"]) if ft else 0.0
    print("%s: y %.2e trans %.2e trans_feat %.2e loss %.2e (reference's own spread under 2^-17 noise: y %.2e)"
          % (tag, e_y, e_t, e_tf, e_l, sp("y")))
    assert e_y < 1e-3 and e_t < 1e-3 and e_tf < 1e-3 and e_l < 1e-4, (e_y, e_t, e_tf, e_l)
    assert rel_err(x.grad, g["dx"]) < max(2e-2, 3 * sp("dx")), (rel_err(x.grad, g["dx"]), sp("dx"))
    assert (trans_feat is not None) == ft
    _check_grads(model, g, 1e-1, 3.5e-1)
    sd = model.state_dict()
    for k in params:
        if "bn/" + k in g:
            assert rel_err(sd[k], g["bn/" + k]) < 1e-3, k


def test_segmentation_model_feature_dis_vs_reference_golden(dev):
    """Segmentation_model(feature_dis=True) (unet.py:139-162): a second 1x1 classifier on the bottleneck output, returned
    as the second element of forward's tuple; golden from the reference module (filters = 32: classifier2 is hard-wired to
    512 input channels), loss = the supervised loss on the logits + 0.5 mean(output2^2)."""
    from oracle import nets as ON
    from oracle.synth import synth_batch
    from pointcloududa_amd.networks.unet import Segmentation_model
    from pointcloududa_amd.utils import loss as L
    g = np.load(os.path.join(GOLD, "seg_featdis.npz"))
    cfg = ON.SegCfg(filters=32, in_channels=1, n_class=4, pointnet=False, feature_dis=True)
    params = ON.make_params(ON.seg_param_shapes(cfg), int(g["seed"]))
    model = _load(Segmentation_model(filters=32, in_channels=1, n_class=4, feature_dis=True), params, dev)
    assert list(model.state_dict().keys()) == list(params.keys())
    img, mask, _, _, _ = synth_batch(int(g["b"]), 1, 4, int(g["hw"]), seed=int(g["seed"]) + 1)
    x = torch.from_numpy(img).to(dev).requires_grad_(True)
    logits, out2, none = model(x)
    assert none is None and model(x, features_out=False).shape == logits.shape
    l_main, l_jac = L.seg_loss(logits, torch.from_numpy(mask).to(dev), "sigmoid")
    l2 = 0.5 * (out2 * out2).mean()      # (test-side loss on the second head: plain torch ops feeding the HIP node's backward)
    (l_main + l_jac + l2).backward()
    assert rel_err(logits, g["logits"]) < 1e-3 and rel_err(out2, g["out2"]) < 1e-3
    assert abs(float((l_main + l_jac + l2).detach()) - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    _check_grads(model, g, 5e-2, 1e-1)
    for k in ("classifier2.weight", "classifier2.bias"):
        assert rel_err(_strided(dict(model.named_parameters())[k].grad, 512), g["gs/" + k]) < 2e-2, k


def test_backward_through_eval_mode_batchnorm(dev):
    """Frozen-BatchNorm fine-tuning: ``model.eval()`` puts nn.BatchNorm2d / BatchNorm1d on their running statistics
    (unet.py:26,30; PointNetCls.py:32-36) and the reference's autograd differentiates through that fixed affine.  The
    HIP modules' backward passes do the same (bn_bwd_finalize with frozen statistics): every gradient against the CPU
    restatement in eval mode.  (No batch statistics couple the samples here, so the comparison is well conditioned: 2e-3
    of each tensor's scale, the remaining differences being LeakyReLU / max-pool routing flips of independent passes.)"""
    from oracle import losses as OL
    from oracle import nets as ON
    from oracle.synth import synth_batch
    from pointcloududa_amd.networks import PointNetCls, Segmentation_model_Point
    from pointcloududa_amd.utils import loss as L
    cfg_kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    cfg = ON.SegCfg(**cfg_kw)
    params = ON.make_params(ON.seg_param_shapes(cfg), 1600)
    rng = np.random.default_rng(1601)
    for k in params:      # non-trivial running statistics
        if k.endswith("running_mean"):
            params[k] = torch.from_numpy(rng.normal(0, 0.2, tuple(params[k].shape)).astype(np.float32))
        if k.endswith("running_var"):
            params[k] = torch.from_numpy(rng.uniform(0.5, 1.5, tuple(params[k].shape)).astype(np.float32))
    model = _load(Segmentation_model_Point(**cfg_kw), params, dev).eval()
    img, mask, vert, _, _ = synth_batch(2, 1, 4, 128, seed=1602)
    x = torch.from_numpy(img).to(dev).requires_grad_(True)
    logits, _, verts = model(x)
    one = torch.ones((), device=dev)
    l_main, l_jac = L.seg_loss(logits, torch.from_numpy(mask).to(dev), "sigmoid")
    torch.autograd.backward([l_main, l_jac, L.batch_NN_loss(verts, torch.from_numpy(vert).to(dev))], [one, one, one])
    p2 = {k: (v.clone().requires_grad_(True) if ON.is_trainable(k) else v.clone()) for k, v in params.items()}
    xo = torch.from_numpy(img).requires_grad_(True)
    lo2, ve2 = ON.seg_forward(p2, xo, cfg, training=False)
    m2, j2 = OL.seg_loss_sigmoid(lo2, torch.from_numpy(mask))
    (m2 + j2 + OL.batch_nn_loss(ve2, torch.from_numpy(vert))).backward()
    assert rel_err(logits, lo2) < 1e-3 and rel_err(verts, ve2) < 1e-3
    sd = model.state_dict()
    worst = 0.0
    total = sum(float(v.grad.double().norm()) ** 2 for k, v in p2.items() if ON.is_trainable(k) and v.grad is not None) ** 0.5
    for k, p in model.named_parameters():
        g = p2[k].grad
        if g is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        if float(g.double().norm()) < 1e-4 * total:
            continue
        e = rel_err(p.grad, g); worst = max(worst, e)
        assert e < 2e-2, (k, e)
    # (the INPUT gradient is not compared here: with 4 filters and no normalisation of the gradient scale in eval mode it is a
    # sparse map in which single LeakyReLU / max-pool flips between two independent forward passes move 10-30 % of the
    # L2 norm; tests/test_backward_exact_gpu.py holds it -- and every gradient above -- to 1e-4 in eval mode with the
    # routing shared)
    e_dx = float((x.grad.cpu().double() - xo.grad.double()).norm() / xo.grad.double().norm())
    for k in params:      # eval mode leaves the running statistics alone
        if k.endswith(("running_mean", "running_var")):
            assert torch.equal(sd[k].cpu(), params[k]), k
    # the point-cloud discriminator in eval mode (BatchNorm1d on running statistics, dropout off)
    pp = ON.make_params(ON.pointnet_cls_param_shapes(True, ext=True), 1610)
    for k in pp:
        if k.endswith("running_mean") and ".in" not in k and not k.startswith("in"):
            pp[k] = torch.from_numpy(rng.normal(0, 0.2, tuple(pp[k].shape)).astype(np.float32))
        if k.endswith("running_var") and ".in" not in k and not k.startswith("in"):
            pp[k] = torch.from_numpy(rng.uniform(0.5, 1.5, tuple(pp[k].shape)).astype(np.float32))
    d4 = _load(PointNetCls(feature_transform=True, ext=True, drop=0.3), pp, dev).eval()
    xn = rng.random((6, 3, 300), dtype=np.float32)
    xp = torch.from_numpy(xn).to(dev).requires_grad_(True)
    y, _, _ = d4(xp)
    L.bce_logits_const(y, 0.0).backward()
    q2 = {k: (v.clone().requires_grad_(True) if ON.is_trainable(k) else v.clone()) for k, v in pp.items()}
    xq = torch.from_numpy(xn).requires_grad_(True)
    y2, _, _ = ON.pointnet_cls_forward(q2, xq, feature_transform=True, ext=True, drop=0.3, training=False)
    OL.bce_logits_const(y2, 0.0).backward()
    assert rel_err(y, y2) < 1e-3
    tot4 = sum(float(v.grad.double().norm()) ** 2 for k, v in q2.items() if ON.is_trainable(k) and v.grad is not None) ** 0.5
    worst4 = 0.0
    for k, p in d4.named_parameters():
        g = q2[k].grad
        if g is None or float(g.double().norm()) < 1e-4 * tot4:
            continue
        e = rel_err(p.grad, g); worst4 = max(worst4, e)
        assert e < 2e-2, (k, e)
    print("eval-mode backward: worst segmenter gradient error %.2e (dx, L2, not asserted: %.2e), worst PointNetCls %.2e" % (worst, e_dx, worst4))


def test_block_backward_exact(dev):
    """One decoder block (concat -> conv -> LeakyReLU -> BN -> conv -> LeakyReLU -> BN) backward on the HIP
    kernels against PyTorch-CPU autograd fed the IDENTICAL block inputs: with the routing decisions
    shared, every gradient agrees to 1e-4 (observed ~1e-5)."""
    import torch.nn.functional as F
    from oracle import nets as ON
    from pointcloududa_amd import kernels as K
    from pointcloududa_amd.kernels import TA
    from pointcloududa_amd.networks import Segmentation_model_Point
    cfg_kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    params = ON.make_params(ON.seg_param_shapes(ON.SegCfg(**cfg_kw)), 100)
    m = _load(Segmentation_model_Point(**cfg_kw), params, dev)
    rng = np.random.default_rng(101)
    x = torch.from_numpy(rng.random((2, 1, 128, 128), dtype=np.float32)).to(dev)
    P = m._tensor_dict()
    _, _, S = m._engine.forward(P, x, True)
    blk = "decoder.decoder2_1"
    xs, x2s, a0, st0, a1, st1 = S[blk]
    p = {k: v.clone() for k, v in params.items()}
    skip = (xs.t * xs.scale[None, :, None, None] + xs.shift[None, :, None, None]).cpu()
    xin = torch.cat([skip, x2s.cpu()], 1).requires_grad_(True)
    w0, w3 = p[blk + ".0.weight"].requires_grad_(True), p[blk + ".3.weight"].requires_grad_(True)
    # share the LeakyReLU routing: the reference's pre-activations take the VALUES the HIP forward produced
    # (inverted from the stored post-activations) while autograd still sees the convolutions -- otherwise an
    # element within ~1e-5 of zero may take the other branch and move its gradient by a factor of 100
    z0_hip = torch.where(a0 > 0, a0, a0 / 0.01).cpu()
    z1_hip = torch.where(a1 > 0, a1, a1 / 0.01).cpu()
    z0p = F.conv2d(xin, w0, p[blk + ".0.bias"], padding=1)
    assert rel_err(z0_hip, z0p) < 1e-4
    z0p = z0p + (z0_hip - z0p).detach(); z0p.retain_grad()
    y0 = F.batch_norm(F.leaky_relu(z0p, 0.01), None, None, p[blk + ".2.weight"], p[blk + ".2.bias"], True); y0.retain_grad()
    z1p = F.conv2d(y0, w3, p[blk + ".3.bias"], padding=1)
    assert rel_err(z1_hip, z1p) < 1e-4
    z1p = z1p + (z1_hip - z1p).detach(); z1p.retain_grad()
    y1 = F.batch_norm(F.leaky_relu(z1p, 0.01), None, None, p[blk + ".5.weight"], p[blk + ".5.bias"], True)
    gy = torch.from_numpy(rng.normal(0, 1, y1.shape).astype(np.float32))
    y1.backward(gy)
    dg, db = torch.zeros(4, device=dev), torch.zeros(4, device=dev)
    op3, op0 = m._engine.ops[blk + ".3"], m._engine.ops[blk + ".0"]
    dz1 = K.bn_backward(gy.to(dev), a1, st1, P[blk + ".5.weight"], dg, db, act_slope=0.01, accumulate=False)
    assert rel_err(dz1, z1p.grad) < 1e-4
    dw3, db3 = torch.zeros(4, 4, 3, 3, device=dev), torch.zeros(4, device=dev)
    op3.wgrad(TA(a0, st0.scale, st0.shift), dz1, dw3, db3, 128, 128, accumulate=False)
    assert rel_err(dw3, w3.grad) < 1e-4
    d_y0 = op3.dgrad(dz1, P[blk + ".3.weight"], 128, 128)
    assert rel_err(d_y0, y0.grad) < 1e-4
    dz0 = K.bn_backward(d_y0, a0, st0, P[blk + ".2.weight"], dg, db, act_slope=0.01, accumulate=False)
    assert rel_err(dz0, z0p.grad) < 1e-4
    dw0, db0 = torch.zeros(4, 8, 3, 3, device=dev), torch.zeros(4, device=dev)
    op0.wgrad(xs, dz0, dw0, db0, 128, 128, x2=x2s, accumulate=False)
    assert rel_err(dw0, w0.grad) < 1e-4
    d1, d2 = torch.empty(2, 4, 128, 128, device=dev), torch.empty(2, 4, 128, 128, device=dev)
    op0.dgrad(dz0, P[blk + ".0.weight"], 128, 128, dx=d1, dx2=d2)
    assert rel_err(torch.cat([d1, d2], 1), xin.grad) < 1e-4


def test_discriminator_replay_is_bit_identical_to_a_second_forward(dev):
    """The train step's d1 / d2 update replays the target batch's activations from the frozen adversarial pass instead of
    running the network on it again (same weights, same input values): outputs and every parameter gradient must be the
    bits a second forward + backward produces, and the frozen pass's own input gradient must be untouched."""
    from oracle import nets as ON
    from pointcloududa_amd.networks import UncertaintyDiscriminator
    from pointcloududa_amd.utils import loss as L
    params = ON.make_params(ON.disc_param_shapes(4, False), 77, std=0.02)
    rng = np.random.default_rng(78)
    x_np = rng.normal(0, 1, (3, 4, 96, 96)).astype(np.float32)

    def grads(replay):
        m = _load(UncertaintyDiscriminator(4), params, dev)
        x = torch.from_numpy(x_np).to(dev).requires_grad_(True)
        m.requires_grad_(False)                                   # phase 2: frozen, gradient to the input only
        d = m.forward_cached(x) if replay else m(x)
        L.bce_logits_const(d, 1.0, weight=0.01).backward()
        m.requires_grad_(True)                                    # phases 3-4: the update on the same (detached) values
        d2 = m.replay() if replay else m(x.detach())
        assert torch.equal(d2, d)
        L.bce_logits_const(d2, 0.0).backward()
        m.drop_cache()
        return [x.grad.clone()] + [p.grad.clone() for p in m.parameters()]

    for a, b in zip(grads(False), grads(True)):
        assert torch.equal(a, b)
    m = _load(UncertaintyDiscriminator(4), params, dev)
    with pytest.raises(RuntimeError):
        m.replay()


def test_discriminator_joint_replay_matches_the_one_batch_form(dev):
    """forward_cached(target, room=1) + forward_fill(source) + replay(): the update's backward pass over source + target
    as ONE batch without a second forward on the target -- against the network run on cat([source, target])."""
    from oracle import nets as ON
    from pointcloududa_amd.networks import UncertaintyDiscriminator
    from pointcloududa_amd.utils import loss as L
    params = ON.make_params(ON.disc_param_shapes(4, False), 81, std=0.02)
    rng = np.random.default_rng(82)
    xs_np, xt_np = (rng.normal(0, 1, (3, 4, 96, 96)).astype(np.float32) for _ in range(2))

    def run(joint):
        m = _load(UncertaintyDiscriminator(4), params, dev)
        xs = torch.from_numpy(xs_np).to(dev)
        xt = torch.from_numpy(xt_np).to(dev).requires_grad_(True)
        m.requires_grad_(False)
        d_t = m.forward_cached(xt, room=1) if joint else m(xt)
        L.bce_logits_const(d_t, 1.0, weight=0.01).backward()
        m.requires_grad_(True)
        if joint:
            m.forward_fill(xs)
            d = m.replay()
        else:
            d = m(torch.cat([xs, xt.detach()], 0))
        assert torch.equal(d[3:], d_t)
        ls = [L.bce_logits_const(d[:3], 1.0), L.bce_logits_const(d[3:], 0.0)]
        torch.autograd.backward(ls, [torch.ones((), device=dev)] * 2)
        if joint:
            m.drop_cache()
        return [d.detach().clone(), xt.grad.clone()] + [p.grad.clone() for p in m.parameters()]

    one, joint = run(False), run(True)
    # outputs, the frozen pass's input gradient and every layer's weight gradient are the same bits; the FIRST layer's
    # weight gradient is accumulated batch by batch (its inputs stay two tensors): same sum, other rounding order
    for i, (a, b) in enumerate(zip(one, joint)):
        if i == 2:
            assert rel_err(b, a) < 1e-6
        else:
            assert torch.equal(a, b), i
