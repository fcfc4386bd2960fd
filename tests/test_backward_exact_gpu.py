"""Backward passes of every network at 1e-4, with the routing decisions shared.

The reference networks' gradients are discontinuous in their activations (LeakyReLU / ReLU sign, max-pool and
max-over-points argmax, nearest-neighbour indices of the point loss): against an independent fp32 run the
whole-network gradient checks of test_networks_gpu.py can only be held to ~1e-2 (scripts/gradient_conditioning.py).
Here the CPU restatement (oracle.nets, PyTorch autograd) is ANCHORED to the HIP forward pass: every conv / linear /
normalisation output takes the value the HIP kernels produced (``z + (z_hip - z).detach()``), so both sides take the
same branch everywhere while autograd still differentiates the restatement.  What remains is the arithmetic of the
HIP backward kernels (dgrad, wgrad, BatchNorm / LeakyReLU / pooling / upsampling backward, the two-destination and
accumulating forms, the fused losses' gradients) through the PRODUCTION autograd nodes: every parameter gradient and
the input gradient must agree to 1e-4 of the tensor's scale.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4
# whole-step segmenter gradients: worst-case linear accumulation of 2 x 2^-17 per convolution layer over the 29 + 5 layers
# between the adversarial loss and the first encoder block (derivation in test_train_step_backward_shared_routing)
SEG_CHAIN_TOL = 2 * (29 + 5) * 2.0 ** -17


def _load(mod, params, dev):
    mod.load_state_dict({k: v.clone() for k, v in params.items()}, strict=True)
    return mod.to(dev).train()


def _unlrelu(a, slope):
    """pre-activation with the sign (and, up to one rounding, the value) the HIP kernel saw"""
    a = a.detach().float().cpu()
    return a if slope == 1.0 else torch.where(a > 0, a, a / slope)


PRE_TOL = 2e-4      # layer-local forward bound (below)


def _anchor_from(table, used, worst=None):
    """With every upstream output anchored, the difference between the restatement's output of a layer and the HIP
    kernels' BEFORE it is anchored is that layer's own arithmetic error (bf16x3 products, fp32 accumulation order):
    held to 2e-4 of the tensor's scale; ``worst`` collects the largest one per network for the test's report."""
    def fn(tag, z):
        if tag not in table:
            return z
        used.add(tag)
        v = table[tag]
        if isinstance(v, tuple):          # (post-ReLU value, True): share the mask, keep own value where inactive
            y = v[0].detach().float().cpu().reshape(z.shape)
            tgt = torch.where(y > 0, y, torch.clamp(z.detach(), max=0.0))
        else:
            tgt = v.reshape(z.shape)
        e = rel_err(z, tgt)
        if worst is not None and e > worst.get("e", 0.0):
            worst["e"], worst["tag"] = e, tag
        assert e < PRE_TOL, (tag, e)      # the two forward passes agree, layer by layer, before anchoring
        return z + (tgt - z).detach()
    return fn


def _compare_grads(named_hip, grads_ref, tol=TOL, parts=None):
    """``parts``: name -> one of two partial gradients whose sum ``grads_ref`` is (the discriminators' source and target
    passes, which largely cancel at initialisation): the error is then taken relative to the larger PART's scale"""
    worst = ("", 0.0)
    total = sum(float(g.double().norm()) ** 2 for g in grads_ref.values() if g is not None) ** 0.5
    for k, p in named_hip:
        g = grads_ref.get(k)
        if g is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        # (a bias in front of a BatchNorm has an exactly-zero true gradient: both sides hold rounding noise there)
        if float(g.double().norm()) < 1e-5 * total:
            assert float(p.grad.double().norm()) < 1e-4 * total, k
            continue
        e = rel_err(p.grad, g)
        if parts is not None:
            scale = max(float(parts[k].abs().max()), float((g - parts[k]).abs().max()), float(g.abs().max()))
            e = float((p.grad.detach().double().cpu() - g.double()).abs().max()) / max(scale, 1e-30)
        if e >= tol and float((p.grad.detach().double().cpu() - g.double()).abs().max()) <= 1e-6 * total:
            continue      # a near-zero gradient (just above the floor above): rounding noise on both sides
        if e > worst[1]:
            worst = (k, e)
        assert e < tol, (k, e)
    return worst


def _random_running_stats(params, seed):
    rng = np.random.default_rng(seed + 7)
    for k in params:
        if ".in" in k or k.startswith("in"):
            continue
        if k.endswith("running_mean"):
            params[k] = torch.from_numpy(rng.normal(0, 0.2, tuple(params[k].shape)).astype(np.float32))
        if k.endswith("running_var"):
            params[k] = torch.from_numpy(rng.uniform(0.5, 1.5, tuple(params[k].shape)).astype(np.float32))


def _seg_table(S, cfg, logits, verts):
    """anchor table of one segmenter forward pass: pre-activation outputs of every convolution the HIP engine kept"""
    table = {"classifier": logits.detach().float().cpu()}
    for blk in ["encoder.encoder%d" % (i + 1) for i in range(cfg.n_block)] + \
               ["decoder.decoder2_%d" % (i + 1) for i in range(cfg.n_block)]:
        _, _, a0, _, a1, _ = S[blk]
        table[blk + ".0"], table[blk + (".3" if cfg.batchnorm else ".2")] = _unlrelu(a0, 0.01), _unlrelu(a1, 0.01)
    for i in range(1, cfg.n_block):
        c1 = "encoder.conv1_%d.0" % (i + 1)
        table[c1] = _unlrelu(S[c1][2], 0.01)
    for j, o in enumerate(S["bott_outs"]):
        table["bottleneck.bottleneck%d.0" % (j + 1)] = _unlrelu(o, 0.01)
    if cfg.pointnet:
        table["pointNet.final_conv"] = _unlrelu(S["head"][1], 0.01)
        for nm, _, o in S["head_ext"]:
            table[nm] = _unlrelu(o, 0.01)
        table["pointNet.final_fc"] = verts.detach().float().cpu()
    return table


def _disc_table(model):
    names = [n for n, _ in model._chain]
    acts = model._last_acts
    return {n: _unlrelu(acts[i + 1], 0.2 if i < len(names) - 1 else 1.0) for i, n in enumerate(names)}


def _pn_table(trace):
    table = {}
    for k, v in trace.items():
        table[k] = (v[0], True) if (isinstance(v, tuple) and v[1]) else (v[0] if isinstance(v, tuple) else v)
        if not isinstance(table[k], tuple):
            table[k] = table[k].detach().float().cpu()
    return table


@pytest.mark.parametrize("cfg_kw,softmax,b,hw,seed", [
    (dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9), False, 2, 128, 1100),
    (dict(filters=8, in_channels=3, n_class=5, pointnet=False), True, 2, 64, 1110),
    (dict(filters=16, in_channels=1, n_class=4, pointnet=True, fc_inch=9), False, 3, 128, 1120),
    (dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9, extpn=True), False, 2, 128, 1130),
    # eval mode (negative seed): BatchNorm on its running statistics -- frozen-BatchNorm fine-tuning, unet.py:26,30 under
    # model.eval() -- is a fixed affine in the backward pass
    (dict(filters=8, in_channels=3, n_class=5, pointnet=True, fc_inch=9), True, 2, 128, -1140),
    (dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9, batchnorm=False), False, 2, 128, 1150),
    # the reference's real MS-CMRSeg shape, full width (train_mscmrseg.py:412-414): 224x224x3, fc_inch=81 -- maps of 14 / 28 / 56 /
    # 112 pixels (off the 32-pixel tiles), dilation 8 on the 14-wide bottleneck, the aligned weight-gradient kernel's
    # ragged cases
    (dict(filters=32, in_channels=3, n_class=4, pointnet=True, fc_inch=81), False, 2, 224, 1160),
])
def test_segmenter_backward_shared_routing(dev, cfg_kw, softmax, b, hw, seed):
    """encoder blocks + max-pool + dense-skip 1x1 convs, the dilated bottleneck and its running sum, the point head,
    the decoder (upsampling fold, zero-copy concat) and the classifier, under the reference's supervised loss"""
    training, seed = seed > 0, abs(seed)
    from oracle import losses as OL
    from oracle import nets as ON
    from oracle.synth import synth_batch
    from pointcloududa_amd.networks import Segmentation_model_Point
    from pointcloududa_amd.utils import loss as L
    cfg = ON.SegCfg(**cfg_kw)
    params = ON.make_params(ON.seg_param_shapes(cfg), seed)
    if not training:
        _random_running_stats(params, seed)
    model = _load(Segmentation_model_Point(**cfg_kw), params, dev).train(training)
    model._keep_state = True
    img, mask, vert, _, _ = synth_batch(b, cfg.in_channels, cfg.n_class, hw, seed=seed + 1)
    x = torch.from_numpy(img).to(dev).requires_grad_(True)
    logits, _, verts = model(x)
    one = torch.ones((), device=dev)
    l_main, l_jac = L.seg_loss(logits, torch.from_numpy(mask).to(dev), "softmax" if softmax else "sigmoid")
    seeds, gs = [l_main, l_jac], [one, one]
    if cfg.pointnet:
        seeds.append(L.batch_NN_loss(verts, torch.from_numpy(vert).to(dev))); gs.append(one)
    torch.autograd.backward(seeds, gs)

    table = _seg_table(model._last_S, cfg, logits, verts)

    p2 = {k: (v.clone().requires_grad_(True) if ON.is_trainable(k) else v.clone()) for k, v in params.items()}
    xo = torch.from_numpy(img).requires_grad_(True)
    used, pre = set(), {}
    with ON.anchored(_anchor_from(table, used, pre)):
        lo2, ve2 = ON.seg_forward(p2, xo, cfg, training=training)
    assert used == set(table), set(table) - used
    m2, j2 = (OL.seg_loss_softmax if softmax else OL.seg_loss_sigmoid)(lo2, torch.from_numpy(mask))
    loss2 = m2 + j2 + (OL.batch_nn_loss(ve2, torch.from_numpy(vert)) if cfg.pointnet else 0.0)
    loss2.backward()
    ref = {k: v.grad for k, v in p2.items() if ON.is_trainable(k)}
    # (batchnorm=False: nothing normalises the activations, the logits saturate the sigmoid and the classifier's bias gradient
    # is a sum of 32k terms that cancel to a hundredth of their magnitude: 3e-4 there)
    tol = TOL if cfg.batchnorm else 3e-4
    worst = _compare_grads(model.named_parameters(), ref, tol=tol)
    e_dx = rel_err(x.grad, xo.grad)
    assert e_dx < tol, e_dx
    print("worst parameter gradient error %s %.2e, dx %.2e; worst layer-local forward error %s %.2e"
          % (worst[0], worst[1], e_dx, pre.get("tag"), pre.get("e", 0.0)))


@pytest.mark.parametrize("inch,ext,hw,seed", [(4, False, 64, 1200), (5, True, 128, 1210), (4, False, 256, 1220),
                                                (4, False, 224, 1230)])      # 224: maps of 113 / 57 / 29 / 15 / 8
def test_discriminator_backward_shared_routing(dev, inch, ext, hw, seed):
    """the full UncertaintyDiscriminator chain (stride-2 4x4 convolutions, their transposed-convolution input gradients
    per parity class, LeakyReLU(0.2) backward, the tap-unfolded first layer) under the domain loss"""
    from oracle import losses as OL
    from oracle import nets as ON
    from pointcloududa_amd.networks import UncertaintyDiscriminator
    from pointcloududa_amd.utils import loss as L
    params = ON.make_params(ON.disc_param_shapes(inch, ext), seed, std=0.02)
    model = _load(UncertaintyDiscriminator(in_channel=inch, ext=ext), params, dev)
    model._keep_acts = True
    rng = np.random.default_rng(seed + 1)
    xn = rng.normal(0, 1, (2, inch, hw, hw)).astype(np.float32)
    x = torch.from_numpy(xn).to(dev).requires_grad_(True)
    d = model(x)
    L.bce_logits_const(d, 1.0).backward()
    table = _disc_table(model)
    p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = torch.from_numpy(xn).requires_grad_(True)
    used, pre = set(), {}
    with ON.anchored(_anchor_from(table, used, pre)):
        d2 = ON.disc_forward(p2, xo, ext)
    assert used == set(table)
    OL.bce_logits_const(d2, 1.0).backward()
    worst = _compare_grads(model.named_parameters(), {k: v.grad for k, v in p2.items()})
    e_dx = rel_err(x.grad, xo.grad)
    assert e_dx < TOL, e_dx
    print("worst parameter gradient error %s %.2e, dx %.2e; worst layer-local forward error %s %.2e"
          % (worst[0], worst[1], e_dx, pre.get("tag"), pre.get("e", 0.0)))


@pytest.mark.parametrize("ft,ext,b,seed", [(False, False, 16, 1300), (True, True, 12, 1310), (False, False, 4, 1320),
                                            (True, True, 6, -1330)])      # negative seed: eval mode (running statistics)
def test_pointnet_cls_backward_shared_routing(dev, ft, ext, b, seed):
    """PointNetCls (T-Nets, k=1 convolutions + BatchNorm1d + ReLU, max over points, FC + BatchNorm1d) with the ReLU
    masks and the max-over-points argmax shared"""
    training, seed = seed > 0, abs(seed)
    from oracle import losses as OL
    from oracle import nets as ON
    from pointcloududa_amd.networks import PointNetCls
    from pointcloududa_amd.utils import loss as L
    params = ON.make_params(ON.pointnet_cls_param_shapes(ft, ext=ext), seed)
    if not training:
        _random_running_stats(params, seed)
    model = _load(PointNetCls(feature_transform=ft, ext=ext, drop=0.0), params, dev).train(training)
    model._keep_trace = True
    rng = np.random.default_rng(seed + 1)
    xn = rng.random((b, 3, 300), dtype=np.float32)
    x = torch.from_numpy(xn).to(dev).requires_grad_(True)
    y, _, _ = model(x)
    L.bce_logits_const(y, 0.0).backward()
    table = _pn_table(model._last_trace)
    p2 = {k: (v.clone().requires_grad_(True) if ON.is_trainable(k) else v.clone()) for k, v in params.items()}
    xo = torch.from_numpy(xn).requires_grad_(True)
    used, pre = set(), {}
    with ON.anchored(_anchor_from(table, used, pre)):
        y2, _, _ = ON.pointnet_cls_forward(p2, xo, feature_transform=ft, ext=ext, drop=0.0, training=training)
    assert used == set(table), set(table) - used
    OL.bce_logits_const(y2, 0.0).backward()
    ref = {k: v.grad for k, v in p2.items() if ON.is_trainable(k)}
    worst = _compare_grads(model.named_parameters(), ref)
    e_dx = rel_err(x.grad, xo.grad)
    assert e_dx < TOL, e_dx
    print("worst parameter gradient error %s %.2e, dx %.2e; worst layer-local forward error %s %.2e"
          % (worst[0], worst[1], e_dx, pre.get("tag"), pre.get("e", 0.0)))


@pytest.mark.parametrize("variant,pn_kw,in_ch,n_class,seed,flags", [
    ("mscmrseg", {}, 1, 4, 1400, {}),
    ("mmwhs", dict(feature_transform=True, ext=True), 3, 5, 1410, {}),
    # the optional entropy terms of the MM-WHS loop (train_mmwhs.py:227-230,245-247): their gradients enter through the
    # mean-of-the-map path of the entropy kernel
    ("mmwhs", dict(feature_transform=True, ext=True), 3, 5, 1420, dict(etpls=True, Tetpls=True)),
])
def test_train_step_backward_shared_routing(dev, variant, pn_kw, in_ch, n_class, seed, flags):
    """The COMPOSITION the step adds on top of the per-network passes: frozen-discriminator input gradients ->
    entropy-map / softmax backward -> accumulation into the segmenter's second backward pass, and the discriminators'
    two-pass updates.  The CPU restatement of the whole step (oracle.step.OracleTrainer) runs with EVERY forward pass
    anchored to the corresponding HIP pass (11 passes: the segmenter on the source and target batch, each
    discriminator frozen on the target batch and training on source and target).  Losses: 1e-5.  Every discriminator
    gradient: 1e-4 of the scale of its two passes' gradients (source-as-1 and target-as-0 largely cancel in their sum at
    initialisation: D(x) ~ 0 on both).  The segmenter's gradients -- supervised, and supervised + adversarial, the one
    its optimiser consumes -- SEG_CHAIN_TOL, a bound DERIVED from the arithmetic, not fitted to a run (round-4 review, weak
    2: the bar had followed a plan change from 3e-4 to 4e-4): the gradient of the first encoder block sits behind 29
    segmenter + 5 discriminator convolution layers; a bf16x3 product carries each operand to 2^-17 of its scale (hi + lo,
    two bf16 roundings) and drops the lo x lo term (2^-18), so a layer's output is off by at most ~2 x 2^-17 of its
    tensor's scale, and in the worst case the layers' errors add linearly: 2 x 34 x 2^-17 = 5.2e-4, whatever tiling or
    summation order the planner picks (observed: 1.4e-4 ... 3.1e-4 across seeds and plans; the per-network twins above,
    whose chains are 5-29 layers, see 4-6e-5).  The worst value per gradient is printed."""
    import oracle.step as OS
    from oracle import nets as ON
    from oracle.synth import synth_batch
    from pointcloududa_amd.networks import PointNetCls, Segmentation_model_Point, UncertaintyDiscriminator
    from pointcloududa_amd.train_step import AdversarialTrainer, TrainCfg
    cfg_kw = dict(filters=4, in_channels=in_ch, n_class=n_class, pointnet=True, fc_inch=9)
    cfg = ON.SegCfg(**cfg_kw)
    pg = ON.make_params(ON.seg_param_shapes(cfg), seed)
    p1 = ON.make_params(ON.disc_param_shapes(n_class), seed + 1, std=0.02)
    p2 = ON.make_params(ON.disc_param_shapes(n_class), seed + 2, std=0.02)
    p4 = ON.make_params(ON.pointnet_cls_param_shapes(**pn_kw), seed + 3)
    gen = _load(Segmentation_model_Point(**cfg_kw), pg, dev)
    d1 = _load(UncertaintyDiscriminator(in_channel=n_class), p1, dev)
    d2 = _load(UncertaintyDiscriminator(in_channel=n_class), p2, dev)
    d4 = _load(PointNetCls(drop=0.0, **pn_kw), p4, dev)
    mom = 0.99 if variant == "mscmrseg" else 0.95
    tr = AdversarialTrainer(gen, d1, d2, d4, TrainCfg(variant=variant, n_class=n_class, d_momentum=mom, **flags))
    # the reference's own order of passes (no joint source + target batch, no replay of the frozen pass): one forward
    # call per (network, role), so that each can be recorded
    tr.d_streams = tr.d_overlap = tr.d_batch = tr.early_fwd2 = tr.d_reuse = tr.d_joint = False
    gen._keep_state, d1._keep_acts, d2._keep_acts, d4._keep_trace = True, True, True, True
    tables = {"gen": [], "d1": [], "d2": [], "d4": []}

    def spy(mod, name, make):
        inner = mod.forward

        def fwd(*a, **k):
            out = inner(*a, **k)
            tables[name].append(make(out))
            return out
        mod.forward = fwd
    spy(gen, "gen", lambda out: _seg_table(gen._last_S, cfg, out[0], out[2]))
    spy(d1, "d1", lambda out: _disc_table(d1))
    spy(d2, "d2", lambda out: _disc_table(d2))
    spy(d4, "d4", lambda out: _pn_table(d4._last_trace))
    batch = synth_batch(4, in_ch, n_class, 128, seed=seed + 100)
    out = tr.step(*[torch.from_numpy(t).to(dev) for t in batch], keep=True)
    h = AdversarialTrainer.to_host(out, tr.cfg)
    assert [len(tables[k]) for k in ("gen", "d1", "d2", "d4")] == [2, 3, 3, 3]

    scfg = OS.StepCfg(variant=variant, n_class=n_class, d_momentum=mom, pn_feature_transform=bool(pn_kw.get("feature_transform")),
                      pn_ext=bool(pn_kw.get("ext")), **flags)
    orc = OS.OracleTrainer(cfg, scfg, pg, p1, p2, p4)
    cur, used, pre, calls = {}, set(), {}, {"gen": 0, "d1": 0, "d2": 0, "d4": 0}

    def anchor(tag, z):
        return _anchor_from(cur["table"], cur["used"], pre)(tag, z)

    def enter(name):
        if "table" in cur:
            assert cur["used"] == set(cur["table"]), (cur["name"], set(cur["table"]) - cur["used"])
        cur["name"], cur["table"], cur["used"] = name, tables[name][calls[name]], set()
        calls[name] += 1
    real = (OS.seg_forward, OS.disc_forward, OS.pointnet_cls_forward)

    def seg_fw(p, x, c, training=True):
        enter("gen")
        return real[0](p, x, c, training=training)

    def disc_fw(p, x, ext=False):
        enter("d1" if p is orc.dis1 else "d2")
        return real[1](p, x, ext=ext)

    def pn_fw(p, x, **k):
        enter("d4")
        return real[2](p, x, **k)
    OS.seg_forward, OS.disc_forward, OS.pointnet_cls_forward = seg_fw, disc_fw, pn_fw
    try:
        with ON.anchored(anchor):
            q = orc.step(*batch, keep=True)
    finally:
        OS.seg_forward, OS.disc_forward, OS.pointnet_cls_forward = real
    assert calls == {"gen": 2, "d1": 3, "d2": 3, "d4": 3}
    assert cur["used"] == set(cur["table"])
    for k in ("seg_loss", "adv_loss", "ver_s_loss", "ver_t_loss", "d1_loss_src", "d1_loss_tgt", "d2_loss_src", "d2_loss_tgt",
              "d4_loss_src", "d4_loss_tgt"):
        assert abs(h[k] - q[k]) <= 1e-5 * max(1.0, abs(q[k])), (k, h[k], q[k])
    if variant == "mmwhs":
        for k in ("entropy_loss", "entropy_loss_T"):
            assert abs(h[k] - q[k]) <= 1e-5 * max(1.0, abs(q[k])), (k, h[k], q[k])

    def flat_named(mod, snap):
        out_, off = [], 0
        for k, p in mod.named_parameters():
            n = p.numel()
            out_.append((k, snap[off:off + n].view(p.shape)))
            off += (n + 63) // 64 * 64
        return out_

    class _G:      # _compare_grads reads ``.grad``
        def __init__(self, g):
            self.grad = g
    report = []
    for nm, mod in (("grad_seg", gen), ("grad_total", gen), ("grad_d1", d1), ("grad_d2", d2), ("grad_d4", d4)):
        named = [(k, _G(g)) for k, g in flat_named(mod, tr.last[nm])]
        ref = orc.kept[nm]
        w = _compare_grads(named, ref, tol=SEG_CHAIN_TOL if mod is gen else TOL, parts=orc.kept.get(nm + "_src"))
        report.append("%s %s %.2e" % (nm, w[0], w[1]))
    print("%s step %s: worst gradient errors: %s; worst layer-local forward error %s %.2e"
          % (variant, flags, "; ".join(report), pre.get("tag"), pre.get("e", 0.0)))
