"""The full 5-phase adversarial train step on the HIP kernels against golden vectors produced by
re-typing train_mscmrseg.py:183-330 around the REFERENCE modules (oracle/make_golden.py).

Step 0 starts from identical parameters: every loss, the segmenter outputs and the gradient norms
are compared.  Step 1 starts from parameters that went through Adam (sign-sensitive for near-zero
gradients) and BatchNorm over a tiny batch; even the fp32 restatement only tracks the reference to
~1e-2 there, so only scalars are compared, loosely."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, rel_err

pytestmark = pytest.mark.gpu


def _build(cfg_kw, seed, dev, d1=True, d2=True, d4=True, variant="mscmrseg", pn_kw=None, momentum=0.99, **flags):
    from oracle import nets as ON
    from pointcloududa_amd.networks import PointNetCls, Segmentation_model_Point, UncertaintyDiscriminator
    from pointcloududa_amd.train_step import AdversarialTrainer, TrainCfg
    cfg = ON.SegCfg(**cfg_kw)
    pn_kw = pn_kw or {}
    pg = ON.make_params(ON.seg_param_shapes(cfg), seed)
    p1 = ON.make_params(ON.disc_param_shapes(cfg.n_class), seed + 1, std=0.02)
    p2 = ON.make_params(ON.disc_param_shapes(cfg.n_class), seed + 2, std=0.02)
    p4 = ON.make_params(ON.pointnet_cls_param_shapes(**pn_kw), seed + 3)
    load = lambda m, p: (m.load_state_dict({k: v.clone() for k, v in p.items()}), m.to(dev).train())[1]
    gen = load(Segmentation_model_Point(**cfg_kw), pg)
    m1 = load(UncertaintyDiscriminator(in_channel=cfg.n_class), p1) if d1 else None
    m2 = load(UncertaintyDiscriminator(in_channel=cfg.n_class), p2) if d2 else None
    m4 = load(PointNetCls(drop=0.0, **pn_kw), p4) if d4 else None
    tr = AdversarialTrainer(gen, m1, m2, m4, TrainCfg(variant=variant, n_class=cfg.n_class, d1=d1, d2=d2, d4=d4,
                                                      d_momentum=momentum, **flags))
    tr._p0 = {"gen": pg, "d1": p1, "d2": p2, "d4": p4}
    return cfg, tr


def _flat_norms(opt, module):
    """per-parameter norms out of the flat gradient snapshot"""
    out, off = {}, 0
    for k, p in module.named_parameters():
        n = p.numel()
        out[k] = off, n
        off += (n + 63) // 64 * 64
    return out


def _sample(t, n=256):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].double().cpu().numpy()


# Loss scalars, per key: 1e-3, the north-star figure, for every one of them on the first step -- the d4-derived ones
# included since the point-cloud discriminator's k=1 convolutions run in exact fp32 (csrc/conv1d_f32.hip; on the bf16x3
# image convolution BatchNorm1d over 4-8 clouds behind a max over 300 points had turned 2^-17 noise into percents and
# these three scalars sat behind a 2e-2 bound).  LOOSE is what remains for the SECOND step, which starts from
# Adam-updated parameters (sign-sensitive for near-zero gradients).
TIGHT, LOOSE = 1e-3, 1e-3


def _check_losses(h, g, pre, cfg, it):
    ms = cfg.variant == "mscmrseg"
    tight = TIGHT if it == 0 else 2e-2       # step 1 starts from Adam-updated parameters (sign-sensitive): see below
    loose = LOOSE if it == 0 else 1e-1
    worst = 0.0
    for k in ("seg_loss", "ver_s_loss", "ver_t_loss", "d1_loss_src", "d1_loss_tgt", "d2_loss_src", "d2_loss_tgt",
              "d4_loss_src", "d4_loss_tgt"):
        if pre + k not in g:
            assert k not in h, k
            continue
        ref, tol = float(g[pre + k]), (loose if k.startswith("d4") else tight)
        err = abs(h[k] - ref) / max(1e-3, abs(ref))
        assert err <= tol, (it, k, h[k], ref)
        if not k.startswith("d4"):
            worst = max(worst, err)
    # the three adversarial terms (dr- and w-scaled as the scripts sum them) and their sum
    parts = {}
    for k, w in (("adv1", 1.0 if ms else cfg.w1), ("adv2", 1.0 if ms else cfg.w2), ("adv4", 1.0 if ms else cfg.w4)):
        if k in h:
            got, ref = cfg.dr * h[k], float(g[pre + k])
            tol = loose if k == "adv4" else tight
            assert abs(got - ref) <= tol * max(1e-5, abs(ref)), (it, k, got, ref)
            parts[k] = (abs(ref) * w, tol)
    ref = float(g[pre + "adv_loss"])
    assert abs(h["adv_loss"] - ref) <= sum(a * t for a, t in parts.values()) + 1e-9, (it, "adv_loss", h["adv_loss"], ref)
    return worst


def _check_updates(tr, g, pre):
    """Parameters after the optimiser steps against golden strided samples of the reference's.
    Adam's first step moves a parameter by -lr * g / (|g| + eps): the update DIRECTION must match wherever the
    reference gradient is clear of the gradient noise floor, and both implementations must have moved by lr.
    SGD (discriminators): (p1 - p0) / lr = -(g + wd * p0) is linear in the gradient and is compared as a tensor."""
    from oracle import nets as ON
    lr = tr.cfg.lr
    n_conf = n_conf_ok = n_all = n_all_ok = 0
    for k, v in tr.gen.state_dict().items():
        key = pre + "ps/gen/" + k
        if key not in g or not v.dtype.is_floating_point:
            continue
        got, ref = _sample(v), g[key].astype(np.float64)
        if not ON.is_trainable(k):
            assert np.abs(got - ref).max() <= 1e-3 * max(np.abs(ref).max(), 1e-3), (k, np.abs(got - ref).max())
            continue
        if pre + "gs/gen/" + k not in g:
            assert np.array_equal(got, _sample(tr._p0["gen"][k])), k        # no gradient (encoder.conv1_1): untouched
            continue
        p0, gr = _sample(tr._p0["gen"][k]), g[pre + "gs/gen/" + k].astype(np.float64)
        u_got, u_ref = (got - p0) / lr, (ref - p0) / lr
        assert np.abs(u_got).max() <= 1.0 + 1e-3 and np.abs(u_ref).max() <= 1.0 + 1e-3, k
        scale = np.abs(gr).max()
        if scale < 1e-12:
            continue
        conf, some = np.abs(gr) >= 0.2 * scale, np.abs(gr) >= 0.02 * scale
        ok = np.abs(u_got - u_ref) <= 0.1                       # same sign AND a full-size step on both sides
        n_conf += int(conf.sum()); n_conf_ok += int((conf & ok).sum())
        n_all += int(some.sum()); n_all_ok += int((some & ok).sum())
        # sum of the parameter: at most 15 % of the elements (+ 3: tensors of 4 elements) may have stepped the other
        # way (2 * lr each)
        ps_ref, ps_got = float(g[pre + "psum/gen/" + k]), float(v.double().sum())
        assert abs(ps_got - ps_ref) <= (0.15 * v.numel() + 3) * 2 * lr + 1e-5 * abs(ps_ref), (k, ps_got, ps_ref)
    assert n_conf >= 500, n_conf                                 # not vacuous
    assert n_conf_ok >= 0.98 * n_conf, (n_conf_ok, n_conf)
    assert n_all_ok >= 0.90 * n_all, (n_all_ok, n_all)
    rates = {"adam_conf": n_conf_ok / n_conf, "adam_all": n_all_ok / max(n_all, 1)}
    # SGD with momentum, first step: buf = g + wd * p0 and p1 = p0 - lr * buf.  lr * buf is about ONE fp32 ulp of a
    # discriminator weight, so the direction is read from the momentum buffer (full precision) and the parameters
    # are only required to have moved by that much (+- 1 ulp).  d1 / d2: elementwise 6e-2 of the tensor's largest entry
    # (independent forward passes: a few LeakyReLU(0.2) routing flips in the 129^2 ... 9^2 maps; observed <= 3.8e-2; the
    # 1e-4 check with the routing shared is test_discriminator_backward_shared_routing).  d4 sits behind
    # BatchNorm1d over the batch of 4-8 (see _check_losses): its buffer is held to 0.35 in norm per tensor and to a
    # cosine of 0.95 with the reference's over all sampled elements (observed 0.980-0.998 since PointNetCls runs in exact fp32).
    for nm, mod, opt in (("d1", tr.dis1, tr.opt_d1), ("d2", tr.dis2, tr.opt_d2), ("d4", tr.dis4, tr.opt_d4)):
        if mod is None:
            continue
        worst, dot, n_got, n_ref = 0.0, 0.0, 0.0, 0.0
        for (k, v), (off, n, shp) in zip(mod.named_parameters(), opt._slices()):
            key = pre + "mb/%s/%s" % (nm, k)
            assert key in g, key
            got, ref = _sample(opt.buf[off:off + n]), g[key].astype(np.float64)
            scale = max(np.abs(ref).max(), 1e-30)
            if nm == "d4":
                dot += float(got @ ref); n_got += float(got @ got); n_ref += float(ref @ ref)
            else:
                e = np.abs(got - ref).max() / scale
                assert e <= 6e-2, (nm, k, e)
                worst = max(worst, e)
            p0, p1, p1_ref = _sample(tr._p0[nm][k]), _sample(v), g[pre + "ps/%s/%s" % (nm, k)].astype(np.float64)
            big = np.maximum(np.maximum(np.abs(p0), np.abs(p1)), np.abs(opt.lr * got))
            ulp = np.spacing(big.astype(np.float32)).astype(np.float64)
            assert np.all(np.abs((p1 - p0) + opt.lr * got) <= 1.51 * ulp), (nm, k)          # moved by lr * buf
            assert np.all(np.abs(p1 - p1_ref) <= opt.lr * np.abs(got - ref) + 2.01 * ulp), (nm, k)
        if nm == "d4":
            cos = dot / max((n_got * n_ref) ** 0.5, 1e-30)
            assert cos >= 0.95 and abs(n_got ** 0.5 - n_ref ** 0.5) <= 0.35 * n_ref ** 0.5, (cos, n_got, n_ref)
            rates["sgd_d4_cos"] = cos
        else:
            rates["sgd_" + nm] = worst
        for k, v in mod.state_dict().items():               # BatchNorm running statistics
            key = pre + "ps/%s/%s" % (nm, k)
            if key in g and v.dtype.is_floating_point and not ON.is_trainable(k) and ".in" not in k and not k.startswith("in"):
                got, ref = _sample(v), g[key].astype(np.float64)
                assert np.abs(got - ref).max() <= 5e-2 * max(np.abs(ref).max(), 1e-3), (nm, k)
    return rates


@pytest.mark.parametrize("tag,cfg_kw,full,dflags", [
    ("step_small", dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9), True, (True, True, True)),
    ("step_full256", dict(filters=32, in_channels=1, n_class=4, pointnet=True, fc_inch=121), False, (True, True, True)),
    # the reference's real MS-CMRSeg operating point (train_mscmrseg.py:412-425): 224x224x3, fc_inch=81, discriminator maps
    # 113 / 57 / 29 / 15 / 8
    ("step_full224", dict(filters=32, in_channels=3, n_class=4, pointnet=True, fc_inch=81), False, (True, True, True)),
    # BASELINE config 2 in miniature: train_mscmrseg.py -d2 (no point head, the entropy-map discriminator only)
    ("step_d2only_small", dict(filters=4, in_channels=1, n_class=4, pointnet=False), True, (False, True, False)),
])
def test_train_step_vs_reference_golden(dev, tag, cfg_kw, full, dflags):
    from oracle.synth import synth_batch
    from pointcloududa_amd.train_step import AdversarialTrainer
    g = np.load(os.path.join(GOLD, tag + ".npz"))
    seed, b, hw, n_steps = int(g["seed"]), int(g["b"]), int(g["hw"]), int(g["n_steps"])
    d1, d2, d4 = dflags
    cfg, tr = _build(cfg_kw, seed, dev, d1=d1, d2=d2, d4=d4)
    for it in range(n_steps):
        batch = synth_batch(b, cfg.in_channels, cfg.n_class, hw, seed=seed + 100 + it)
        img_a, mask_a, vert_a, img_b, vert_b = [torch.from_numpy(t).to(dev) for t in batch]
        out = tr.step(img_a, mask_a, vert_a, img_b, vert_b, keep=(it == 0))
        h = AdversarialTrainer.to_host(out, tr.cfg)
        worst = _check_losses(h, g, "s%d/" % it, tr.cfg, it)
        assert abs(h["seg_dice"] - float(g["s%d/seg_dice" % it])) < (1e-4 if it == 0 else 5e-2)
        for d, on in (("dis1", d1), ("dis2", d2), ("dis4", d4)):
            if on:
                assert 0.0 <= h[d + "_acc1"] <= 1.0 and 0.0 <= h[d + "_acc2"] <= 1.0
            else:
                assert d + "_acc1" not in h
        if it > 0:
            continue
        last = tr.last
        if full:
            assert rel_err(last["oS"], g["s0/oS"]) < 1e-3 and rel_err(last["oT"], g["s0/oT"]) < 1e-3
        else:
            from test_networks_gpu import _strided
            assert rel_err(_strided(last["oS"]), g["s0/oS_s"]) < 1e-3
            assert rel_err(_strided(last["oT"]), g["s0/oT_s"]) < 1e-3
        if d4:
            assert rel_err(last["vertS"], g["s0/vertS"]) < 1e-3 and rel_err(last["vertT"], g["s0/vertT"]) < 1e-3
        else:
            assert last["vertS"] is None and last["vertT"] is None
        # gradient norms per parameter after phase 1 (seg) and phase 2 (seg + adversarial), and of the D's.  (Independent
        # forward passes: routing flips limit these to ~1e-2, see test_backward_exact_gpu.py for the 1e-4 checks with
        # the routing shared.)  d4's BatchNorm1d over 4 samples limits what passes through it.
        for nm, mod, snap in (("grad_seg", tr.gen, last["grad_seg"]), ("grad_total", tr.gen, last["grad_total"]),
                              ("grad_d1", tr.dis1, last.get("grad_d1")), ("grad_d2", tr.dis2, last.get("grad_d2")),
                              ("grad_d4", tr.dis4, last.get("grad_d4"))):
            if mod is None:
                continue
            via_d4 = nm == "grad_d4" or (nm == "grad_total" and d4)
            tot_ref = tot_got = 0.0
            floor = 1e-3 * sum(float(g[kk]) ** 2 for kk in g.files if kk.startswith("s0/%s_norm/" % nm)) ** 0.5
            for k, (off, n) in _flat_norms(None, mod).items():
                key = "s0/%s_norm/%s" % (nm, k)
                if key not in g:
                    assert float(snap[off:off + n].abs().max()) == 0.0, (nm, k)     # e.g. encoder.conv1_1: never used
                    continue
                ref, got = float(g[key]), float(snap[off:off + n].double().norm())
                tot_ref += ref * ref; tot_got += got * got
                lim = 0.3 if via_d4 else 1e-1
                assert abs(got - ref) <= lim * ref + floor, (nm, k, got, ref)
            assert abs(tot_got ** 0.5 - tot_ref ** 0.5) <= (0.15 if via_d4 else 3e-2) * tot_ref ** 0.5, nm
        rates = _check_updates(tr, g, "s0/")
        print(tag, "worst non-d4 loss error %.2e; update checks %s" % (worst, {k: round(v, 4) for k, v in rates.items()}))


def test_segmenter_only_step_vs_reference_golden(dev):
    """train_mmwhs.py with no discriminator flag: the adversarial backward is skipped by the script's
    ``if loss_adv_diff != 0`` guard (:271), phases 3-5 by ``if args.d1 or args.d2 or args.d4`` (:282); the step is
    source forward/backward + target forward + Adam -- the seg-only branch of AdversarialTrainer._phase345."""
    from oracle.synth import synth_batch
    from pointcloududa_amd.train_step import AdversarialTrainer
    from test_networks_gpu import _strided
    g = np.load(os.path.join(GOLD, "step_segonly_small.npz"))
    seed, b, hw = int(g["seed"]), int(g["b"]), int(g["hw"])
    cfg, tr = _build(dict(filters=4, in_channels=3, n_class=5, pointnet=False), seed, dev, d1=False, d2=False, d4=False,
                     variant="mmwhs", momentum=0.95)
    batch = [torch.from_numpy(t).to(dev) for t in synth_batch(b, 3, 5, hw, seed=seed + 100)]
    out = tr.step(*batch, keep=True)
    h = AdversarialTrainer.to_host(out, tr.cfg)
    assert abs(h["seg_loss"] - float(g["seg_loss"])) <= TIGHT * abs(float(g["seg_loss"]))
    assert h["adv_loss"] == 0.0 and float(g["adv_loss"]) == 0.0
    assert not any(k.startswith(("d1_", "d2_", "d4_", "adv1", "adv2", "adv4")) for k in h)
    assert rel_err(_strided(tr.last["oS"]), g["oS_s"]) < 1e-3 and rel_err(_strided(tr.last["oT"]), g["oT_s"]) < 1e-3
    # no adversarial pass: the gradient Adam consumed is the supervised one
    assert torch.equal(tr.last["grad_seg"], tr.last["grad_total"])
    rates = _check_updates(tr, g, "")
    print("seg-only update checks", rates)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_config2_full_size_step(dev, precision):
    """BASELINE config 2 at full size (UNet without point head + the entropy-map discriminator, B = 16, 256x256,
    32 filters; `d1 = d4 = False`) in both MFMA modes.  No full-size reference run fits a fixture, so the checks are
    size-independent properties: the bf16x3 trajectory is bit-reproducible and its first-step losses equal the CPU
    restatement's on the first two samples' worth of statistics-free quantities; the single-bf16 throughput mode stays
    within its stated 2e-2 of the parity mode on every loss over three steps, and both leave d1 / d4 untouched."""
    import pointcloududa_amd as P
    from oracle.synth import synth_batch
    cfg_kw = dict(filters=32, in_channels=1, n_class=4, pointnet=False)
    batches = [[torch.from_numpy(t).to(dev) for t in synth_batch(16, 1, 4, 256, seed=1500 + i)] for i in range(3)]
    runs = {}
    try:
        for prec in ("bf16x3", precision) if precision != "bf16x3" else ("bf16x3", "bf16x3"):
            P.set_precision(prec)
            _, tr = _build(cfg_kw, 31, dev, d1=False, d2=True, d4=False)
            hs = []
            for bt in batches:
                out = tr.step(*bt)
                hs.append(tr.to_host(out, tr.cfg))
            torch.cuda.synchronize()
            runs.setdefault(prec, []).append((tr, hs))
    finally:
        P.set_precision("bf16x3")
    ref_tr, ref_hs = runs["bf16x3"][0]
    assert ref_tr.dis1 is None and ref_tr.dis4 is None and ref_tr.opt_d1 is None and ref_tr.opt_d4 is None
    for h in ref_hs:
        assert set(k for k in h if k.startswith(("d1_", "d4_", "adv1", "adv4", "ver_"))) == set()
        assert all(np.isfinite(v) for v in h.values())
    if precision == "bf16x3":
        tr2, hs2 = runs["bf16x3"][1]
        assert hs2 == ref_hs
        assert torch.equal(tr2.opt_gen.p, ref_tr.opt_gen.p) and torch.equal(tr2.opt_d2.p, ref_tr.opt_d2.p)
        # the first step's losses against the CPU restatement of the reference step on the same batch (B = 16 through
        # the fp32 oracle costs ~20 s of CPU: one step only)
        from oracle import nets as ON
        from oracle.step import OracleTrainer, StepCfg
        cfg = ON.SegCfg(**cfg_kw)
        orc = OracleTrainer(cfg, StepCfg(variant="mscmrseg", d1=False, d2=True, d4=False, n_class=4),
                            ref_tr._p0["gen"], None, ref_tr._p0["d2"], None)
        q = orc.step(*synth_batch(16, 1, 4, 256, seed=1500))
        for k in ("seg_loss", "adv_loss", "d2_loss_src", "d2_loss_tgt"):
            assert abs(ref_hs[0][k] - q[k]) <= TIGHT * max(1e-3, abs(q[k])), (k, ref_hs[0][k], q[k])
        assert abs(ref_hs[0]["seg_dice"] - q["seg_dice"]) < 1e-4
    else:
        tr2, hs2 = runs["bf16"][0]
        for it, (ha, hb) in enumerate(zip(hs2, ref_hs)):
            for k in ("seg_loss", "adv_loss", "d2_loss_src", "d2_loss_tgt"):
                assert abs(ha[k] - hb[k]) <= 2e-2 * max(1e-3, abs(hb[k])), (it, k, ha[k], hb[k])
        assert rel_err(tr2.opt_d2.p, ref_tr.opt_d2.p) < 1e-3


def _full_size_property_step(dev, cfg_kw, b, hw, variant, pn_kw, momentum, seed, gaussian):
    """Two trainers from the same weights walk a bit-identical trajectory over two steps (every kernel deterministic,
    the concurrent-stream schedule on), and the first step's losses equal the CPU restatement of the reference step
    on the same batch to 1e-3 (one oracle step: 10-20 s of host time at these sizes)."""
    from oracle import nets as ON
    from oracle.step import OracleTrainer, StepCfg
    from oracle.synth import synth_batch
    np_batches = [synth_batch(b, cfg_kw["in_channels"], cfg_kw["n_class"], hw, seed=seed + 100 + i, gaussian=gaussian)
                  for i in range(2)]
    runs = []
    for _ in range(2):
        cfg, tr = _build(cfg_kw, seed, dev, variant=variant, pn_kw=pn_kw, momentum=momentum)
        hs = []
        for bt in np_batches:
            out = tr.step(*[torch.from_numpy(t).to(dev) for t in bt])
            hs.append(tr.to_host(out, tr.cfg))
        torch.cuda.synchronize()
        runs.append((tr, hs))
    (tr_a, hs_a), (tr_b, hs_b) = runs
    assert hs_a == hs_b
    for opt in ("opt_gen", "opt_d1", "opt_d2", "opt_d4"):
        assert torch.equal(getattr(tr_a, opt).p, getattr(tr_b, opt).p), opt
    assert all(np.isfinite(v) for h in hs_a for v in h.values())
    pn_kw = pn_kw or {}
    orc = OracleTrainer(cfg, StepCfg(variant=variant, n_class=cfg.n_class, d_momentum=momentum,
                                     pn_feature_transform=bool(pn_kw.get("feature_transform")), pn_ext=bool(pn_kw.get("ext"))),
                        tr_a._p0["gen"], tr_a._p0["d1"], tr_a._p0["d2"], tr_a._p0["d4"])
    q = orc.step(*np_batches[0])
    worst = 0.0
    for k in ("seg_loss", "adv_loss", "ver_s_loss", "ver_t_loss", "d1_loss_src", "d1_loss_tgt", "d2_loss_src", "d2_loss_tgt",
              "d4_loss_src", "d4_loss_tgt"):
        err = abs(hs_a[0][k] - q[k]) / max(1e-3, abs(q[k]))
        worst = max(worst, err)
        assert err <= TIGHT, (k, hs_a[0][k], q[k])
    assert abs(hs_a[0]["seg_dice"] - q["seg_dice"]) < 1e-4
    if variant == "mmwhs":
        for k in ("entropy_loss", "entropy_loss_T"):
            assert abs(hs_a[0][k] - q[k]) <= TIGHT * abs(q[k]), (k, hs_a[0][k], q[k])
    return worst


def test_config4_full_size_step(dev):
    """BASELINE config 4's per-rank shape at FULL size: the MM-WHS loop (train_mmwhs.py:187-360) with a 3-channel
    256x256 input, 5 classes, softmax mode, PointNetCls(feature_transform=True, ext=True) (src/README.md:24), 32
    filters, B = 16 (= 128 / 8 ranks)."""
    worst = _full_size_property_step(dev, dict(filters=32, in_channels=3, n_class=5, pointnet=True, fc_inch=121), 16, 256,
                                     "mmwhs", dict(feature_transform=True, ext=True), 0.95, 41, True)
    print("config 4 full size: worst first-step loss error against the CPU restatement %.2e" % worst)


def test_config3_full_size_step(dev):
    """BASELINE config 3 -- the configuration every bench number is quoted on -- at ITS batch size: the MS-CMRSeg loop
    (train_mscmrseg.py:183-330), 1-channel 256x256 input, 4 classes, 32 filters, the three discriminators, B = 32:
    bit-reproducible trajectory AND the first step's losses held to the CPU restatement of the reference step on the same
    batch (round-4 review, weak 1: this shape was only compared with the oracle at B = 4, golden `step_full256`)."""
    worst = _full_size_property_step(dev, dict(filters=32, in_channels=1, n_class=4, pointnet=True, fc_inch=121), 32, 256,
                                     "mscmrseg", None, 0.99, 31, False)
    print("config 3 full size (B = 32): worst first-step loss error against the CPU restatement %.2e" % worst)


def test_config5_standin_512_step(dev):
    """BASELINE config 5 names a 512x512 input on a DeepLab-v3+ backbone the reference does not contain (SURVEY section
    0).  STAND-IN, labelled as such: the reference's own segmenter at that size (fc_inch=729, unet.py:169-178) with the
    three discriminators, B = 8 per rank (64 / 8): the full step at 512x512, same properties as the other full-size
    configurations.  (The segmenter alone at this size is pinned by tests/golden/seg_full512.npz.)"""
    worst = _full_size_property_step(dev, dict(filters=32, in_channels=1, n_class=4, pointnet=True, fc_inch=729), 8, 512,
                                     "mscmrseg", None, 0.99, 51, False)
    print("config 5 stand-in (512x512): worst first-step loss error against the CPU restatement %.2e" % worst)


def test_mscmrseg_224_full_size_step(dev):
    """The reference's real MS-CMRSeg operating point at its batch scale: 224x224x3 input, fc_inch=81 (train_mscmrseg.py:412-414),
    the three discriminators, B = 32: bit-reproducible trajectory, first-step losses against the CPU restatement.  (The
    segmenter and one step at this shape are pinned by tests/golden/seg_full224.npz / step_full224.npz from the reference.)"""
    worst = _full_size_property_step(dev, dict(filters=32, in_channels=3, n_class=4, pointnet=True, fc_inch=81), 32, 224,
                                     "mscmrseg", None, 0.99, 61, False)
    print("MS-CMRSeg 224x224x3 B=32: worst first-step loss error against the CPU restatement %.2e" % worst)


def test_graph_replay_matches_eager_steps(dev):
    """AdversarialTrainer.step_graphed (hipGraph replay, Adam's step count on the device) walks the same
    parameter trajectory as eager steps: two trainers from identical weights, five identical batches."""
    from oracle.synth import synth_batch
    cfg_kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    cfg, tr_e = _build(cfg_kw, 7, dev)
    _, tr_g = _build(cfg_kw, 7, dev)
    batch = [torch.from_numpy(t).to(dev) for t in synth_batch(4, cfg.in_channels, cfg.n_class, 128, seed=300)]
    for it in range(5):
        out_e = tr_e.step(*batch)
        out_g = tr_g.step_graphed(*batch)
    assert getattr(tr_g, "_graph", None) is not None, "the step was not captured"
    he, hg = tr_e.to_host(out_e, tr_e.cfg), tr_g.to_host(out_g, tr_g.cfg)
    for k in ("seg_loss", "adv_loss"):
        assert abs(he[k] - hg[k]) <= 1e-5 * max(1.0, abs(he[k])), (k, he[k], hg[k])
    assert int(tr_g.opt_gen.step_t.item()) == int(tr_e.opt_gen.step_t.item()) == 5
    assert rel_err(tr_g.opt_gen.p, tr_e.opt_gen.p) < 1e-6
    assert rel_err(tr_g.opt_d4.p, tr_e.opt_d4.p) < 1e-6


def test_mmwhs_variant_step_vs_reference_golden(dev):
    """The MM-WHS loop (train_mmwhs.py:187-360; SURVEY config 4 in miniature: 3-channel input, 5 classes, softmax
    mode, PointNetCls(feature_transform=True, ext=True), discriminator momentum 0.95) on the HIP kernels against
    the loop re-typed around the reference modules."""
    from oracle.synth import synth_batch
    from pointcloududa_amd.train_step import AdversarialTrainer
    from test_networks_gpu import _strided
    g = np.load(os.path.join(GOLD, "step_mmwhs_small.npz"))
    seed, b, hw = int(g["seed"]), int(g["b"]), int(g["hw"])
    cfg_kw = dict(filters=4, in_channels=3, n_class=5, pointnet=True, fc_inch=9)
    cfg, tr = _build(cfg_kw, seed, dev, variant="mmwhs", pn_kw=dict(feature_transform=True, ext=True), momentum=0.95)
    batch = [torch.from_numpy(t).to(dev) for t in synth_batch(b, 3, 5, hw, seed=seed + 100)]
    out = tr.step(*batch, keep=True)
    h = AdversarialTrainer.to_host(out, tr.cfg)
    worst = _check_losses(h, g, "", tr.cfg, 0)
    last = tr.last
    assert rel_err(_strided(last["oS"]), g["oS_s"]) < 1e-3 and rel_err(_strided(last["oT"]), g["oT_s"]) < 1e-3
    assert rel_err(last["vertS"], g["vertS"]) < 1e-3 and rel_err(last["vertT"], g["vertT"]) < 1e-3
    for nm, mod, snap, lim in (("grad_seg", tr.gen, last["grad_seg"], 3e-2), ("grad_d1", tr.dis1, last["grad_d1"], 3e-2),
                               ("grad_d2", tr.dis2, last["grad_d2"], 3e-2), ("grad_d4", tr.dis4, last["grad_d4"], 0.15)):
        tot_ref = tot_got = 0.0
        for k, (off, n) in _flat_norms(None, mod).items():
            key = "%s_norm/%s" % (nm, k)
            if key in g:
                tot_ref += float(g[key]) ** 2; tot_got += float(snap[off:off + n].double().norm()) ** 2
        assert abs(tot_got ** 0.5 - tot_ref ** 0.5) <= lim * tot_ref ** 0.5, (nm, tot_got ** 0.5, tot_ref ** 0.5)
    rates = _check_updates(tr, g, "")
    print("mmwhs worst non-d4 loss error %.2e; update checks %s" % (worst, {k: round(v, 4) for k, v in rates.items()}))


@pytest.mark.parametrize("tag,dflags,flags", [
    ("step_mmwhs_etpls_small", (True, True, True), dict(etpls=True, Tetpls=True)),
    ("step_mmwhs_d4aux_sgd_small", (False, True, False), dict(d4aux=True, gen_sgd=True)),
])
def test_mmwhs_optional_branches_vs_reference_golden(dev, tag, dflags, flags):
    """The optional branches of the MM-WHS loop against the loop re-typed around the reference modules with the same
    flags: -etpls / -Tetpls (train_mmwhs.py:227-230,245-247: the entropy means enter the supervised and the adversarial
    loss), -d4aux (:220,248,256: point head trained, target point loss reported, no point-cloud discriminator) and
    -sgd (:453-459: SGD with momentum .95 and weight decay on the segmenter; parameters that never receive a gradient
    in the reference -- encoder.conv1_1 -- are skipped as torch.optim skips ``grad is None``)."""
    from oracle.synth import synth_batch
    from pointcloududa_amd.train_step import AdversarialTrainer
    from test_networks_gpu import _strided
    g = np.load(os.path.join(GOLD, tag + ".npz"))
    seed, b, hw = int(g["seed"]), int(g["b"]), int(g["hw"])
    d1, d2, d4 = dflags
    cfg_kw = dict(filters=4, in_channels=3, n_class=5, pointnet=True, fc_inch=9)
    cfg, tr = _build(cfg_kw, seed, dev, d1=d1, d2=d2, d4=d4, variant="mmwhs", pn_kw=dict(feature_transform=True, ext=True),
                     momentum=0.95, **flags)
    batch = [torch.from_numpy(t).to(dev) for t in synth_batch(b, 3, 5, hw, seed=seed + 100)]
    out = tr.step(*batch, keep=True)
    h = AdversarialTrainer.to_host(out, tr.cfg)
    for k in ("entropy_loss", "entropy_loss_T", "ver_s_loss", "ver_t_loss", "seg_loss", "adv_loss"):
        ref = float(g[k])
        assert abs(h[k] - ref) <= TIGHT * max(1e-3, abs(ref)), (k, h[k], ref)
    for k in ("d1_loss_src", "d1_loss_tgt", "d2_loss_src", "d2_loss_tgt", "d4_loss_src", "d4_loss_tgt"):
        if k in g:
            assert abs(h[k] - float(g[k])) <= TIGHT * max(1e-3, abs(float(g[k]))), (k, h[k], float(g[k]))
        else:
            assert k not in h, k
    last = tr.last
    assert rel_err(_strided(last["oS"]), g["oS_s"]) < 1e-3 and rel_err(_strided(last["oT"]), g["oT_s"]) < 1e-3
    assert rel_err(last["vertS"], g["vertS"]) < 1e-3 and rel_err(last["vertT"], g["vertT"]) < 1e-3
    # total norm of the supervised gradient (the adversarial one passes through PointNetCls's BatchNorm1d over a batch of
    # 4 behind a max over 300 points -- ill-conditioned between independent forward passes; the entropy terms' gradients
    # are held to 1e-4 / 3e-4 with the routing shared: test_train_step_backward_shared_routing[...etpls...])
    tot_ref = tot_got = 0.0
    for k, (off, n) in _flat_norms(None, tr.gen).items():
        key = "grad_seg_norm/%s" % k
        if key in g:
            tot_ref += float(g[key]) ** 2; tot_got += float(last["grad_seg"][off:off + n].double().norm()) ** 2
        else:
            assert float(last["grad_seg"][off:off + n].abs().max()) == 0.0, k
    assert abs(tot_got ** 0.5 - tot_ref ** 0.5) <= 3e-2 * tot_ref ** 0.5, (tot_got ** 0.5, tot_ref ** 0.5)
    if not flags.get("gen_sgd"):
        print(tag, _check_updates(tr, g, ""))
        return
    # SGD on the segmenter, first step: buf = g + wd * p0, p1 = p0 - lr * buf
    opt, lr = tr.opt_gen, tr.cfg.lr
    dot = n_got = n_ref = 0.0
    for (k, v), (off, n, shp) in zip(tr.gen.named_parameters(), opt._slices()):
        p0, p1 = _sample(tr._p0["gen"][k]), _sample(v)
        if "mb/gen/" + k not in g:                 # never receives a gradient in the reference: untouched
            assert k.startswith("encoder.conv1_1."), k
            assert np.array_equal(p0, p1) and float(opt.buf[off:off + n].abs().max()) == 0.0, k
            continue
        got, ref = _sample(opt.buf[off:off + n]), g["mb/gen/" + k].astype(np.float64)
        dot += float(got @ ref); n_got += float(got @ got); n_ref += float(ref @ ref)
        big = np.maximum(np.maximum(np.abs(p0), np.abs(p1)), np.abs(lr * got))
        ulp = np.spacing(big.astype(np.float32)).astype(np.float64)
        assert np.all(np.abs((p1 - p0) + lr * got) <= 1.51 * ulp), k          # moved by lr * buf
    cos = dot / max((n_got * n_ref) ** 0.5, 1e-30)
    assert cos >= 0.995 and abs(n_got ** 0.5 - n_ref ** 0.5) <= 3e-2 * n_ref ** 0.5, (cos, n_got, n_ref)
    print(tag, "segmenter SGD buffer cosine %.5f" % cos)


def test_step_with_rccl_collectives_in_a_one_rank_group(dev, monkeypatch):
    """The N > 1 wiring on one GPU: with PCUDA_FORCE_COLLECTIVES=1 the step issues its RCCL all-reduces (the
    segmenter's asynchronously, under the discriminator passes) in a one-rank nccl group; the trajectory must be
    the one of the collective-free step (sum over one rank, scale 1)."""
    import socket
    import torch.distributed as dist
    from oracle.synth import synth_batch
    if not dist.is_nccl_available():
        pytest.skip("no RCCL in this torch build")
    cfg_kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    cfg, tr_a = _build(cfg_kw, 11, dev)
    _, tr_b = _build(cfg_kw, 11, dev)
    batch = [torch.from_numpy(t).to(dev) for t in synth_batch(4, cfg.in_channels, cfg.n_class, 128, seed=301)]
    for _ in range(2):
        tr_a.step(*batch)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.setenv("PCUDA_FORCE_COLLECTIVES", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        work, scale = tr_b.opt_gen.all_reduce_grads_async()
        assert work is not None and scale == 1.0
        tr_b.opt_gen.finish_all_reduce(work)
        # the bucketed form: the non-encoder tail goes out from inside the adversarial backward pass, the encoder's
        # head after it -- count the collectives of one step and check that the two slices tile the buffer
        split = tr_b.opt_gen.split_after("encoder.")
        assert 0 < split < tr_b.opt_gen.g.numel()
        calls = []
        real = tr_b.opt_gen.all_reduce_grads_async
        monkeypatch.setattr(tr_b.opt_gen, "all_reduce_grads_async",
                            lambda group=None, lo=0, hi=None: (calls.append((lo, hi)), real(group, lo, hi))[1])
        for _ in range(2):
            tr_b.step(*batch)
        torch.cuda.synchronize()
        assert calls == [(split, None), (0, split)] * 2, calls
    finally:
        dist.destroy_process_group()
    assert rel_err(tr_b.opt_gen.p, tr_a.opt_gen.p) < 1e-6 and rel_err(tr_b.opt_d1.p, tr_a.opt_d1.p) < 1e-6


def test_graph_replay_with_rccl_collectives_in_a_one_rank_group(dev, monkeypatch):
    """hipGraph replay of the step in a process group (data parallel: 8 ranks x ~850 launches per step share one host; a
    replay is one launch).  A collective cannot be captured here (AdversarialTrainer.step_graphed), so the step replays
    as graph A (phases 1-4) + four eager all-reduces + graph B (phase 5): in a one-rank nccl group with
    PCUDA_FORCE_COLLECTIVES=1 it must walk the eager trajectory."""
    import socket
    import torch.distributed as dist
    from oracle.synth import synth_batch
    if not dist.is_nccl_available():
        pytest.skip("no RCCL in this torch build")
    cfg_kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    cfg, tr_e = _build(cfg_kw, 13, dev)
    _, tr_g = _build(cfg_kw, 13, dev)
    batch = [torch.from_numpy(t).to(dev) for t in synth_batch(4, cfg.in_channels, cfg.n_class, 128, seed=302)]
    for _ in range(5):
        out_e = tr_e.step(*batch)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1"); monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.setenv("PCUDA_FORCE_COLLECTIVES", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        calls = []
        real = tr_g.opt_gen.all_reduce_grads_async
        monkeypatch.setattr(tr_g.opt_gen, "all_reduce_grads_async",
                            lambda group=None, lo=0, hi=None: (calls.append((lo, hi)), real(group, lo, hi))[1])
        for _ in range(5):
            out_g = tr_g.step_graphed(*batch)
        torch.cuda.synchronize()
        assert getattr(tr_g, "_graph", None) is not None and tr_g._graph_b is not None, "the step with collectives was not captured"
        # two eager steps (two buckets each); the capture issues none; each of the three replays one whole-buffer all-reduce
        assert calls == [(tr_g.opt_gen.split_after("encoder."), None), (0, tr_g.opt_gen.split_after("encoder."))] * 2 + [(0, None)] * 3, calls
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()
    he, hg = tr_e.to_host(out_e, tr_e.cfg), tr_g.to_host(out_g, tr_g.cfg)
    for k in ("seg_loss", "adv_loss"):
        assert abs(he[k] - hg[k]) <= 1e-5 * max(1.0, abs(he[k])), (k, he[k], hg[k])
    assert int(tr_g.opt_gen.step_t.item()) == 5
    assert rel_err(tr_g.opt_gen.p, tr_e.opt_gen.p) < 1e-6 and rel_err(tr_g.opt_d2.p, tr_e.opt_d2.p) < 1e-6


def test_full_size_step_is_reproducible_and_stream_schedule_keeps_the_arithmetic(dev):
    """BASELINE config 3 at full size (B=32, 256x256, 32 filters, three discriminators).  (a) Two trainers from the
    same weights walk a BIT-IDENTICAL trajectory over three steps with the concurrent-stream schedule on: every
    kernel is deterministic (fixed-order split-K reductions, no atomics), so any difference would be a race between
    streams.  (b) The single-stream schedule with separate source / target discriminator passes (the reference's
    order of operations) gives the same losses and parameters up to fp32 summation order."""
    from oracle.synth import synth_batch
    cfg_kw = dict(filters=32, in_channels=1, n_class=4, pointnet=True, fc_inch=121)
    cfg, tr_a = _build(cfg_kw, 21, dev)
    _, tr_b = _build(cfg_kw, 21, dev)
    _, tr_c = _build(cfg_kw, 21, dev)
    tr_c.d_streams = tr_c.d_overlap = tr_c.d_batch = tr_c.early_fwd2 = False
    batches = [[torch.from_numpy(t).to(dev) for t in synth_batch(32, 1, 4, 256, seed=900 + i)] for i in range(3)]
    outs = []
    for tr in (tr_a, tr_b, tr_c):
        for b in batches:
            out = tr.step(*b)
        torch.cuda.synchronize()
        outs.append(tr.to_host(out, tr.cfg))
    for opt in ("opt_gen", "opt_d1", "opt_d2", "opt_d4"):
        assert torch.equal(getattr(tr_a, opt).p, getattr(tr_b, opt).p), opt
    assert outs[0] == outs[1]
    for k in ("seg_loss", "adv_loss", "d1_loss_src", "d2_loss_tgt", "d4_loss_src"):
        assert abs(outs[0][k] - outs[2][k]) <= 2e-3 * max(1e-3, abs(outs[2][k])), (k, outs[0][k], outs[2][k])
    # three Adam steps from identical weights: lr * sign-like updates amplify summation-order noise on near-zero
    # gradients, so parameters are compared in aggregate
    for opt, lim in (("opt_gen", 2e-3), ("opt_d1", 1e-5), ("opt_d2", 1e-5), ("opt_d4", 1e-3)):
        assert rel_err(getattr(tr_a, opt).p, getattr(tr_c, opt).p) < lim, opt
