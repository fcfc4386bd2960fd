"""CPU tests: the oracle (CPU restatement of the reference) against the golden vectors that
oracle/make_golden.py produced from the REAL reference modules.  No GPU needed."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, rel_err


def _g(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def test_param_inventories_match_reference_counts():
    from oracle import nets as ON
    g = _g("param_counts")
    cnt = lambda shapes: sum(int(np.prod(s)) for k, s in shapes.items() if ON.is_trainable(k))
    assert cnt(ON.seg_param_shapes(ON.SegCfg(pointnet=True, fc_inch=81))) == int(g["seg_pointnet_fc81"]) == 19013990
    assert cnt(ON.seg_param_shapes(ON.SegCfg(pointnet=False))) == int(g["seg_nopoint"]) == 13483844
    assert cnt(ON.seg_param_shapes(ON.SegCfg(n_class=5, pointnet=True, fc_inch=121))) == int(g["seg_5class_fc121"])
    assert cnt(ON.disc_param_shapes(4)) == int(g["disc4"]) == 2764800
    assert cnt(ON.disc_param_shapes(5, True)) == int(g["disc5_ext"])
    assert cnt(ON.pointnet_cls_param_shapes()) == int(g["pncls"]) == 1604106
    assert cnt(ON.pointnet_cls_param_shapes(True, ext=True)) == int(g["pncls_ft_ext"])


def test_hip_modules_keep_reference_state_dict_keys():
    """the drop-in contract: same keys and shapes as the reference modules (no GPU needed to build them)"""
    from oracle import nets as ON
    from pointcloududa_amd.networks import PointNetCls, Segmentation_model_Point, UncertaintyDiscriminator
    for kw in (dict(filters=32, in_channels=3, n_class=4, pointnet=True, fc_inch=81),
               dict(filters=8, in_channels=1, n_class=5, pointnet=False)):
        m = Segmentation_model_Point(**kw)
        shapes = ON.seg_param_shapes(ON.SegCfg(**kw))
        assert list(m.state_dict().keys()) == list(shapes.keys())
        assert all(tuple(v.shape) == shapes[k] for k, v in m.state_dict().items())
    for inch, ext in ((4, False), (5, True)):
        m = UncertaintyDiscriminator(in_channel=inch, ext=ext)
        shapes = ON.disc_param_shapes(inch, ext)
        assert list(m.state_dict().keys()) == list(shapes.keys())
        assert all(tuple(v.shape) == shapes[k] for k, v in m.state_dict().items())
    for ft, ext in ((False, False), (True, True)):
        m = PointNetCls(feature_transform=ft, ext=ext)
        shapes = ON.pointnet_cls_param_shapes(ft, ext=ext)
        assert list(m.state_dict().keys()) == list(shapes.keys())
        assert all(tuple(v.shape) == shapes[k] for k, v in m.state_dict().items())


def test_product_has_no_cpu_fallback():
    from pointcloududa_amd.networks import Segmentation_model_Point, UncertaintyDiscriminator
    from pointcloududa_amd import kernels as K
    m = Segmentation_model_Point(filters=4, in_channels=1, pointnet=False)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 1, 32, 32))
    with pytest.raises(RuntimeError):
        UncertaintyDiscriminator(4)(torch.zeros(2, 4, 32, 32))
    with pytest.raises(RuntimeError):
        K.entropy_fwd(torch.zeros(1, 4, 8, 8))
    with pytest.raises(RuntimeError):
        m.encoder.encoder1[0](torch.zeros(1, 1, 8, 8))      # holders never run ATen ops


def test_oracle_losses_vs_reference_golden():
    from oracle import losses as OL
    g = _g("losses")
    rng = np.random.default_rng(int(g["seed"]))
    b, c, hw = 2, 4, 32
    logits = torch.from_numpy(rng.normal(0, 2, (b, c, hw, hw)).astype(np.float32))
    lab = rng.integers(0, c, (b, hw, hw))
    onehot = torch.from_numpy(np.moveaxis(np.eye(c, dtype=np.uint8)[lab], -1, 1).copy())
    x = torch.from_numpy(rng.random((3, 300, 3), dtype=np.float32))
    y = torch.from_numpy(rng.random((3, 300, 3), dtype=np.float32))
    l = logits.clone().requires_grad_(True)
    m, j = OL.seg_loss_sigmoid(l, onehot); (m + j).backward()
    assert abs(float(m) - float(g["bce"])) < 1e-6 and abs(float(j) - float(g["jac"])) < 1e-6
    assert rel_err(l.grad, g["dlogits_sig"]) < 1e-5
    l = logits.clone().requires_grad_(True)
    m, j = OL.seg_loss_softmax(l, onehot); (m + j).backward()
    assert abs(float(m) - float(g["ce"])) < 1e-6 and abs(float(j) - float(g["jac_sm"])) < 1e-6
    assert rel_err(l.grad, g["dlogits_sm"]) < 1e-5
    for name, mode, norm in (("ent_sig", "sigmoid", False), ("ent_sm_n", "softmax", True), ("ent_sig_n", "sigmoid", True)):
        assert rel_err(OL.entropy_map(logits, mode, norm), g[name]) < 1e-6
    xr = x.clone().requires_grad_(True)
    nn = OL.batch_nn_loss(xr, y); nn.backward()
    assert abs(float(nn) - float(g["nn"])) < 1e-6 and rel_err(xr.grad, g["nn_dx"]) < 1e-3


def test_oracle_fps_bit_exact_vs_reference_golden():
    from oracle import sampler as OS
    g = _g("fps")
    for name in ("rand", "lattice", "dup", "surface"):
        for trial in range(2):
            idx = OS.fps_indices(g[name + "_pts"], 300, int(g["%s_%d_first" % (name, trial)]))
            assert np.array_equal(idx, g["%s_%d_idx" % (name, trial)])
    assert np.array_equal(OS.surface_vertices(g["surface_mask"]).astype(np.float64), g["surface_pts"])


def test_oracle_sampler_edge_cases():
    from oracle import sampler as OS
    assert OS.surface_vertices(np.zeros((16, 16), np.uint8)).shape == (0, 3)            # empty mask
    assert OS.surface_vertices(np.ones((16, 16), np.uint8)).shape == (0, 3)             # no background at all
    m = np.zeros((32, 32, 1), np.int64); m[4:8, 4:8] = 2
    assert (OS.mask_to_pointcloud(m) == 0).all()                                        # <= 50 px: zeros (npy2point.py:116)
    m[4:20, 4:20] = 1
    v = OS.mask_to_pointcloud(m, 300, first=7)
    assert v.shape == (300, 3) and set(np.unique(v[:, 0])) <= {0, 1, 2}
    # one-pixel-wide vertical line: every neighbour column is boundary background
    line = np.zeros((8, 8), np.uint8); line[:, 3] = 1
    sv = OS.surface_vertices(line)
    assert len(sv) == 3 * 16 and (sv[:16, 0] == 0).all()


def _mc_walk(fg):
    """cell-by-cell restatement of the marching-cubes-order mode: the first cell that contains a crossing edge emits it"""
    from oracle import sampler as OS
    h, w = fg.shape
    vol = np.broadcast_to(fg[None], (3, h, w))
    out, seen = [], set()
    for i in range(2):
        for j in range(h - 1):
            for k in range(w - 1):
                for e in OS._MC_ORDER:
                    a, b = OS._MC_CORNER[OS._MC_EDGE[e][0]], OS._MC_CORNER[OS._MC_EDGE[e][1]]
                    pa, pb = (i + a[0], j + a[1], k + a[2]), (i + b[0], j + b[1], k + b[2])
                    if vol[pa] == vol[pb] or frozenset((pa, pb)) in seen:
                        continue
                    seen.add(frozenset((pa, pb)))
                    out.append(pb if vol[pa] else pa)
    return np.array(out, dtype=np.int64).reshape(-1, 3)


def test_oracle_sampler_marching_cubes_order_mode():
    """the second vertex-list mode (parity unpinned): one vertex per crossing edge, in cell traversal order"""
    from oracle import sampler as OS
    rng = np.random.default_rng(0)
    for t in range(4):
        m = (rng.random((20, 23)) > 0.6).astype(np.uint8)
        if t == 3:
            m[:, 0] = 1; m[0, :] = 1; m[-1, :] = 0          # foreground on the volume faces
        v, fg = OS.surface_vertices_mc(m), m > 0
        assert np.array_equal(v, _mc_walk(fg)), t
        assert len(v) == 3 * ((fg[1:] != fg[:-1]).sum() + (fg[:, 1:] != fg[:, :-1]).sum())
        assert set(map(tuple, v)) == set(map(tuple, OS.surface_vertices(m)))      # same points, other order / multiplicity
    assert OS.surface_vertices_mc(np.zeros((16, 16), np.uint8)).shape == (0, 3)
    assert OS.surface_vertices_mc(np.ones((16, 16), np.uint8)).shape == (0, 3)
    sq = np.pad(np.ones((12, 12), np.uint8), 6)[..., None]
    v = OS.mask_to_pointcloud(sq, 300, first=5, order="mc")
    assert v.shape == (300, 3) and set(np.unique(v[:, 0])) <= {0, 1, 2}
    assert not np.array_equal(v, OS.mask_to_pointcloud(sq, 300, first=5))


def test_oracle_metrics_hand_cases():
    from oracle import metrics as OM
    pred = np.zeros((1, 3, 2, 2), np.float32)
    pred[0, 1] = [[2, 0], [0, 0]]; pred[0, 2] = [[0, 0], [0, 3]]
    hard = OM.soft_to_hard_pred(pred, 1)
    assert hard[0, :, 0, 0].tolist() == [0, 1, 0] and hard[0, :, 0, 1].tolist() == [1, 1, 1]   # tie marks all
    y = np.zeros((1, 3, 2, 2), np.uint8); y[0, 1, 0, 0] = 1; y[0, 2, 1, 1] = 1; y[0, 0, 0, 1] = 1; y[0, 0, 1, 0] = 1
    # label 1: |A.B| = 1, |A| = 1, |B| = 3 (the two tie pixels count) -> (2+1)/(1+3+1); same for label 2
    assert abs(OM.dice_coef_multilabel(y, hard, 3) - 0.6) < 1e-12
    assert OM.disc_accuracy(np.array([2.0, -1.0, 0.0, 3.0]), True) == 0.75
    assert OM.disc_accuracy(np.array([2.0, -1.0, 0.0, 3.0]), False) == 0.25


@pytest.mark.parametrize("tag,kw,softmax", [
    ("seg_small", dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9), False),
    ("seg_small_3ch_nopoint", dict(filters=8, in_channels=3, n_class=5, pointnet=False), True)])
def test_oracle_segmenter_vs_reference_golden(tag, kw, softmax):
    from oracle import losses as OL
    from oracle import nets as ON
    from oracle.synth import synth_batch
    g = _g(tag)
    seed, b, hw = int(g["seed"]), int(g["b"]), int(g["hw"])
    cfg = ON.SegCfg(**kw)
    p = ON.make_params(ON.seg_param_shapes(cfg), seed)
    p = {k: (v.requires_grad_(True) if ON.is_trainable(k) else v) for k, v in p.items()}
    img, mask, vert, _, _ = synth_batch(b, cfg.in_channels, cfg.n_class, hw, seed=seed + 1)
    lo, ve = ON.seg_forward(p, torch.from_numpy(img), cfg, True)
    m, j = (OL.seg_loss_softmax if softmax else OL.seg_loss_sigmoid)(lo, torch.from_numpy(mask))
    loss = m + j + (OL.batch_nn_loss(ve, torch.from_numpy(vert)) if cfg.pointnet else 0.0)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert rel_err(lo, g["logits"]) < 1e-5
    for k in p:
        if "g/" + k in g:
            assert rel_err(p[k].grad, g["g/" + k]) < 5e-4, k
        if "bn/" + k in g:
            assert rel_err(p[k], g["bn/" + k]) < 1e-5, k
    assert p["encoder.conv1_1.0.weight"].grad is None       # unet.py:45: conv1_1 never runs


def test_oracle_step_vs_reference_golden():
    """the 5-phase step restatement against the loop re-typed around the reference modules"""
    from oracle import nets as ON
    from oracle.step import OracleTrainer, StepCfg
    from oracle.synth import synth_batch
    g = _g("step_small")
    seed, b, hw = int(g["seed"]), int(g["b"]), int(g["hw"])
    cfg = ON.SegCfg(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    orc = OracleTrainer(cfg, StepCfg(n_class=4), ON.make_params(ON.seg_param_shapes(cfg), seed),
                        ON.make_params(ON.disc_param_shapes(4), seed + 1, std=0.02),
                        ON.make_params(ON.disc_param_shapes(4), seed + 2, std=0.02),
                        ON.make_params(ON.pointnet_cls_param_shapes(), seed + 3))
    out = orc.step(*synth_batch(b, 1, 4, hw, seed=seed + 100), keep=True)
    for k in ("seg_loss", "ver_s_loss", "ver_t_loss", "adv_loss", "d2_loss_src", "d1_loss_src", "d4_loss_src",
              "d2_loss_tgt", "d1_loss_tgt", "d4_loss_tgt"):
        assert abs(out[k] - float(g["s0/" + k])) <= 2e-5 * max(1.0, abs(float(g["s0/" + k]))), k
    assert rel_err(orc.kept["oS"], g["s0/oS"]) < 1e-4 and rel_err(orc.kept["oT"], g["s0/oT"]) < 1e-4


def test_oracle_validation_loop_matches_reference_golden():
    """oracle.validate.valid_batch (train_mscmrseg.py:67-92 restated) against the values the REFERENCE model
    produced in eval mode (oracle/make_golden.py:gold_valid)."""
    from oracle import nets as ON
    from oracle import validate as OV
    from oracle.synth import synth_batch
    g = np.load(os.path.join(GOLD, "valid_small.npz"))
    cfg = ON.SegCfg(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    seed, b, hw = int(g["seed"]), int(g["b"]), int(g["hw"])
    params = ON.make_params(ON.seg_param_shapes(cfg), seed)
    img, mask, vert, _, _ = synth_batch(b, cfg.in_channels, cfg.n_class, hw, seed=seed + 1)
    o = OV.valid_batch(params, img, mask, vert, cfg)
    assert rel_err(o["logits"], g["logits"]) < 1e-5
    assert abs(o["loss"] - float(g["loss"])) < 1e-5 and abs(o["vert_loss"] - float(g["vert_loss"])) < 1e-6
    assert np.array_equal(o["labels"], g["labels"])
    assert np.allclose(o["dice_per_class"][1:4], g["dice_per_class"], atol=1e-12)


def test_validation_metrics_known_answers():
    from oracle import metrics as OM
    logits = np.zeros((1, 3, 1, 4), dtype=np.float32)
    logits[0, :, 0, 0] = [0.2, 0.9, 0.9]      # tie between channels 1 and 2 -> first (1)
    logits[0, :, 0, 1] = [5.0, -1.0, 2.0]
    logits[0, :, 0, 2] = [-3.0, -2.0, -1.0]
    logits[0, :, 0, 3] = [0.0, 0.0, 0.0]      # all equal -> 0
    assert OM.argmax_labels(logits).tolist() == [[[1, 0, 2, 0]]]
    pred = np.array([0, 1, 1, 2, 2, 2], dtype=np.uint8)
    gt = np.array([0, 1, 2, 2, 2, 0], dtype=np.uint8)
    dc = OM.label_dice(pred, gt, 4)
    assert np.allclose(dc, [2 * 1 / 3, 2 * 1 / 3, 2 * 2 / 6, 0.0])   # class 3 empty on both sides -> 0


def test_batch_assembly_known_answers():
    """oracle.batch (data_generator_mmwhs.py:265-274, utils.py:7-29) on hand-checkable inputs."""
    from oracle import batch as OB
    img = np.arange(2 * 6 * 6 * 3, dtype=np.float32).reshape(2, 6, 6, 3)
    m = (np.arange(2 * 6 * 6).reshape(2, 6, 6, 1) % 5).astype(np.int64)
    im, oh, v = OB.assemble_batch(img, m, [[[255, 0, 51]]] * 2, num_classes=5, crop_size=4)
    assert im.shape == (2, 3, 4, 4) and oh.shape == (2, 5, 4, 4) and oh.dtype == np.uint8
    assert im[1, 2, 0, 0] == img[1, 1, 1, 2] and im[0, 0, 3, 3] == img[0, 4, 4, 0]      # rows/cols 1..4 survive
    assert oh.sum(1).min() == 1 and oh.sum(1).max() == 1 and oh[0, m[0, 1, 1, 0], 0, 0] == 1
    assert np.allclose(v[0, 0], [1.0, 0.0, 0.2])


def test_oracle_mmwhs_step_vs_reference_golden():
    """the MM-WHS variant of the step (train_mmwhs.py:187-360: softmax + double-softmax CE, normalised entropy
    map, w1/w2/w4, PointNetCls(feature_transform, ext)) against the loop re-typed around the reference modules"""
    from oracle import nets as ON
    from oracle.step import OracleTrainer, StepCfg
    from oracle.synth import synth_batch
    g = _g("step_mmwhs_small")
    seed, b, hw = int(g["seed"]), int(g["b"]), int(g["hw"])
    cfg = ON.SegCfg(filters=4, in_channels=3, n_class=5, pointnet=True, fc_inch=9)
    scfg = StepCfg(variant="mmwhs", n_class=5, softmax=True, d_momentum=0.95, pn_feature_transform=True, pn_ext=True)
    orc = OracleTrainer(cfg, scfg, ON.make_params(ON.seg_param_shapes(cfg), seed),
                        ON.make_params(ON.disc_param_shapes(5), seed + 1, std=0.02),
                        ON.make_params(ON.disc_param_shapes(5), seed + 2, std=0.02),
                        ON.make_params(ON.pointnet_cls_param_shapes(feature_transform=True, ext=True), seed + 3))
    out = orc.step(*synth_batch(b, 3, 5, hw, seed=seed + 100), keep=True)
    for k in ("seg_loss", "ver_s_loss", "ver_t_loss", "adv_loss", "d2_loss_src", "d1_loss_src", "d4_loss_src",
              "d2_loss_tgt", "d1_loss_tgt", "d4_loss_tgt"):
        assert abs(out[k] - float(g[k])) <= 2e-5 * max(1.0, abs(float(g[k]))), k
    f = orc.kept["oS"].reshape(-1)
    assert rel_err(f[::max(1, f.numel() // 4096)][:4096], g["oS_s"]) < 1e-4
    assert rel_err(orc.kept["vertS"], g["vertS"]) < 1e-4
