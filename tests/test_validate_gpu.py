"""Validation / inference forward on the HIP kernels: eval-mode BatchNorm folding, label maps
and per-class Dice against the reference's eval-mode golden and the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, rel_err

pytestmark = pytest.mark.gpu


def test_label_kernels_bit_exact_against_oracle(dev):
    from oracle import metrics as OM
    from pointcloududa_amd.utils import metric as M
    rng = np.random.default_rng(5)
    logits = rng.normal(0, 1, (3, 5, 37, 41)).astype(np.float32)
    logits[:, :, ::3, ::2] = np.round(logits[:, :, ::3, ::2])          # plenty of exact ties
    lab = M.argmax_labels(torch.from_numpy(logits).to(dev))
    assert lab.dtype == torch.uint8 and np.array_equal(lab.cpu().numpy(), OM.argmax_labels(logits))
    onehot = np.moveaxis(np.eye(5, dtype=np.uint8)[rng.integers(0, 4, (3, 37, 41))], -1, 1).copy()   # class 4 empty
    gt = M.argmax_labels(torch.from_numpy(onehot).to(dev))
    assert np.array_equal(gt.cpu().numpy(), OM.argmax_labels(onehot))
    dc = M.label_dice(lab, gt, 5).cpu().numpy()
    ref = OM.label_dice(OM.argmax_labels(logits), OM.argmax_labels(onehot), 5)
    assert np.allclose(dc, ref, atol=1e-6)


def _setup(dev):
    from oracle import nets as ON
    from oracle.synth import synth_batch
    from pointcloududa_amd.networks import Segmentation_model_Point
    g = np.load(os.path.join(GOLD, "valid_small.npz"))
    cfg_kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    cfg = ON.SegCfg(**cfg_kw)
    seed, b, hw = int(g["seed"]), int(g["b"]), int(g["hw"])
    params = ON.make_params(ON.seg_param_shapes(cfg), seed)
    m = Segmentation_model_Point(**cfg_kw)
    m.load_state_dict({k: v.clone() for k, v in params.items()})
    m = m.to(dev).eval()
    img, mask, vert, _, _ = synth_batch(b, cfg.in_channels, cfg.n_class, hw, seed=seed + 1)
    return g, m, [torch.from_numpy(t).to(dev) for t in (img, mask, vert)]


def test_eval_forward_and_validation_batch_vs_reference_golden(dev):
    from pointcloududa_amd import validate as V
    g, m, (x, y, z) = _setup(dev)
    with torch.no_grad():
        logits, _, verts = m(x)
    assert rel_err(logits, g["logits"]) < 1e-3           # eval-mode BN folded from the running statistics
    assert rel_err(verts, g["verts"]) < 1e-3
    r = V.valid_batch(m, x, y, z)
    assert abs(float(r["loss"]) - float(g["loss"])) < 1e-3 * max(1.0, abs(float(g["loss"])))
    assert abs(float(r["vert_loss"]) - float(g["vert_loss"])) < 1e-3
    lab = V.predict_labels(m, x).cpu().numpy()
    gl = g["labels"]
    # a pixel whose two best logits differ by less than the forward tolerance may legitimately flip
    srt = np.sort(g["logits"], axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 2e-3 * np.abs(g["logits"]).max()
    assert np.array_equal(lab[decided], gl[decided]) and decided.mean() > 0.99
    dcs = r["dice_per_class"].cpu().numpy()[1:4]
    assert np.allclose(dcs, g["dice_per_class"], atol=5e-3)
    assert m.training is False


def test_valid_model_with_one_dataset_means(dev):
    from pointcloududa_amd import validate as V
    g, m, (x, y, z) = _setup(dev)
    m.train()
    out = V.valid_model_with_one_dataset(m, [(x, y, z), (x, y, z)])
    assert m.training is True                                  # restored
    assert abs(out["loss"] - float(g["loss"])) < 1e-3 * max(1.0, abs(float(g["loss"])))
    assert abs(out["dice"] - float(np.mean(g["dice_per_class"]))) < 5e-3
    assert abs(out["valid_vert_loss"] - float(g["vert_loss"])) < 1e-3
