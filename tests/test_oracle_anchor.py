"""CPU check of the shared-routing anchor of oracle.nets (used by tests/test_backward_exact_gpu.py): anchoring a
perturbed copy of the network to the routing of the unperturbed one reproduces the unperturbed gradients."""
import numpy as np
import torch

from oracle import losses as OL
from oracle import nets as ON
from oracle.synth import synth_batch


def _record():
    rec = {}

    def fn(tag, z):
        rec[tag] = z.detach().clone()
        return z
    return rec, fn


def test_anchor_shares_routing_between_two_forward_passes():
    cfg = ON.SegCfg(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=1)
    params = ON.make_params(ON.seg_param_shapes(cfg), 5)
    img, mask, vert, _, _ = synth_batch(2, 1, 4, 96, seed=6)

    def run(anchor, noise):
        p = {k: (v.clone().requires_grad_(True) if ON.is_trainable(k) else v.clone()) for k, v in params.items()}
        x = torch.from_numpy(img) + noise
        with ON.anchored(anchor):
            lo, ve = ON.seg_forward(p, x, cfg, training=True)
        m, j = OL.seg_loss_sigmoid(lo, torch.from_numpy(mask))
        (m + j + OL.batch_nn_loss(ve, torch.from_numpy(vert))).backward()
        return {k: v.grad for k, v in p.items() if ON.is_trainable(k) and v.grad is not None}

    rec, fn = _record()
    g0 = run(fn, 0.0)
    assert "encoder.encoder1.0" in rec and "pointNet.final_fc" in rec and "classifier" in rec and len(rec) == 30
    # a forward pass whose input differs by 1e-5 takes other branches on a few elements; anchored to the recorded
    # values it reproduces the recorded run's gradients (except the first conv's, whose INPUT really differs)
    g_free = run(None, 1e-5)
    g_anch = run(lambda tag, z: z + (rec[tag] - z).detach(), 1e-5)
    worst_free = worst_anch = 0.0
    for k in g0:
        if k.startswith("encoder.encoder1.0"):
            continue
        s = float(g0[k].abs().max()) + 1e-30
        worst_free = max(worst_free, float((g_free[k] - g0[k]).abs().max()) / s)
        worst_anch = max(worst_anch, float((g_anch[k] - g0[k]).abs().max()) / s)
    assert worst_anch < 1e-6, worst_anch
    assert worst_free > worst_anch


def test_anchor_tags_cover_discriminator_and_pointnet():
    rec, fn = _record()
    pd = ON.make_params(ON.disc_param_shapes(5, True), 1, std=0.02)
    with ON.anchored(fn):
        ON.disc_forward(pd, torch.zeros(1, 5, 64, 64), ext=True)
    assert set(rec) == {"conv1", "conv2", "conv3", "conv4", "conv4_2", "conv4_3", "conv5"}
    rec.clear()
    pp = ON.make_params(ON.pointnet_cls_param_shapes(True, ext=True), 2)
    x = torch.from_numpy(np.random.default_rng(0).random((3, 3, 300), dtype=np.float32))
    with ON.anchored(fn):
        ON.pointnet_cls_forward(pp, x, feature_transform=True, ext=True, drop=0.0, training=True)
    for k in ("feat.stn.conv1", "feat.stn.bn3", "feat.stn.fc3", "feat.fstn.bn5", "feat.conv3_1", "feat.bn3_1", "fc1", "bn2"):
        assert k in rec, k
