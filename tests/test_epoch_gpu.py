"""train_epoch / valid_model with the reference's signatures (train_mscmrseg.py:102-345; train_mmwhs.py:102-377) on the HIP
path, fed the way the scripts feed them: host-numpy generators and torch.optim objects built by the caller."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(dev, cfg_kw, seed, pn_kw=None):
    from oracle import nets as ON
    from pointcloududa_amd.networks import PointNetCls, Segmentation_model_Point, UncertaintyDiscriminator
    cfg = ON.SegCfg(**cfg_kw)
    pn_kw = pn_kw or {}
    ps = (ON.make_params(ON.seg_param_shapes(cfg), seed), ON.make_params(ON.disc_param_shapes(cfg.n_class), seed + 1, std=0.02),
          ON.make_params(ON.disc_param_shapes(cfg.n_class), seed + 2, std=0.02),
          ON.make_params(ON.pointnet_cls_param_shapes(**pn_kw), seed + 3))
    load = lambda m, p: (m.load_state_dict({k: v.clone() for k, v in p.items()}), m.to(dev).train())[1]
    mods = (load(Segmentation_model_Point(**cfg_kw), ps[0]), load(UncertaintyDiscriminator(in_channel=cfg.n_class), ps[1]),
            load(UncertaintyDiscriminator(in_channel=cfg.n_class), ps[2]), load(PointNetCls(drop=0.0, **pn_kw), ps[3]))
    return cfg, ps, mods


def _gen(batches, which):
    """a generator in the reference's batch layout (data_generator_mscmrseg.py:274-319): numpy x, one-hot uint8 y, z"""
    for img_a, mask_a, vert_a, img_b, vert_b in batches:
        if which == "A":
            yield img_a, mask_a, vert_a
        else:
            yield img_b, np.zeros_like(mask_a), vert_b


@pytest.mark.parametrize("variant", ["mscmrseg", "mmwhs"])
def test_train_epoch_with_reference_signature_matches_the_cpu_restatement(dev, variant):
    from oracle.step import OracleTrainer, StepCfg
    from oracle.synth import synth_batch
    import pointcloududa_amd.train_mmwhs as TW
    import pointcloududa_amd.train_mscmrseg as TM
    ms = variant == "mscmrseg"
    T = TM if ms else TW
    cin, nc = (1, 4) if ms else (3, 5)
    cfg_kw = dict(filters=4, in_channels=cin, n_class=nc, pointnet=True, fc_inch=9)
    mom = 0.99 if ms else 0.95
    cfg, ps, (gen, d1, d2, d4) = _nets(dev, cfg_kw, 2100 + cin)
    # the caller's own optimisers, built as the scripts build them (train_mscmrseg.py:427-455)
    og = torch.optim.Adam(gen.parameters(), lr=1e-3, betas=(0.9, 0.99))
    mk = lambda m: torch.optim.SGD(m.parameters(), lr=2.5e-5, momentum=mom, weight_decay=0.0005)
    o1, o2, o4 = mk(d1), mk(d2), mk(d4)
    T.args = types.SimpleNamespace(d1=True, d2=True, d4=True, dr=0.01, wp=1.0, softmax=True, w1=1.0, w2=1.0, w4=1.0,
                                   etpls=False, Tetpls=False, d4aux=False)
    b, hw = 4, 128
    # epoch 0: ONE batch (starts from identical parameters: tight); epoch 1: four batches on Adam-updated parameters
    # (sign-sensitive for near-zero gradients; BatchNorm1d over 4 clouds in d4), held like the second step of the step tests
    epochs = [[synth_batch(b, cin, nc, hw, seed=2200 + 10 * e + i, gaussian=not ms) for i in range(n)] for e, n in enumerate((1, 4))]
    orc = OracleTrainer(cfg, StepCfg(variant=variant, n_class=nc, d_momentum=mom), *ps)
    for e, batches in enumerate(epochs):
        res = T.train_epoch(gen, d2, d4, d1, og, o2, o4, o1, _gen(batches, "A"), _gen(batches, "B"))
        want_keys = {"seg_loss", "seg_dice", "dis1_acc1", "dis1_acc2", "dis2_acc1", "dis2_acc2", "dis4_acc1", "dis4_acc2",
                     "ver_s_loss", "ver_t_loss"} | (set() if ms else {"entropy_loss", "entropy_loss_T"})
        assert set(res) == want_keys and all(isinstance(v, float) and np.isfinite(v) for v in res.values())
        qs = [orc.step(*bt) for bt in batches]
        mean = lambda k: float(np.mean([q[k] for q in qs]))
        tol = 1e-3 if e == 0 else 1e-1
        for k in ("seg_loss", "ver_s_loss", "ver_t_loss") + (() if ms else ("entropy_loss", "entropy_loss_T")):
            assert abs(res[k] - mean(k)) <= tol * max(1e-3, abs(mean(k))), (e, k, res[k], mean(k))
        assert abs(res["seg_dice"] - mean("seg_dice")) <= (1e-4 if e == 0 else 5e-2)
        for d in ("dis1", "dis2", "dis4"):
            assert 0.0 <= res[d + "_acc1"] <= 1.0 and 0.0 <= res[d + "_acc2"] <= 1.0
            if e == 0:
                # (at initialisation D(x) ~ 0: a few of the 81 x 4 logits sit within rounding of the threshold)
                assert abs(res[d + "_acc1"] - qs[0][d + "_acc_src"]) < 2e-2 and abs(res[d + "_acc2"] - qs[0][d + "_acc_tgt"]) < 2e-2
    # the caller's optimisers hold the state again (what callbacks.py:78-80 would save): 5 steps of Adam, momentum buffers
    st = og.state_dict()["state"]
    assert st and all(int(float(s["step"])) == 5 for s in st.values())
    names = [k for k, _ in gen.named_parameters()]
    assert all(not names[i].startswith("encoder.conv1_1.") for i in st)            # never updated in the reference either
    for o in (o1, o2, o4):
        assert all(s.get("momentum_buffer") is not None for s in o.state_dict()["state"].values())
    # lr decay between epochs (train_mscmrseg.py:585-589) is picked up
    for pg in og.param_groups:
        pg["lr"] *= 0.2
    T.train_epoch(gen, d2, d4, d1, og, o2, o4, o1, _gen(epochs[0][:1], "A"), _gen(epochs[0][:1], "B"))
    assert abs(gen._pcuda_trainer[1].opt_gen.lr - 2e-4) < 1e-12
    # an empty epoch: np.mean of empty lists in the reference
    empty = T.train_epoch(gen, d2, d4, d1, og, o2, o4, o1, iter(()), iter(()))
    assert np.isnan(empty["seg_loss"]) and np.isnan(empty["ver_s_loss"])


def test_valid_model_with_reference_signature(dev):
    from oracle import validate as OV
    from oracle.synth import synth_batch
    import pointcloududa_amd.train_mscmrseg as TM
    cfg_kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    cfg, ps, (gen, _, _, _) = _nets(dev, cfg_kw, 2300)
    TM.args = types.SimpleNamespace(d1=True, d2=True, d4=True, dr=0.01, wp=1.0)
    sets = [[synth_batch(3, 1, 4, 128, seed=2400 + 10 * s + i) for i in range(2)] for s in range(3)]
    res = TM.valid_model(gen, _gen(sets[0], "A"), _gen(sets[1], "A"), _gen(sets[2], "A"))
    assert set(res) == {"val_dice", "val_loss", "valid_vert_loss", "val_lge_dice", "val_lge_loss", "test_lge_dice", "test_lge_loss"}
    assert not gen.training                                              # the reference leaves the model in eval mode
    for key_d, key_l, batches in (("val_dice", "val_loss", sets[0]), ("val_lge_dice", "val_lge_loss", sets[1]),
                                  ("test_lge_dice", "test_lge_loss", sets[2])):
        os_ = [OV.valid_batch(ps[0], bt[0], bt[1], bt[2], cfg) for bt in batches]
        assert abs(res[key_l] - np.mean([o["loss"] for o in os_])) <= 1e-3 * abs(np.mean([o["loss"] for o in os_]))
        assert abs(res[key_d] - np.mean([o["dice"] for o in os_])) <= 1e-4


class _RefGen:
    """the reference generators' iterator protocol (data_generator_mscmrseg.py:270-283): the epoch count is reset AS
    StopIteration is raised, so a consumer that asks once more starts another epoch"""

    def __init__(self, batches, which):
        self.items, self.calls, self._totalcount, self._index = list(_gen(batches, which)), 0, 0, 0

    def __iter__(self):
        self._totalcount = 0
        return self

    def __next__(self):
        self.calls += 1
        if self._totalcount >= len(self.items):
            self._totalcount = 0
            raise StopIteration
        self._totalcount += 1
        it = self.items[self._index % len(self.items)]
        self._index += 1
        return it


def test_epoch_functions_end_with_reference_style_generators(dev):
    """`train_epoch` / `valid_model` must consume ONE epoch of generators that restart when asked again"""
    from oracle.synth import synth_batch
    import pointcloududa_amd.train_mscmrseg as TM
    cfg_kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    cfg, ps, (gen, d1, d2, d4) = _nets(dev, cfg_kw, 2500)
    og = torch.optim.Adam(gen.parameters(), lr=1e-3, betas=(0.9, 0.99))
    mk = lambda m: torch.optim.SGD(m.parameters(), lr=2.5e-5, momentum=0.99, weight_decay=0.0005)
    o1, o2, o4 = mk(d1), mk(d2), mk(d4)
    TM.args = types.SimpleNamespace(d1=True, d2=True, d4=True, dr=0.01, wp=1.0)
    batches = [synth_batch(4, 1, 4, 128, seed=2600 + i) for i in range(5)]
    ga, gb = _RefGen(batches[:3], "A"), _RefGen(batches, "B")            # A runs out first: zip never asks B a fourth time
    TM.train_epoch(gen, d2, d4, d1, og, o2, o4, o1, ga, gb)
    assert int(float(next(iter(og.state_dict()["state"].values()))["step"])) == 3
    assert (ga.calls, gb.calls) == (4, 3)
    import copy
    snap_g, snap_2 = copy.deepcopy(og.state_dict()), copy.deepcopy(o2.state_dict())
    TM.train_epoch(gen, d2, d4, d1, og, o2, o4, o1, ga, gb)              # second epoch: B continues with its batches 3, 4, 0
    assert (ga.calls, gb.calls, gb._index) == (8, 6, 6)
    va, vb, vt = _RefGen(batches[:2], "A"), _RefGen(batches[:3], "A"), _RefGen(batches[:1], "A")
    res = TM.valid_model(gen, va, vb, vt)
    assert (va.calls, vb.calls, vt.calls) == (3, 4, 2) and np.isfinite(res["val_dice"])
    # a rollback between epochs (optimizer.load_state_dict of an earlier checkpoint with the same objects): the next epoch
    # continues from the LOADED state and does not overwrite it with the trainer's own
    assert int(float(next(iter(og.state_dict()["state"].values()))["step"])) == 6
    og.load_state_dict(snap_g)
    o2.load_state_dict(snap_2)
    TM.train_epoch(gen, d2, d4, d1, og, o2, o4, o1, _RefGen(batches[:1], "A"), _RefGen(batches[:1], "B"))
    assert int(float(next(iter(og.state_dict()["state"].values()))["step"])) == 4
