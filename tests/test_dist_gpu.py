"""The N > 1 path of the REAL step on the HIP kernels: two ranks (processes) share the one GPU of the test box and
talk over gloo (RCCL refuses two ranks on one device; the step's collective calls are backend-agnostic
``torch.distributed`` all-reduces / broadcasts, and the one-rank RCCL group of tests/test_step_gpu.py covers the RCCL
stream semantics).  Exercised here, on hardware, with world size 2: parameter broadcast from rank 0, the segmenter's
two-bucket all-reduce started from inside the backward pass, the per-discriminator all-reduces on their side streams,
the 1/N mean folded into the optimiser kernels -- over two optimiser steps.

* same shard on both ranks  -> (g + g) / 2 == g exactly: both ranks must reproduce the single-process trajectory bit
  for bit;
* different shards          -> replicas that started from DIFFERENT seeds end identical on both ranks, and differ from
  the single-shard trajectory."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
KW = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=1)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _build(seed, dev):
    from oracle import nets as ON
    from pointcloududa_amd.networks import PointNetCls, Segmentation_model_Point, UncertaintyDiscriminator
    cfg = ON.SegCfg(**KW)
    load = lambda m, p: (m.load_state_dict({k: v.clone() for k, v in p.items()}), m.to(dev).train())[1]
    return (load(Segmentation_model_Point(**KW), ON.make_params(ON.seg_param_shapes(cfg), seed + 1)),
            load(UncertaintyDiscriminator(4), ON.make_params(ON.disc_param_shapes(4), seed + 2, std=0.02)),
            load(UncertaintyDiscriminator(4), ON.make_params(ON.disc_param_shapes(4), seed + 3, std=0.02)),
            load(PointNetCls(drop=0.0), ON.make_params(ON.pointnet_cls_param_shapes(), seed + 4)))


def _run(nets, shards, dev, group=None):
    from oracle.synth import synth_batch
    from pointcloududa_amd.train_step import AdversarialTrainer, TrainCfg
    tr = AdversarialTrainer(*nets, TrainCfg(n_class=4), process_group=group)
    out = None
    for seed in shards:
        batch = synth_batch(4, 1, 4, 96, seed=seed)
        out = tr.step(*[torch.from_numpy(t).to(dev) for t in batch])
    torch.cuda.synchronize()
    host = AdversarialTrainer.to_host(out, tr.cfg)
    assert all(np.isfinite(v) for v in host.values()), host
    return [o.p.detach().cpu().clone() for o in [tr.opt_gen, tr.opt_d1, tr.opt_d2, tr.opt_d4]]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    # (a) replicas built from different seeds (the trainer's broadcast must make them rank 0's), same shards
    same = _run(_build(0 if rank == 0 else 50, dev), [9, 10], dev)
    # (b) different shards per rank
    diff = _run(_build(0 if rank == 0 else 50, dev), [9 + 100 * rank, 10 + 100 * rank], dev)
    torch.save({"same": same, "diff": diff}, os.path.join(tmp, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_on_one_gpu(dev, tmp_path):
    single = _run(_build(0, dev), [9, 10], dev)            # no process group: the single-process trajectory
    torch.cuda.synchronize()
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"))
    for i, name in enumerate(("segmenter", "d1", "d2", "d4")):
        assert torch.equal(r0["same"][i], r1["same"][i]), name
        assert torch.equal(r0["same"][i], single[i]), name + ": two ranks on one shard != single process"
        assert torch.equal(r0["diff"][i], r1["diff"][i]), name + ": replicas diverged"
        assert not torch.equal(r0["diff"][i], single[i]), name


def _shared_gpu_worker(rank, tmp, iters):
    """two of these run at once: a persistent MFMA convolution in front of every launch of two direct kernels"""
    sys.path.insert(0, ROOT)
    import pointcloududa_amd.kernels as K
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    big, wb, xb, bb = K.ConvOp(32, 32, 3, pad=1), rn(32, 32, 3, 3) * 0.05, rn(4, 32, 256, 256), torch.zeros(32, device=dev)
    cls, wc, xc, bc = K.ConvOp(32, 4, 1), rn(4, 32, 1, 1) * 0.05, rn(4, 32, 256, 256), torch.zeros(4, device=dev)
    d1, wd, dyd = K.ConvOp(4, 64, 4, stride=2, pad=2), rn(64, 4, 4, 4) * 0.05, rn(4, 64, 49, 49)
    victims = {"classifier forward": lambda: cls.forward(xc, wc, bc, 1.0, 256, 256)[0],
               "discriminator first-layer dgrad": lambda: d1.dgrad(dyd, wd, 96, 96)}
    refs = {k: f().clone() for k, f in victims.items()}
    bad = {k: torch.zeros((), dtype=torch.int64, device=dev) for k in victims}
    torch.cuda.synchronize()
    for _ in range(iters):
        big.forward(xb, wb, bb, 1.0, 256, 256)
        for k, f in victims.items():
            bad[k] += (f() != refs[k]).any().long()
    torch.cuda.synchronize()
    torch.save({k: int(v) for k, v in bad.items()}, os.path.join(tmp, "shared%d.pt" % rank))


def test_direct_kernels_return_the_same_bits_on_a_shared_gpu(dev, tmp_path):
    """Regression test of the VMEM address rule (csrc/common.h): before it, the classifier forward and the
    discriminators' first-layer data gradient returned wrong 16-lane groups in 20-70 % of their launches whenever a
    second process ran this library's MFMA convolution on the same GPU (never with one process)."""
    iters = 400
    mp.spawn(_shared_gpu_worker, args=(str(tmp_path), iters), nprocs=2, join=True)
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), "shared%d.pt" % r))
        assert all(v == 0 for v in got.values()), (r, got, iters)
