"""CPU tests of the N > 1 path with world_size 2 over gloo: the flat-gradient all-reduce of
pointcloududa_amd.optim (the only collective of the step) and the rank sharding of bench.py."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointcloududa_amd.networks import UncertaintyDiscriminator
    from pointcloududa_amd.optim import FusedSGD, flatten_module
    torch.manual_seed(0)                                   # identical replicas, as after a broadcast
    m = UncertaintyDiscriminator(in_channel=4)
    opt = FusedSGD(m, lr=0.1)
    flat, grad = flatten_module(m)
    assert flat.numel() >= 2764800 and grad.data_ptr() == opt.g.data_ptr()
    assert all(p.grad is not None and p.grad.data_ptr() >= grad.data_ptr() for p in m.parameters())
    # rank-dependent "gradients" written through the per-parameter views
    for i, p in enumerate(m.parameters()):
        p.grad.fill_(float(rank + 1) * (i + 1))
    scale = opt.all_reduce_grads()
    assert scale == 0.5
    for i, p in enumerate(m.parameters()):
        assert torch.allclose(p.grad, torch.full_like(p.grad, 3.0 * (i + 1)))      # 1x + 2x summed over ranks
    opt.zero_grad()
    assert float(grad.abs().max()) == 0.0
    # the overlapped form used for the segmenter: start, do unrelated work, finish, then step
    for i, p in enumerate(m.parameters()):
        p.grad.fill_(float(rank + 1) * (i + 2))
    work, scale2 = opt.all_reduce_grads_async()
    assert work is not None and scale2 == 0.5
    unrelated = torch.ones(1000).sum()
    opt.finish_all_reduce(work)
    for i, p in enumerate(m.parameters()):
        assert torch.allclose(p.grad, torch.full_like(p.grad, 3.0 * (i + 2)))
    assert float(unrelated) == 1000.0
    opt.zero_grad()
    # the bucketed form of the segmenter's all-reduce: everything behind the head ("conv1." here) goes out first
    # (from inside the backward pass), the head afterwards; together they cover the buffer exactly once
    split = opt.split_after("conv1.")
    first = next(m.parameters())
    assert split >= first.numel() and split < grad.numel() and opt.split_after("conv3.") == 0     # not at the head
    for i, p in enumerate(m.parameters()):
        p.grad.fill_(float(rank + 1) * (i + 3))
    w_tail, _ = opt.all_reduce_grads_async(lo=split)
    w_head, _ = opt.all_reduce_grads_async(lo=0, hi=split)
    opt.finish_all_reduce([w_tail, None, w_head])
    for i, p in enumerate(m.parameters()):
        assert torch.allclose(p.grad, torch.full_like(p.grad, 3.0 * (i + 3)))
    opt.zero_grad()
    ret[rank] = float(flat.sum())
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2_gloo():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == 2 and ret[0] == ret[1]


def _worker_trainer(rank, world, port, ret):
    """Rank-sharded step surrogate: replicas built from DIFFERENT seeds, the trainer's broadcast makes them equal;
    different per-rank gradients (as different batch shards produce) go through the trainer's own all-reduce calls
    (segmenter in two buckets, discriminator in one) and a plain-torch restatement of the fused optimiser kernels
    (they need a GPU); every rank must end with the same parameters, equal to the update by the MEAN gradient."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointcloududa_amd.networks import Segmentation_model_Point, UncertaintyDiscriminator
    from pointcloududa_amd.train_step import AdversarialTrainer, TrainCfg
    torch.manual_seed(100 + rank)                          # replicas start DIFFERENT
    gen = Segmentation_model_Point(filters=4, in_channels=1, n_class=4, pointnet=False)
    d2 = UncertaintyDiscriminator(in_channel=4)
    for m in (gen, d2):                                    # running statistics differ too
        for k, b in m.named_buffers():
            if b.dtype.is_floating_point:
                b.add_(0.01 * (rank + 1))
    before = float(sum(p.double().sum() for p in gen.parameters()))
    tr = AdversarialTrainer(gen, None, d2, None, TrainCfg(d1=False, d2=True, d4=False))
    after = [float(tr.opt_gen.p.double().sum()), float(tr.opt_d2.p.double().sum()),
             float(sum(b.double().sum() for b in gen.buffers()))]
    gathered = [None] * world
    dist.all_gather_object(gathered, (before, after))
    assert gathered[0][0] != gathered[1][0]                        # the seeds did differ
    assert gathered[0][1] == gathered[1][1], gathered              # ... and the broadcast removed the difference
    # per-rank "shard" gradients
    g_gen = torch.Generator().manual_seed(7 + rank)
    tr.opt_gen.g.copy_(torch.randn(tr.opt_gen.g.shape, generator=g_gen))
    tr.opt_d2.g.copy_(torch.randn(tr.opt_d2.g.shape, generator=g_gen))
    mine_gen, mine_d2 = tr.opt_gen.g.clone(), tr.opt_d2.g.clone()
    split = tr.opt_gen.split_after("encoder.")
    assert 0 < split < tr.opt_gen.g.numel()
    w_tail, sc = tr.opt_gen.all_reduce_grads_async(tr.group, lo=split)
    w_head, _ = tr.opt_gen.all_reduce_grads_async(tr.group, lo=0, hi=split)
    w_d, sc_d = tr.opt_d2.all_reduce_grads_async(tr.group)
    tr.opt_gen.finish_all_reduce([w_tail, w_head]); tr.opt_d2.finish_all_reduce(w_d)
    assert sc == sc_d == 1.0 / world
    both = [None] * world
    dist.all_gather_object(both, (mine_gen, mine_d2))
    mean_gen = sum(b[0] for b in both) / world
    assert torch.allclose(tr.opt_gen.g * sc, mean_gen, atol=1e-6)
    p0 = tr.opt_gen.p.clone()
    g = tr.opt_gen.g * sc                                   # Adam's first step (optim.hip adam_kernel, restated)
    tr.opt_gen.p.sub_(1e-3 * g / (g.abs() + 1e-8))
    tr.opt_d2.p.sub_(2.5e-5 * (tr.opt_d2.g * sc_d + 0.0005 * tr.opt_d2.p))
    ends = [None] * world
    dist.all_gather_object(ends, (tr.opt_gen.p.clone(), tr.opt_d2.p.clone()))
    assert torch.equal(ends[0][0], ends[1][0]) and torch.equal(ends[0][1], ends[1][1])
    assert not torch.equal(p0, tr.opt_gen.p)
    ret[rank] = float(tr.opt_gen.p.double().sum())
    dist.barrier()
    dist.destroy_process_group()


def test_rank_sharded_step_surrogate_world2_gloo():
    world, port = 2, _free_port()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker_trainer, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == 2 and ret[0] == ret[1]


def test_bench_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks itself (before any GPU
    call), relays rank 0's JSON line as its last line of stdout and reports the group size the collective saw."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["ranks_in_group"] == 2 and res["steps"] == 3 and "settle" in res
    # every rank pinned itself to its own slice of the cores before its first GPU call (bench.pin_rank_to_cores)
    cores = res["config"]["rank_cores"]
    if len(os.sched_getaffinity(0)) >= 2:
        assert len(cores) == 2 and all(cores) and not (set(cores[0]) & set(cores[1])), cores
        assert set(cores[0]) | set(cores[1]) <= set(os.sched_getaffinity(0))
    # a failing rank makes the launcher exit non-zero (no JSON line)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--dry-run",
                        "--workload", "nope"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and '"metric"' not in r.stdout
    # a rank >= 1 that dies during start-up: rank 0 is then stuck in the rendezvous / a collective; the launcher polls
    # every child, ends the survivors and exits with the failed rank's status instead of waiting for rank 0's stdout
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--dry-run"],
                       capture_output=True, text=True, timeout=120, env=dict(env, PCUDA_DRYRUN_FAIL_RANK="1"))
    assert r.returncode == 3 and '"metric"' not in r.stdout and time.time() - t0 < 60, (r.returncode, time.time() - t0)


def test_single_process_allreduce_is_identity():
    from pointcloududa_amd.networks import UncertaintyDiscriminator
    from pointcloududa_amd.optim import FusedAdam
    opt = FusedAdam(UncertaintyDiscriminator(in_channel=2))
    assert opt.all_reduce_grads() == 1.0
    assert opt.all_reduce_grads_async() == (None, 1.0)
    flat_before = opt.p.clone()
    for p in opt.module.parameters():              # parameters are views of the flat buffer
        assert p.data_ptr() >= opt.p.data_ptr()
    assert torch.equal(flat_before, opt.p)
