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
