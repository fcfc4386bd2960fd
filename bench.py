#!/usr/bin/env python3
"""Benchmark of the PointCloudUDA adversarial train step on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: either under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (RANK / LOCAL_RANK /
WORLD_SIZE in the environment), or plainly as ``python bench.py --gpus N``: the parent then starts one child process per
GPU BEFORE anything touches the GPU, relays rank 0's JSON line and exits non-zero if any child fails.

Metric (BASELINE.json): adversarial train-step images/sec (segmenter + 3 discriminators) at 256x256.
One "step" = one iteration of the reference's train_epoch loop (train_mscmrseg.py:183-330): source
batch fwd+bwd, target batch fwd + adversarial bwd, Adam on G, two passes per discriminator, SGD on
the D's.  img/s = bs * steps/s, bs = the reference's -bs (source images per step; an equal number
of target images rides along).  Weak scaling: every rank runs the full per-GPU batch, gradients
are all-reduced over RCCL before the optimiser kernels.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     the dominant kernel family (implicit-GEMM MFMA convolution: forward + dgrad launches)
               timed live with HIP events on the launch stream over extra profiled steps
  cpu_baseline the oracle's CPU restatement of the same step, timed on this host (rank 0, N = 1)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric (seg + 3 discriminators) is quoted on
    "full_uda": dict(desc="MS-CMRSeg full UDA: UNet(PointNet head) + d1 + d2 + d4, 256x256x1, 4 classes",
                     batch=32, d1=True, d2=True, d4=True, gflop_per_pair=282.0),
    # BASELINE.json configs[1]
    "unet_d2": dict(desc="MS-CMRSeg UNet + entropy-map discriminator (d2), 256x256x1, 4 classes",
                    batch=16, d1=False, d2=True, d4=False, gflop_per_pair=243.0),
    # BASELINE.json configs[3], per rank: global batch 128 = 16 / rank x 8 (run with --gpus 8)
    "mmwhs_uda": dict(desc="MM-WHS-like full UDA: 3-channel input, 5 classes, softmax mode, PointNetCls(feature_transform,"
                           " ext), train_mmwhs.py loop",
                      batch=16, d1=True, d2=True, d4=True, gflop_per_pair=286.0, variant="mmwhs", in_channels=3,
                      n_class=5, pn=dict(feature_transform=True, ext=True)),
    # BASELINE.json configs[4] names a 512x512 input on a DeepLab-v3+ backbone; the reference holds no such model (SURVEY
    # section 0).  STAND-IN, labelled as such: the reference's own segmenter at that input size (fc_inch=729:
    # unet.py:169-178) with the three discriminators, per rank 8 of the global batch of 64.  Work per pair: 4x the
    # 256x256 figure for everything except the 6x6 valid head convolution (27x27 instead of 11x11 outputs).
    "uda_512": dict(desc="STAND-IN for config 5 (no DeepLab in the reference): UNet(PointNet head, fc_inch=729) + d1 + d2 + d4, "
                         "512x512x1, 4 classes",
                    batch=8, d1=True, d2=True, d4=True, gflop_per_pair=1166.0, hw=512, fc_inch=729),
    # The reference's REAL MS-CMRSeg operating point (train_mscmrseg.py:412-425): crop 224, 3-channel PNG slices,
    # Segmentation_model_Point(filters=32, pointnet=True) with its constructor defaults in_channels=3, fc_inch=81; the
    # image discriminators end in [B,1,8,8].  Not a BASELINE.json config (those are 256x256); the shape a user of the
    # reference runs.  Work per pair: 6 x 28.28 + 2 x 8 x 2.78 + 8 x 0.17 GFLOP (SURVEY 8a5 / 8d convention).
    "mscmrseg_224": dict(desc="MS-CMRSeg as the reference runs it: UNet(PointNet head, fc_inch=81) + d1 + d2 + d4, 224x224x3, 4 classes",
                         batch=32, d1=True, d2=True, d4=True, gflop_per_pair=215.6, hw=224, fc_inch=81, in_channels=3,
                         d_gflop=2.783),
}
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def synth_device_batch(b, hw, n_class, seed, dev, in_channels=1, gaussian=False):
    """device-resident synthetic batch in the reference's layout: images U[0,1),
    one-hot uint8 nested-ellipse masks, vertices = HIP sampler(mask) / 255."""
    from pointcloududa_amd.utils.npy2point import masks_to_pointclouds
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:hw, 0:hw].astype(np.float64)

    def labels():
        lab = np.zeros((b, hw, hw), dtype=np.int64)
        for i in range(b):
            cy = hw * (0.5 + 0.08 * (rng.random() - 0.5))
            cx = hw * (0.5 + 0.08 * (rng.random() - 0.5))
            for k in range(1, n_class):
                ry = hw * 0.36 * (n_class - k) / (n_class - 1)
                rx = hw * 0.28 * (n_class - k) / (n_class - 1)
                lab[i][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = k
        return lab

    g = torch.Generator(device="cpu").manual_seed(seed)
    draw = torch.randn if gaussian else torch.rand           # MM-WHS slices are z-scored, MS-CMRSeg uint8/255
    img_a = draw((b, in_channels, hw, hw), generator=g).to(dev)
    img_b = draw((b, in_channels, hw, hw), generator=g).to(dev)
    lab_a, lab_b = labels(), labels()
    onehot = torch.from_numpy(np.ascontiguousarray(np.moveaxis(np.eye(n_class, dtype=np.uint8)[lab_a], -1, 1))).to(dev)
    firsts = torch.from_numpy(rng.integers(0, 1 << 30, size=(2, b)).astype(np.int32)).to(dev)
    va = masks_to_pointclouds(torch.from_numpy((lab_a > 0).astype(np.uint8)).to(dev), firsts[0]).float() / 255.0
    vb = masks_to_pointclouds(torch.from_numpy((lab_b > 0).astype(np.uint8)).to(dev), firsts[1]).float() / 255.0
    return img_a, onehot, va.contiguous(), img_b, vb.contiguous()


def build_trainer(wl, dev, seed, group=None):
    from pointcloududa_amd.networks import PointNetCls, Segmentation_model_Point, UncertaintyDiscriminator
    from pointcloududa_amd.train_step import AdversarialTrainer, TrainCfg
    torch.manual_seed(seed)                       # train_mscmrseg.py:668-670 seeds torch + numpy with 0
    np.random.seed(seed)
    nc, cin, variant = wl.get("n_class", 4), wl.get("in_channels", 1), wl.get("variant", "mscmrseg")
    gen = Segmentation_model_Point(filters=32, in_channels=cin, n_class=nc, pointnet=wl["d4"],
                                   fc_inch=wl.get("fc_inch", 121)).to(dev)
    d1 = UncertaintyDiscriminator(in_channel=nc).to(dev) if wl["d1"] else None
    d2 = UncertaintyDiscriminator(in_channel=nc).to(dev) if wl["d2"] else None
    d4 = PointNetCls(**wl.get("pn", {})).to(dev) if wl["d4"] else None          # dropout p = 0.3 as in the reference
    cfg = TrainCfg(variant=variant, d1=wl["d1"], d2=wl["d2"], d4=wl["d4"], n_class=nc,
                   d_momentum=0.95 if variant == "mmwhs" else 0.99)     # train_mmwhs.py:856-859 / train_mscmrseg.py:437
    tr = AdversarialTrainer(gen, d1, d2, d4, cfg, process_group=group)
    tr.train()
    return tr


def cpu_baseline(wl, budget_s=20.0, batches=(2, 16)):
    """The oracle (CPU restatement of the reference step) on this host's cores, on a bounded sample (BASELINE.md
    section 3: B = 2 and B = 16).  B = 2: one warm-up step, then as many steps as fit the budget (>= 1).  B = 16: ONE
    step (its first, no warm-up) and only if the B = 2 rate predicts it inside ~2x the budget; reported beside the
    B = 2 figure, which stays `value`."""
    from oracle import nets as ON
    from oracle.step import OracleTrainer, StepCfg
    from oracle.synth import synth_batch
    cores = min(os.cpu_count() or 1, 16)       # oneDNN convs at batch 2 do not scale past a socket's worth of threads
    torch.set_num_threads(cores)
    nc, cin, variant = wl.get("n_class", 4), wl.get("in_channels", 1), wl.get("variant", "mscmrseg")
    pn = wl.get("pn", {})
    hw = wl.get("hw", 256)
    cfg = ON.SegCfg(filters=32, in_channels=cin, n_class=nc, pointnet=wl["d4"], fc_inch=wl.get("fc_inch", 121))
    scfg = StepCfg(variant=variant, d1=wl["d1"], d2=wl["d2"], d4=wl["d4"], n_class=nc,
                   d_momentum=0.95 if variant == "mmwhs" else 0.99,
                   pn_feature_transform=pn.get("feature_transform", False), pn_ext=pn.get("ext", False))

    def fresh():
        pg = ON.make_params(ON.seg_param_shapes(cfg), 1)
        p1 = ON.make_params(ON.disc_param_shapes(nc), 2, std=0.02) if wl["d1"] else None
        p2 = ON.make_params(ON.disc_param_shapes(nc), 3, std=0.02) if wl["d2"] else None
        p4 = ON.make_params(ON.pointnet_cls_param_shapes(**pn), 4) if wl["d4"] else None
        return OracleTrainer(cfg, scfg, pg, p1, p2, p4)

    orc = fresh()
    b = batches[0]
    batch = synth_batch(b, cin, nc, hw, seed=5, gaussian=variant == "mmwhs")
    t0 = time.perf_counter()
    orc.step(*batch)                                           # warm-up (also the fallback sample)
    warm = time.perf_counter() - t0
    t0, n = time.perf_counter(), 0
    while warm < budget_s / 2 and n < 8:
        orc.step(*batch)
        n += 1
        if time.perf_counter() - t0 + warm > budget_s:
            break
    dt = time.perf_counter() - t0
    if n == 0:
        n, dt = 1, warm
    res = {"value": round(b * n / dt, 3), "unit": "img/s", "cores": cores, "kind": "port",
           "sample": "%d step(s) of the same workload at batch %d (fp32, torch CPU oracle, %d threads)" % (n, b, cores)}
    for b2 in batches[1:]:
        est = (dt / n) * b2 / b * 1.3
        if est > 2.0 * budget_s:
            res["batch%d" % b2] = {"value": None, "skipped": "predicted %.0f s for one step on this host" % est}
            continue
        orc2 = fresh()
        batch2 = synth_batch(b2, cin, nc, hw, seed=5, gaussian=variant == "mmwhs")
        t0 = time.perf_counter()
        orc2.step(*batch2)
        d2 = time.perf_counter() - t0
        res["batch%d" % b2] = {"value": round(b2 / d2, 3), "unit": "img/s", "sample": "1 step at batch %d, no warm-up" % b2}
    return res


def pmc_traffic_per_launch():
    """HBM bytes per launch of the forward+dgrad conv family from the committed rocprofv3 PMC summary of this same
    command (profiles/r*_pmc_traffic.csv, written by scripts/gpu_pmc_step.sh: one --pmc pass per counter, FETCH_SIZE
    and WRITE_SIZE in KB).  bench.py cannot run under the profiler and time itself at once, so this number is read
    back rather than collected live; FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide loads at 64 B:
    MI355X_MICROARCH.md, HBM section -- exact for the float4 staging loads, uncalibrated for the dword path)."""
    import csv
    import glob
    from pointcloududa_amd._lib import csrc_hash
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.csv")))
    if not files:
        return None, "no profiles/r*_pmc_traffic.csv"
    lines = open(files[-1]).read().split("\n")
    # first line: "# csrc_sha256=<hash>" of the kernel sources the counters were collected on (scripts/gpu_pmc_step.sh)
    have = lines[0].split("=", 1)[1].strip() if lines and lines[0].startswith("# csrc_sha256=") else None
    if have != csrc_hash():
        return None, ("%s was collected on kernel sources %s, this tree is %s: not quoted (re-run scripts/gpu_pmc_step.sh)"
                      % (os.path.basename(files[-1]), have, csrc_hash()))
    n = fetch = write = 0.0
    for r in csv.DictReader([l for l in lines if l and not l.startswith("#")]):
        if r["kernel"].startswith(("igemm_pipe_kernel", "igemm8_kernel", "igemm_kernel", "conv3ap_kernel", "conv3rs_kernel")):
            n += float(r["launches"]); fetch += float(r["FETCH_SIZE_KB_sum_raw"]); write += float(r["WRITE_SIZE_KB_sum"])
    if n == 0:
        return None, "no forward / dgrad launches in " + os.path.basename(files[-1])
    return int((2.0 * fetch + write) * 1024.0 / n), "profiles/%s (FETCH_SIZE x2 + WRITE_SIZE; same kernel sources: %s)" % (
        os.path.basename(files[-1]), have)


def spawn_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher: one child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set), started before this process has made any GPU call (a process that has initialised HIP must not be
    replaced or forked into ranks).  The children's stderr passes through; rank 0's stdout is relayed line by line so
    that its JSON line is this process's last line.  Exit code: the first non-zero child status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   PCUDA_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is read on a thread; the parent polls EVERY child: the first one that exits non-zero ends the
    # run at once (the others would sit in an RCCL collective until its watchdog fires) -- the children we started are
    # terminated by their exact PIDs, nothing is re-executed.
    import threading
    box = {"last": None}

    def relay():
        for line in procs[0].stdout:
            line = line.decode("utf-8", "replace").rstrip("\n")
            if line.startswith("{") and '"metric"' in line:
                box["last"] = line
            else:
                print(line, file=sys.stderr)
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc = 0
    while True:
        codes = [pr.poll() for pr in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    if rc != 0:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        t_end = time.time() + 10.0
        for pr in procs:
            try:
                pr.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                pr.kill()
        raise SystemExit(rc if rc > 0 else 1)
    th.join(timeout=10.0)
    if box["last"] is None:
        raise SystemExit(1)
    print(box["last"], flush=True)


def pin_rank_to_cores(local_rank, local_world):
    """One process per GPU issues ~900 launches per step from Python: N ranks on one host must not migrate over each
    other's cores.  Rank r of N takes the r-th contiguous slice of the cores this process may run on (cores / N each),
    BEFORE its first GPU call, so the HIP runtime's helper threads inherit the mask.  PCUDA_NO_AFFINITY=1 leaves the
    scheduler alone.  Returns the slice (tests)."""
    if local_world <= 1 or os.environ.get("PCUDA_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    cores = sorted(os.sched_getaffinity(0))
    per = len(cores) // local_world
    if per < 1:
        return None
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(per, 4)))       # the host side of a step is one Python thread + small torch ops
    return mine


def dry_run(args, world, rank):
    """``--dry-run``: the launcher and the collective plumbing without a GPU (CPU tests): a gloo group of the spawned
    ranks, the same barrier + max-over-ranks timing, a surrogate step (an all-reduced vector), ONE JSON line."""
    import torch.distributed as dist
    if os.environ.get("PCUDA_DRYRUN_FAIL_RANK") == str(rank):      # (tests: a rank that dies during start-up)
        raise SystemExit(3)
    cores = pin_rank_to_cores(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    g = torch.full((1024,), float(rank + 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            dist.all_reduce(g)
            g /= world
    dt = time.perf_counter() - t0
    ranks = world
    if world > 1:
        t = torch.tensor([dt, 1.0], dtype=torch.float64)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        dt, ranks = float(t[0]), int(t[1])
        # every rank's core slice, gathered: the slices must be disjoint
        mine = torch.full((64,), -1, dtype=torch.int64)
        if cores:
            mine[:min(64, len(cores))] = torch.tensor(cores[:64])
        allc = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        core_sets = [sorted(int(c) for c in t_ if c >= 0) for t_ in allc]
        dist.barrier()
        dist.destroy_process_group()
    else:
        core_sets = [cores or []]
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no kernels)", "value": round(ranks * args.steps / max(dt, 1e-9), 2), "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle": args.settle,
                          "config": {"workload": "dry-run", "ranks_in_group": ranks, "backend": "gloo",
                                     "rank_cores": core_sets}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--settle", type=int, default=40, help="untimed setup steps in front of the warm-up steps")
    ap.add_argument("--workload", default="full_uda", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--precision", default=os.environ.get("PCUDA_PRECISION", "bf16x3"), choices=["bf16x3", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="launcher + collectives only, on CPU over gloo (tests)")
    ap.add_argument("--cpu-batches", default="2,16", help="batch sizes of the cpu_baseline leg")
    args = ap.parse_args()

    # no launcher around us and more than one GPU asked for: become the launcher.  Nothing above this line (and
    # nothing at import time) initialises HIP: torch.cuda.is_available() is first called in the children.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        dry_run(args, world, rank)
        return
    pin_rank_to_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))    # before the first GPU call
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # PCUDA_SHARE_GPU=1 (rehearsal of the N > 1 control flow on a one-GPU box, with PCUDA_DIST_BACKEND=gloo: RCCL refuses two
    # ranks per device): every rank uses device 0.  Not a measurement.
    share = os.environ.get("PCUDA_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("PCUDA_FORCE_COLLECTIVES") == "1":   # (forced: exercise the RCCL path on one GPU)
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
        backend = os.environ.get("PCUDA_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher's WORLD_SIZE is %d" % (args.gpus, world))

    import pointcloududa_amd as P
    from pointcloududa_amd import kernels as K
    P.set_precision(args.precision)
    wl = WORKLOADS[args.workload]
    b = args.batch or wl["batch"]
    tr = build_trainer(wl, dev, seed=0, group=None)     # (the trainer broadcasts rank 0's parameters and buffers)
    batch = synth_device_batch(b, wl.get("hw", 256), wl.get("n_class", 4), seed=100 + rank, dev=dev, in_channels=wl.get("in_channels", 1),
                               gaussian=wl.get("variant") == "mmwhs")

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # PCUDA_GRAPH=1 (single process): replay the step from a captured hipGraph.  Off by default: measured 68.7 vs
    # 69.4 ms/step -- the ~4 ms between kernels is dependent-launch latency on the GPU, not host launch time.
    # Default: eager where the rank has >= 4 host cores (22 ms of host issue time under a 44-ms GPU step), hipGraph replay
    # (10 ms, the tested path of tests/test_step_gpu.py) where it has fewer -- eight ranks on a 16-core host.
    try:
        cores_here = len(os.sched_getaffinity(0))
    except AttributeError:
        cores_here = os.cpu_count() or 1
    graph_env = os.environ.get("PCUDA_GRAPH")
    use_graph = graph_env == "1" if graph_env is not None else cores_here < 4
    graph_why = ("PCUDA_GRAPH=%s" % graph_env) if graph_env is not None else ("%d host cores for this rank" % cores_here)
    step = tr.step_graphed if use_graph else tr.step
    # settle phase (untimed setup, not part of --warmup): lazily created buffers, packed-weight caches, allocator pools
    # of the side streams and the GPU's clocks reach their steady state only after a second or two of work
    for _ in range(args.settle):
        step(*batch)
    for _ in range(args.warmup):
        step(*batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(*batch)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0         # this rank's own work done (before the closing barrier)
    sync()
    dt = time.perf_counter() - t0
    ranks_in_group = 1
    per_rank = {"min": round(b * args.steps / dt, 2), "max": round(b * args.steps / dt, 2)}
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tmin = torch.tensor([dt_own], dtype=torch.float64, device=dev)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        # (a straggler rank shows as min << max: the slowest rank sets `value`; the fastest one's own completion time, taken
        # before the closing barrier, says how long it waited there.  The per-step all-reduces couple the ranks, so a
        # spread here is the last step's skew plus whatever a rank loses outside the collectives)
        per_rank = {"min": round(b * args.steps / dt, 2), "max": round(b * args.steps / float(tmin.item()), 2)}
        cnt = torch.ones(1, dtype=torch.float64, device=dev)      # what RCCL itself says about the group
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        ranks_in_group = int(cnt.item())
    host = tr.to_host(out, tr.cfg)
    if not all(np.isfinite(v) for v in host.values()):
        raise SystemExit("non-finite loss in the benchmark step: %r" % host)

    # host side of a step: time to ISSUE it (Python + launches, nothing waited for) and the library's launch count.  With
    # N ranks on one host this, not the GPU, is what has to stay below ms_per_step.
    n_issue = max(4, min(10, args.steps))
    K.launch_count(reset=True)
    t0 = time.perf_counter()
    for _ in range(n_issue):
        step(*batch)
    host_issue_ms = 1000.0 * (time.perf_counter() - t0) / n_issue
    launches_per_step = K.launch_count(reset=True) / n_issue
    sync()
    # exposed communication: the same steps with the gradient all-reduces switched off on every rank
    comm_exposed_ms = None
    if dist is not None and not use_graph:
        from pointcloududa_amd import optim as O
        n_cmp = max(5, min(30, args.steps))

        def timed(n):
            sync()
            t1 = time.perf_counter()
            for _ in range(n):
                step(*batch)
            sync()
            tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return 1000.0 * float(tt.item()) / n
        with_c = timed(n_cmp)
        O.set_collectives(False)
        try:
            for _ in range(3):
                step(*batch)
            without_c = timed(n_cmp)
        finally:
            O.set_collectives(None)
        tr.broadcast_parameters()          # the replicas drifted apart while nothing was all-reduced
        comm_exposed_ms = round(with_c - without_c, 3)

    result = {
        "metric": "adversarial train-step images/sec (seg+3 discr) at %dx%d" % (wl.get("hw", 256), wl.get("hw", 256)),
        "value": round(b * world * args.steps / dt, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "settle": args.settle, "ms_per_step": round(1000.0 * dt / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "host_issue_ms_per_step": round(host_issue_ms, 3), "launches_per_step": round(launches_per_step, 1),
        "comm_exposed_ms": comm_exposed_ms,
        "per_rank_img_s": per_rank,
        # shader clock while every SIMD issues MFMAs back to back, measured AFTER the timed region on this box (rank 0's GPU):
        # boxes of the pool differ by up to 10 % on one binary; value / clock normalises lines of different rounds
        "clock_ghz_under_load": round(K.clock_ghz_under_load(dev), 3),
        # launches that ran on a generic fallback kernel (csrc/variants.h: pruned default build): must be 0 for a benchmark line
        "fallback_launches": int(K.L.lib().pcuda_fallback_count()),
        "dtype": "bf16x3 MFMA (split-bf16, fp32 accumulate; fp32 storage)" if args.precision == "bf16x3"
                 else "bf16 MFMA (fp32 accumulate; fp32 storage)",
        "data": "synthetic",
        "config": {"workload": wl["desc"], "per_gpu_batch": b, "global_batch": b * world, "precision": args.precision,
                   "parallelism": "dp%d" % world, "ranks_in_group": ranks_in_group,
                   "collective": ("%s all-reduce of the flat gradient buffers (G: 2 buckets, D: 1 each)" % (
                       "RCCL" if os.environ.get("PCUDA_DIST_BACKEND", "nccl") == "nccl" else os.environ["PCUDA_DIST_BACKEND"] + " (REHEARSAL, ranks share one GPU)"))
                                 if world > 1 else ("RCCL all-reduce in a one-rank group (PCUDA_FORCE_COLLECTIVES=1: the N > 1 code path on one GPU)"
                                                    if dist is not None else "none"),
                   "algorithmic_gflop_per_pair": wl["gflop_per_pair"],
                   # the SURVEY 8d convention counts 8 passes per discriminator; with the target forward of d1 / d2
                   # replayed from the adversarial pass, 7 of them execute for those two networks
                   "executed_gflop_per_pair": round(wl["gflop_per_pair"] - (2 * wl.get("d_gflop", 3.60 * (wl.get("hw", 256) / 256.0) ** 2)
                                                    * (int(wl["d1"]) + int(wl["d2"])) / 2.0 if tr.d_reuse else 0.0), 1),
                   "box_to_box": "742-779 img/s measured for the default command on the final kernels across the MI355X boxes of round 6 (probe clocks 2.06-2.37 GHz; profiles/r06_box_scatter.txt); compare lines by img_s_per_ghz",
                   "streams": "discriminators concurrent" if tr.d_streams else "single",
                   **({"experiment": "PCUDA_EXP_SKIP_DUPDATE=1: discriminator update passes left out -- NOT a benchmark line"}
                      if getattr(tr, "_exp_skip_dupdate", False) else {}),
                   # d1 / d2 see the target batch twice per step with the same weights and the same input values
                   # (adversarial pass, then their own update): the second forward is replayed from the first's
                   # activations.  PCUDA_DREUSE=0 runs it again (same bits, ~1.5 ms per step more).
                   "d_target_forward": "replayed from the adversarial pass of the same step" if tr.d_reuse else "run twice",
                   "launch": ("hipGraph replay" if (use_graph and getattr(tr, "_graph", None) is not None) else "eager") + " (" + graph_why + ")",
                   "losses": {k: round(host[k], 5) for k in ("seg_loss", "adv_loss") if k in host}},
    }

    # value / probe clock: what normalises lines from boxes of different clocks (the probe is an all-SIMD MFMA loop on trivial
    # operands: the clock the box GRANTS, not the lower one the convolution kernels hold on real data --
    # profiles/r06_experiment_power_cap.txt)
    if result["clock_ghz_under_load"]:
        result["img_s_per_ghz"] = round(result["value"] / world / result["clock_ghz_under_load"], 1)

    if not args.no_roofline:
        # every rank runs the profiled steps (they hold collectives when world > 1); rank 0 reports its own kernels
        K.prof_reset()
        K.prof_enable(rank == 0)
        nprof = 2
        # per-launch durations are taken with the discriminator streams serialised (PCUDA_DSTREAMS=0 behaviour): kernels
        # that share the CUs with another stream's kernels would each be billed the shared time
        streams_on, tr.d_streams = tr.d_streams, False
        for _ in range(nprof):
            tr.step(*batch)
        torch.cuda.synchronize()
        tr.d_streams = streams_on
        K.prof_enable(False)
        if os.environ.get("PCUDA_PROF_DUMP"):
            K.prof_dump(os.environ["PCUDA_PROF_DUMP"])
        rs_rows = []
        if rank == 0:   # the row-streaming kernel's launches (HBM-bound members of the family): by tag, from the per-launch dump
            import csv, re, tempfile
            with tempfile.TemporaryDirectory() as td:
                K.prof_dump(os.path.join(td, "layers.csv"))
                for r in csv.DictReader(open(os.path.join(td, "layers.csv"))):
                    m = re.match(r"conv3rs n(\d+) red32 rows32 (\d+)x(\d+) taps9 stats(\d)", r["tag"])
                    if m:   # algorithmic bytes: 32 fp32 planes in, 32 out, 32 more of the saved activation with the BatchNorm-backward reduce
                        n_, h_, w_, st_ = (int(v) for v in m.groups())
                        rs_rows.append((float(r["ms"]), (3 if st_ == 2 else 2) * 32.0 * n_ * h_ * w_ * 4))
        ms, flops, launches = K.prof_read(0)
        wms, wflops, wl_n = K.prof_read(1)
        pms, pbytes, pl_n = K.prof_read(2)
        dms, dflops, dl_n = K.prof_read(3)
        K.prof_reset()
    if rank == 0 and not args.no_roofline:
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # the committed PMC passes were taken on the default command (full_uda, batch 32, bf16x3): only that run quotes them
        same_cmd = args.workload == "full_uda" and b == wl["batch"] and args.precision == "bf16x3" and world == 1
        traffic, traffic_src = pmc_traffic_per_launch() if same_cmd else (None, None)
        result["roofline"] = {
            "kernel": "conv3ap_kernel / conv3rs_kernel / igemm_pipe_kernel / igemm8_kernel (implicit-GEMM MFMA conv: forward + dgrad launches)", "bound": "mfma",
            "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
            "traffic": traffic, "traffic_unit": "bytes per launch (HBM, PMC)", "traffic_source": traffic_src,
            "launches_per_step": launches // nprof, "avg_launch_us": round(1000.0 * ms / max(launches, 1), 2),
            "ms_per_step": round(ms / nprof, 3),
            "mfma_flops_per_algorithmic_flop": 3 if args.precision == "bf16x3" else 1,
            # round 6 (profiles/r06_experiment_power_cap.txt, scripts/micro/mfma_power.hip): what the matrix pipe sustains on REAL
            # operands (normal values as bf16 hi | lo, fragments read from LDS as in the tap loop; the chip holds ~1.8 GHz there
            # against 2.39 on zeros): the ceiling of the MFMA stream alone, before staging and epilogue
            "sustained_peak_real_operands": 1516.0,
            "frac_of_sustained": round(ach * (3 if args.precision == "bf16x3" else 1) / 1516.0, 4),
            "other": {
                "wgrad_kernel": {"achieved_tflops": round(wflops / (wms * 1e-3) / 1e12, 2) if wms > 0 else 0.0,
                                 "ms_per_step": round(wms / nprof, 3), "launches_per_step": wl_n // nprof},
                "conv1d_f32": {"what": "PointNetCls k=1 Conv1d layers, exact fp32 MFMA (peak 157 TFLOP/s)",
                               "achieved_tflops": round(dflops / (dms * 1e-3) / 1e12, 2) if dms > 0 else 0.0,
                               "ms_per_step": round(dms / nprof, 3), "launches_per_step": dl_n // nprof},
                "conv3rs_kernel": {"what": "the 32 -> 32-channel 3x3 launches of the family above (row streaming): HBM-bound, algorithmic bytes = the fp32 planes read + written",
                                   "bound": "hbm", "achieved_gbps": round(sum(b_ for _, b_ in rs_rows) / (sum(m_ for m_, _ in rs_rows) * 1e-3) / 1e9, 1) if rs_rows else 0.0,
                                   "peak_gbps": 8000.0, "ms_per_step": round(sum(m_ for m_, _ in rs_rows) / nprof, 3),
                                   "launches_per_step": len(rs_rows) // nprof},
                "pointwise": {"achieved_gbps": round(pbytes / (pms * 1e-3) / 1e9, 1) if pms > 0 else 0.0,
                              "ms_per_step": round(pms / nprof, 3), "launches_per_step": pl_n // nprof,
                              "peak_gbps": 8000.0},
            },
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(wl, batches=tuple(int(v) for v in args.cpu_batches.split(",")))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL's version banner sits in the C library's stdout buffer until exit: flush it first, so that the JSON
        # line is the last line of output
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
