"""Where each stream of the train step is, when -- from timed events recorded at the schedule's joints
(train_step.py, PCUDA_TIMELINE=1), i.e. without a profiler slowing the host down (under rocprofv3 the host falls behind
the GPU and the discriminator updates LOOK as if they ran after the segmenter's backward pass).
usage: PCUDA_TIMELINE=1 python3 scripts/step_timeline.py [--workload full_uda] [--steps 5]"""
import argparse
import os
import sys

os.environ["PCUDA_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="full_uda")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--settle", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda", 0)
wl = B.WORKLOADS[args.workload]
tr = B.build_trainer(wl, dev, seed=0)
batch = B.synth_device_batch(wl["batch"], wl.get("hw", 256), wl.get("n_class", 4), seed=100, dev=dev,
                             in_channels=wl.get("in_channels", 1), gaussian=wl.get("variant") == "mmwhs")
for _ in range(args.settle):
    tr.step(*batch)
torch.cuda.synchronize()
tr._marks = []
for _ in range(args.steps + 1):
    tr.step(*batch)
torch.cuda.synchronize()
marks = tr._marks
starts = [i for i, (l, _) in enumerate(marks) if l == "step"]
for k in range(1, len(starts) - 1):      # (skip the first step of the window: the queue was empty when it started)
    seg = marks[starts[k]:starts[k + 1]]
    t0 = seg[0][1]
    nxt = marks[starts[k + 1]][1]
    print("step %d: %.2f ms to the next step's first mark" % (k, t0.elapsed_time(nxt)))
    for label, ev in sorted(seg[1:], key=lambda m: t0.elapsed_time(m[1])):
        print("   %8.2f ms  %s" % (t0.elapsed_time(ev), label))
