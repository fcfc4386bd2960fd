"""how long does the caller's stream wait for the discriminator streams at the end of a step?  (HIP events on the
caller's stream around the join; no profiler in the way)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pointcloududa_amd as P
dev = torch.device("cuda", 0)
P.set_precision("bf16x3")
wl = bench.WORKLOADS["full_uda"]
tr = bench.build_trainer(wl, dev, seed=0)
batch = bench.synth_device_batch(wl["batch"], 256, 4, seed=100, dev=dev)
marks = []
real_wait = torch.cuda.Stream.wait_stream


def timed_wait(self, other):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(self); real_wait(self, other); b.record(self)
    marks.append((a, b, self == torch.cuda.default_stream()))


for _ in range(5):
    tr.step(*batch)
torch.cuda.synchronize()
torch.cuda.Stream.wait_stream = timed_wait
n = 10
for _ in range(n):
    tr.step(*batch)
torch.cuda.synchronize()
torch.cuda.Stream.wait_stream = real_wait
tot_main = sum(a.elapsed_time(b) for a, b, m in marks if m)
print("waits of the caller's stream for side streams: %.2f ms/step over %d joins/step" % (tot_main / n, sum(1 for m in marks if m[2]) // n))
