"""how far ahead of the GPU is the host?  time to ISSUE n steps vs time until the GPU has finished them"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pointcloududa_amd as P
dev = torch.device("cuda", 0)
P.set_precision("bf16x3")
wl = bench.WORKLOADS["full_uda"]
tr = bench.build_trainer(wl, dev, seed=0)
batch = bench.synth_device_batch(wl["batch"], 256, 4, seed=100, dev=dev)
for _ in range(5):
    tr.step(*batch)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    tr.step(*batch)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("issue %.2f ms/step   complete %.2f ms/step" % (1e3 * t_issue / n, 1e3 * t_all / n))
# lead of the host in steady state: when step k has been issued, has the GPU finished step k-1 / k-2?
evs, lead = [], []
for k in range(30):
    tr.step(*batch)
    e = torch.cuda.Event(); e.record(); evs.append(e)
    lead.append(sum(0 if x.query() else 1 for x in evs))     # issued but unfinished steps
torch.cuda.synchronize()
print("unfinished steps at issue time:", lead)
