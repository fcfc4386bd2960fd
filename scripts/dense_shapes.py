"""which dense (PointNet / point head) GEMMs cost what: per-call HIP-event timing over one train step"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from pointcloududa_amd import kernels as K
dev = torch.device("cuda", 0)
tr = B.build_trainer(B.WORKLOADS["full_uda"], dev, seed=0)
batch = B.synth_device_batch(32, 256, 4, seed=100, dev=dev)
for _ in range(2): tr.step(*batch)
rec = []
def wrap(name):
    f = getattr(K, name)
    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a, **k); e1.record()
        rec.append((name, tuple(tuple(t.shape) for t in a if torch.is_tensor(t)), e0, e1))
        return r
    setattr(K, name, g)
for n in ("linear_fwd", "linear_bwd_x", "linear_bwd_w", "bmm", "max_points_fwd", "max_points_bwd"): wrap(n)
# modules captured K functions by attribute access at call time (K.linear_fwd), so the wrappers are seen
tr.step(*batch); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for name, shp, e0, e1 in rec:
    a = agg[(name, shp)]; a[0] += e0.elapsed_time(e1); a[1] += 1
tot = sum(v[0] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]: print("%7.3f ms n=%2d avg %6.1f us  %s %s" % (v[0], v[1], 1e3 * v[0] / v[1], k[0], k[1]))
print("total %.3f ms over %d calls" % (tot, len(rec)))
