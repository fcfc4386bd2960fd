"""debug aid: the two-step trajectory of tests/test_dist_gpu.py in ONE process, with and without the data-parallel
flow (one-rank gloo group + PCUDA_FORCE_COLLECTIVES=1); prints parameter checksums per network"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
dp = len(sys.argv) > 1 and sys.argv[1] == "dp"
if dp:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", PCUDA_FORCE_COLLECTIVES="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
import test_dist_gpu as T
dev = torch.device("cuda", 0)
out = T._run(T._build(0, dev), [9, 10], dev)
for name, t in zip(("seg", "d1", "d2", "d4"), out):
    print("dp" if dp else "sp", name, "%.10e" % float(t.double().abs().sum()), "%.10e" % float((t.double() ** 2).sum()))
torch.save(out, "/tmp/dpflow_%s.pt" % ("dp" if dp else "sp"))
if os.path.exists("/tmp/dpflow_sp.pt") and os.path.exists("/tmp/dpflow_dp.pt"):
    a, b = torch.load("/tmp/dpflow_sp.pt"), torch.load("/tmp/dpflow_dp.pt")
    for name, x, y in zip(("seg", "d1", "d2", "d4"), a, b):
        d = (x - y).abs()
        print("diff", name, float(d.max()), int((d > 0).sum()), "of", d.numel(), "first at", int(torch.nonzero(d > 0)[0]) if (d > 0).any() else -1)
