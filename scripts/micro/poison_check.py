"""debug aid: run the small two-step trajectory with every torch.empty() filled with NaN (deterministic-mode fill):
a NaN in the parameters afterwards means some kernel read memory nobody had written"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
import test_dist_gpu as T
dev = torch.device("cuda", 0)
out = T._run(T._build(0, dev), [9, 10], dev)
for name, t in zip(("seg", "d1", "d2", "d4"), out):
    print(name, "nan:", int(torch.isnan(t).sum()), "of", t.numel())
