import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import nets as ON
from pointcloududa_amd.networks import Segmentation_model_Point
from test_networks_gpu import _load
dev = torch.device("cuda", 0)
cfg_kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
params = ON.make_params(ON.seg_param_shapes(ON.SegCfg(**cfg_kw)), 100)
m = _load(Segmentation_model_Point(**cfg_kw), params, dev)
rng = np.random.default_rng(101)
x = torch.from_numpy(rng.random((2, 1, 128, 128), dtype=np.float32)).to(dev)
P = m._tensor_dict()
_, _, S = m._engine.forward(P, x, True)
out = {}
for blk, v in S.items():
    if isinstance(v, (tuple, list)):
        for i, t in enumerate(v):
            if torch.is_tensor(t): out["%s/%d" % (blk, i)] = t.detach().cpu()
            elif hasattr(t, "t") and torch.is_tensor(t.t): out["%s/%d.t" % (blk, i)] = t.t.detach().cpu()
            elif hasattr(t, "mean"):
                out["%s/%d.mean" % (blk, i)] = t.mean.detach().cpu(); out["%s/%d.invstd" % (blk, i)] = t.invstd.detach().cpu()
path = sys.argv[1]
if os.path.exists(path):
    ref = torch.load(path)
    for k in sorted(out):
        if k in ref and ref[k].shape == out[k].shape:
            e = ((out[k].double() - ref[k].double()).abs().max() / (ref[k].double().abs().max() + 1e-30)).item()
            if e > 1e-5: print("%-40s %.3e" % (k, e))
    print("compared", len(out))
else:
    torch.save(out, path); print("saved", len(out))
