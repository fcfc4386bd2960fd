import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from pointcloududa_amd import kernels as K
dev = torch.device("cuda", 0)
n, cin, cout, h, w = [int(v) for v in (sys.argv[1:6] or (2, 32, 32, 16, 32))]
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w)).astype(np.float32))
wt = torch.from_numpy(rng.normal(0, 0.1, (cout, cin, 3, 3)).astype(np.float32))
ref = F.conv2d(x, wt, None, padding=1)
yr, _, _ = K.rconv3_forward(K.rec_from_nchw(x.to(dev)), K.rconv3_pack(wt.to(dev)), None, 1.0, cout)
y = K.rec_to_nchw(yr, cout).cpu()
err = (y - ref).abs()
print("max err", float(err.max()), "ref max", float(ref.abs().max()))
print("per image:", err.amax((1, 2, 3)).tolist())
print("per row (image 0):", [round(v, 4) for v in err[0].amax((0, 2)).tolist()])
print("per column (image 0):", [round(v, 4) for v in err[0].amax((0, 1)).tolist()])
print("per channel (image 0):", [round(v, 4) for v in err[0].amax((1, 2)).tolist()])
