"""one conv forward case against torch (debug aid): python scripts/micro/conv_case.py n cin cout h w k s p"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from pointcloududa_amd import kernels as K
n, cin, cout, h, w_, k, s, p = [int(v) for v in sys.argv[1:9]]
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.normal(0, 1, (n, cin, h, w_)).astype(np.float32))
w = torch.from_numpy(rng.normal(0, 0.05, (cout, cin, k, k)).astype(np.float32))
z = F.conv2d(x, w, None, stride=s, padding=p)
op = K.ConvOp(cin, cout, k, stride=s, pad=p)
y, _, _ = op.forward(x.cuda(), w.cuda(), None, 1.0, h, w_)
d = (y.cpu() - z).abs()
print(os.environ.get("TAGX", ""), "max err", float(d.max()), "ref max", float(z.abs().max()), "bad per image", [int((d[i] > 1e-3).sum()) for i in range(n)],
      "bad per row of img0", [int((d[0][:, r] > 1e-3).sum()) for r in range(d.shape[2])])
