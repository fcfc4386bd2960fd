import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pointcloududa_amd import kernels as K
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (n, cin, cout, h) in [(2, 4, 4, 128), (2, 8, 4, 128), (2, 4, 8, 64), (1, 4, 4, 32)]:
    x = torch.randn(n, cin, h, h, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) * 0.2; b = torch.randn(cout, device=dev)
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev)
    op = K.ConvOp(cin, cout, 3, stride=1, pad=1)
    y, part, nt = op.forward(K.TA(x, sc, sh), w, b, 0.01, h, h, want_stats=True)
    xr = x * sc[None, :, None, None] + sh[None, :, None, None]
    yr = F.leaky_relu(F.conv2d(xr, w, b, padding=1), 0.01)
    s = part[:nt].double().sum(0)
    print(n, cin, cout, h, "y err %.2e" % ((y - yr).abs().max() / yr.abs().max()).item(),
          "sum err %.2e" % ((s[:, 0] - yr.double().sum((0, 2, 3))).abs().max() / yr.double().sum((0, 2, 3)).abs().max()).item(),
          "sq err %.2e" % ((s[:, 1] - (yr.double() ** 2).sum((0, 2, 3))).abs().max() / (yr.double() ** 2).sum((0, 2, 3)).abs().max()).item(), "nt", nt)
