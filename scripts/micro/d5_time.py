"""time the discriminators' last layer (cin -> 1, 4x4 stride 2 pad 2 on 17x17): direct kernel vs the MFMA kernel"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloududa_amd import kernels as K
dev = torch.device("cuda", 0)
for n, cin, hw in ((32, 512, 17), (64, 512, 17), (16, 512, 17), (32, 512, 9)):
    op = K.ConvOp(cin, 1, 4, stride=2, pad=2)
    x = torch.randn(n, cin, hw, hw, device=dev); w = torch.randn(1, cin, 4, 4, device=dev) * 0.05
    f = lambda: op.forward(x, w, None, 1.0, hw, hw)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize()
    print("d5 fwd n%d cin%d %dx%d: %.1f us" % (n, cin, hw, hw, (time.perf_counter() - t0) / 50 * 1e6))
