"""time forward / dgrad / wgrad of one convolution shape (HIP events, 20 launches each):
python scripts/micro/conv_time.py n cin cout h w k s p   (env switches apply: PCUDA_W8, PCUDA_NOPIPE, ...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloududa_amd import kernels as K
n, cin, cout, h, w_, k, s, p = [int(v) for v in sys.argv[1:9]]
dev = torch.device("cuda", 0)
x, w = torch.randn(n, cin, h, w_, device=dev), torch.randn(cout, cin, k, k, device=dev) * 0.05
b = torch.zeros(cout, device=dev)
op = K.ConvOp(cin, cout, k, stride=s, pad=p)
oh, ow = op.out_hw(h, w_)
dy = torch.randn(n, cout, oh, ow, device=dev)
dw, db = torch.zeros_like(w), torch.zeros_like(b)


def timed(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3


fl = 2.0 * n * oh * ow * cout * cin * k * k
tag = " ".join("%s=%s" % (e, os.environ[e]) for e in sorted(os.environ) if e.startswith("PCUDA_"))
for name, fn in (("fwd", lambda: op.forward(x, w, b, 0.2, h, w_)), ("dgrad", lambda: op.dgrad(dy, w, h, w_)),
                 ("wgrad", lambda: op.wgrad(x, dy, dw, db, h, w_))):
    us = timed(fn)
    print("%-22s n%d %d->%d %dx%d k%d s%d  %-5s %7.1f us %6.1f T/s" % (tag, n, cin, cout, h, w_, k, s, name, us, fl / us / 1e6))
