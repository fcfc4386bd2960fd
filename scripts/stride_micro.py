"""does the channel (plane) stride matter? time a 3x3 32->32 conv at 256x256 with padded plane strides"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloududa_amd import kernels as K
dev = torch.device("cuda", 0)
n, cin, cout, h, w = 32, int(os.environ.get("C", 32)), int(os.environ.get("C", 32)), int(os.environ.get("HW", 256)), int(os.environ.get("HW", 256))
op = K.ConvOp(cin, cout, 3, pad=1)
wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05; b = torch.zeros(cout, device=dev)
def padded(c, pad, npad=0):
    buf = torch.randn(n, c * (h * w + pad) + npad, device=dev)
    return buf[:, :c * (h * w + pad)].view(n, c, h * w + pad)[:, :, :h * w].view(n, c, h, w)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
fl = 2.0 * n * h * w * cout * cin * 9
for pad_in, pad_out, npad in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (32, 32, 0), (1024, 1024, 0), (64, 64, 4096), (96, 96, 1056), (0, 0, 1056)]:
    x = padded(cin, pad_in, npad); y = padded(cout, pad_out, npad)
    tf = t(lambda: op.forward(x, wt, b, 0.01, h, w, out=y))
    print("pad_in %5d pad_out %5d npad %5d: fwd %7.3f ms %6.1f TF  (x strides %s)" % (pad_in, pad_out, npad, tf * 1e3, fl / tf / 1e12, x.stride()[:2]), flush=True)
