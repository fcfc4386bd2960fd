"""bandwidth of the BatchNorm-backward apply pass (reads dy, a; writes dz) on one layer shape
usage: apply_bw.py channels size n"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import pointcloududa_amd.kernels as K
c, hw, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
a, dy = torch.randn(n, c, hw, hw, device=dev), torch.randn(n, c, hw, hw, device=dev)
gamma = torch.ones(c, device=dev)
p, nt, cnt = K.bn_stats(a)
st = K.bn_finalize(p, nt, cnt, gamma, torch.zeros(c, device=dev), None, None)
dg, dbt = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
# the reduce once (its partials are reused), then time finalize + apply only
conv = K.ConvOp(c, c, 3, pad=1)
_, red = conv.dgrad(dy, torch.randn(c, c, 3, 3, device=dev) * 0.05, hw, hw, bnred=(a, st))
fn = lambda: K.bn_backward(dy, a, st, gamma, dg, dbt, act_slope=0.2, red=red)
fn(); torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20):
    fn()
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 20
print("PCUDA_APPLY_CH=%s  %dch %d^2 n%d: %.1f us  %.2f TB/s" % (os.environ.get("PCUDA_APPLY_CH", "-"), c, hw, n, ms * 1e3, 3 * a.numel() * 4 / ms / 1e9))
