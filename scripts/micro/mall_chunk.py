"""experiment: does processing a hi-resolution layer's backward chain (BatchNorm-backward apply -> weight gradient ->
data gradient) in batch CHUNKS that fit the 256 MB infinity cache beat one pass over the whole batch?
usage: mall_chunk.py [channels] [size] [n]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import pointcloududa_amd.kernels as K
from pointcloududa_amd.kernels import TA

c = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
a0, a1, dy = rn(n, c, hw, hw), rn(n, c, hw, hw), rn(n, c, hw, hw)
w = rn(c, c, 3, 3) * 0.05
gamma = torch.ones(c, device=dev)
conv = K.ConvOp(c, c, 3, pad=1)


def stats(a):
    p, nt, cnt = K.bn_stats(a)
    return K.bn_finalize(p, nt, cnt, gamma, torch.zeros(c, device=dev), None, None)


st0, st1 = stats(a0), stats(a1)
dw, db, dgam, dbet = torch.zeros_like(w), torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev)


def chain(lo, hi, red):
    s = slice(lo, hi)
    dz = K.bn_backward(dy[s], a1[s], st1, gamma, dgam, dbet, act_slope=0.2, red=red)
    conv.wgrad(TA(a0[s], st0.scale, st0.shift), dz, dw, db, hw, hw)
    return conv.dgrad(dz, w, hw, hw, bnred=(a0[s], st0))


# backward-reduce partials of the right shape (their values do not matter for the timing)
_, red_full = conv.dgrad(dy, w, hw, hw, bnred=(a1, st1))
assert red_full is not None


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


print("tensor size %.0f MB" % (a0.numel() * 4 / 1e6))
print("whole batch      %.3f ms" % timed(lambda: chain(0, n, red_full)))
for parts in (2, 4, 8):
    m = n // parts
    _, red_c = conv.dgrad(dy[:m], w, hw, hw, bnred=(a1[:m], st1))
    def run():
        for q in range(parts):
            chain(q * m, (q + 1) * m, red_c)
    print("%d chunks of %2d   %.3f ms" % (parts, m, timed(run)))
