"""debug aid: what do the wrong results of the classifier-forward kernel look like when two processes share the GPU?"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch, torch.multiprocessing as mp


def work(rank, iters):
    import pointcloududa_amd.kernels as K
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    big = K.ConvOp(32, 32, 3, stride=1, pad=1)
    wb, xb, bb = rn(32, 32, 3, 3) * 0.05, rn(4, 32, 256, 256), torch.zeros(32, device=dev)
    op, w, x, b = K.ConvOp(32, 4, 1), rn(4, 32, 1, 1) * 0.05, rn(4, 32, 256, 256), torch.zeros(4, device=dev)
    ref = op.forward(x, w, b, 1.0, 256, 256)[0].clone()
    torch.cuda.synchronize()
    shown = 0
    for it in range(iters):
        big.forward(xb, wb, bb, 1.0, 256, 256)
        y = op.forward(x, w, b, 1.0, 256, 256)[0]
        if it % 50 == 49 or True:
            d = (y != ref)
            nbad = int(d.sum())
            if nbad and shown < 4 and rank == 0:
                shown += 1
                idx = d.nonzero()
                n_, c_ = idx[:, 0], idx[:, 1]
                lin = idx[:, 2] * 256 + idx[:, 3]
                print("launch %d: %d wrong values; images %s channels %s" % (it, nbad, sorted(set(n_.tolist())), sorted(set(c_.tolist()))), flush=True)
                # runs of consecutive linear pixel indices per (n, c)
                for nn in sorted(set(n_.tolist()))[:2]:
                    for cc in sorted(set(c_.tolist())):
                        l = lin[(n_ == nn) & (c_ == cc)].tolist()
                        runs, s0 = [], None
                        for a, bnxt in zip(l, l[1:] + [None]):
                            if s0 is None: s0 = a
                            if bnxt != a + 1: runs.append((s0, a - s0 + 1)); s0 = None
                        print("   n%d c%d: %d runs, first (start, len): %s  start%%1024: %s" % (nn, cc, len(runs), runs[:6], [r[0] % 1024 for r in runs[:6]]), flush=True)
                # are the wrong values what ZERO weights / zero inputs would give?
                print("   sample wrong/ref:", [(round(float(y[tuple(i)]), 4), round(float(ref[tuple(i)]), 4)) for i in idx[:6]], flush=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    mp.spawn(work, args=(int(sys.argv[1]) if len(sys.argv) > 1 else 400,), nprocs=2, join=True)
