"""debug aid: are the kernels deterministic when two processes with several busy streams share the GPU?
Each process repeats the same launch and counts results that differ from its own first one.
usage: d1_repro.py NPROC ITERS VICTIM HEAVY   (env SIDE = extra busy streams per process)"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch, torch.multiprocessing as mp


def work(rank, iters, victim, heavy):
    import pointcloududa_amd.kernels as K
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    big = K.ConvOp(32, 32, 3, stride=1, pad=1)
    wb, xb, bb = rn(32, 32, 3, 3) * 0.05, rn(4, 32, 256, 256), torch.zeros(32, device=dev)
    ma = rn(2048, 2048)
    if victim == "d1dgrad":
        op, w, dy = K.ConvOp(4, 64, 4, stride=2, pad=2), rn(64, 4, 4, 4) * 0.05, rn(4, 64, 49, 49)
        run = lambda: op.dgrad(dy, w, 96, 96)
    elif victim == "d2dgrad":        # the discriminators' second layer (MFMA kernels, four parity classes)
        op, w, dy = K.ConvOp(64, 128, 4, stride=2, pad=2), rn(128, 64, 4, 4) * 0.05, rn(4, 128, 25, 25)
        run = lambda: op.dgrad(dy, w, 49, 49)
    elif victim == "conv":
        op, w, x, b = K.ConvOp(16, 16, 3, stride=1, pad=1), rn(16, 16, 3, 3) * 0.05, rn(4, 16, 128, 128), torch.zeros(16, device=dev)
        run = lambda: op.forward(x, w, b, 1.0, 128, 128)[0]
    elif victim == "c1fwd":
        op, w, x, b = K.ConvOp(1, 4, 3, stride=1, pad=1), rn(4, 1, 3, 3) * 0.05, rn(4, 1, 256, 256), torch.zeros(4, device=dev)
        run = lambda: op.forward(x, w, b, 1.0, 256, 256)[0]
    elif victim == "pwdgrad":
        op, w, dy = K.ConvOp(4, 4, 1), rn(4, 4, 1, 1) * 0.05, rn(4, 4, 256, 256)
        run = lambda: op.dgrad(dy, w, 256, 256)
    elif victim == "c1wgrad":
        op, x, dy = K.ConvOp(1, 32, 3, stride=1, pad=1), rn(4, 1, 256, 256), rn(4, 32, 256, 256)
        def run():
            dw, db = torch.zeros(32, 1, 3, 3, device=dev), torch.zeros(32, device=dev)
            op.wgrad(x, dy, dw, db, 256, 256)
            return torch.cat([dw.flatten(), db])
    elif victim == "c1fwd32":
        op, w, x, b = K.ConvOp(1, 32, 3, stride=1, pad=1), rn(32, 1, 3, 3) * 0.05, rn(4, 1, 256, 256), torch.zeros(32, device=dev)
        run = lambda: torch.cat([t.flatten() for t in op.forward(x, w, b, 0.2, 256, 256, want_stats=True)[:2]])
    elif victim == "pwfwd":
        op, w, x, b = K.ConvOp(32, 4, 1), rn(4, 32, 1, 1) * 0.05, rn(4, 32, 256, 256), torch.zeros(4, device=dev)
        run = lambda: op.forward(x, w, b, 1.0, 256, 256)[0]
    elif victim == "pwdgrad32":
        op, w, dy = K.ConvOp(32, 4, 1), rn(4, 32, 1, 1) * 0.05, rn(4, 4, 256, 256)
        a = rn(4, 32, 256, 256)
        p_, nt_, cnt_ = K.bn_stats(a)
        bst = K.bn_finalize(p_, nt_, cnt_, torch.ones(32, device=dev), torch.zeros(32, device=dev), None, None)
        def run():
            dx, red = op.dgrad(dy, w, 256, 256, bnred=(a, bst))
            return torch.cat([dx.flatten(), red[0].flatten()])
    elif victim.startswith("mfma:"):      # mfma:cin,cout,hw,k,stride,pad,which(fwd|dgrad|wgrad),n
        cin, cout, hw_, k_, s_, p_, which, n_ = victim[5:].split(",")
        cin, cout, hw_, k_, s_, p_, n_ = int(cin), int(cout), int(hw_), int(k_), int(s_), int(p_), int(n_)
        op = K.ConvOp(cin, cout, k_, stride=s_, pad=p_)
        oh, ow = op.out_hw(hw_, hw_)
        x, w, b, dy = rn(n_, cin, hw_, hw_), rn(cout, cin, k_, k_) * 0.05, torch.zeros(cout, device=dev), rn(n_, cout, oh, ow)
        if which == "fwd":
            run = lambda: torch.cat([t.flatten() for t in op.forward(x, w, b, 0.2, hw_, hw_, want_stats=True)[:2]])
        elif which == "dgrad":
            run = lambda: op.dgrad(dy, w, hw_, hw_)
        else:
            def run():
                dw, db = torch.zeros_like(w), torch.zeros_like(b)
                op.wgrad(x, dy, dw, db, hw_, hw_)
                return torch.cat([dw.flatten(), db])
    elif victim.startswith("pw:"):        # the LDS-free pointwise / loss / dense kernels on segmenter-sized tensors
        name = victim[3:]
        a, dyv = rn(4, 32, 256, 256), rn(4, 32, 256, 256)
        gam, bet = torch.ones(32, device=dev), torch.zeros(32, device=dev)
        p_, nt_, cnt_ = K.bn_stats(a)
        bst = K.bn_finalize(p_, nt_, cnt_, gam, bet, None, None)
        logits = rn(4, 4, 256, 256)
        onehot = torch.nn.functional.one_hot(torch.randint(0, 4, (4, 256, 256), generator=g), 4).permute(0, 3, 1, 2).contiguous().to(torch.uint8).to(dev)
        if name == "bnstats":
            run = lambda: K.bn_stats(a)[0]
        elif name == "bnapply":
            run = lambda: K.bn_apply(a, bst)
        elif name == "bnbwd":
            run = lambda: K.bn_backward(dyv, a, bst, gam, torch.zeros(32, device=dev), torch.zeros(32, device=dev), act_slope=0.2)
        elif name == "maxpool":
            run = lambda: K.maxpool2_fwd(a)[0]
        elif name == "maxpoolbwd":
            pooled, idx = K.maxpool2_fwd(a)
            dyp = rn(*pooled.shape)
            run = lambda: K.maxpool2_bwd(dyp, idx, 256, 256)
        elif name == "upbwd":
            run = lambda: K.upsample2_bwd(dyv)
        elif name == "addn":
            b2, b3 = rn(4, 32, 256, 256), rn(4, 32, 256, 256)
            run = lambda: K.add_n([a, b2, b3])
        elif name == "entropy":
            run = lambda: K.entropy_fwd(logits, "sigmoid", 1.0, False)[0]
        elif name == "entropybwd":
            de = rn(4, 4, 256, 256)
            run = lambda: K.entropy_bwd(logits, "sigmoid", 1.0, de, None)
        elif name == "segloss":
            run = lambda: torch.stack(list(K.seg_loss_fwd(logits, onehot, "sigmoid")[:2]))
        elif name == "seglossbwd":
            ws = K.seg_loss_fwd(logits, onehot, "sigmoid")[2]
            run = lambda: K.seg_loss_bwd(logits, onehot, "sigmoid", ws)
        elif name == "adam":
            pp, gg = rn(1 << 22), rn(1 << 22)
            def run():
                q, m_, v_ = pp.clone(), torch.zeros_like(pp), torch.zeros_like(pp)
                K.adam_step(q, gg, m_, v_, 1e-3, 0.9, 0.99, 1e-8, 0.0, 1)
                return q
        elif name == "linear":
            xl, wl, bl = rn(9600, 121), rn(3, 121), rn(3)
            run = lambda: K.linear_fwd(xl, wl, bl)
        elif name == "unfold":
            xu = rn(4, 4, 256, 256)
            run = lambda: K.unfold_taps(xu, 4, 2, 2)
        else:
            raise SystemExit("pw victim?")
    elif victim == "clone":
        x = rn(4, 32, 256, 256)
        run = lambda: x.clone()
    elif victim == "axpb":
        x = rn(4, 32, 256, 256)
        run = lambda: x * 1.5 + 2.0
    elif victim == "lrelubwd":
        a, dy = rn(4, 32, 256, 256), rn(4, 32, 256, 256)
        run = lambda: K.lrelu_bwd(dy, a, 0.2)
    elif victim == "d1wgrad":
        op, x, dy = K.ConvOp(4, 64, 4, stride=2, pad=2), rn(4, 4, 96, 96), rn(4, 64, 49, 49)
        def run():
            dw = torch.zeros(64, 4, 4, 4, device=dev)
            op.wgrad(x, dy, dw, None, 96, 96)
            return dw
    elif victim == "d5fwd":
        op, w, x, b = K.ConvOp(512, 1, 4, stride=2, pad=2), rn(1, 512, 4, 4) * 0.05, rn(4, 512, 7, 7), torch.zeros(1, device=dev)
        run = lambda: op.forward(x, w, b, 1.0, 7, 7)[0]
    elif victim == "torchconv":
        w, x = rn(64, 4, 4, 4) * 0.05, rn(4, 4, 96, 96)
        run = lambda: torch.nn.functional.conv2d(x, w, stride=2, padding=2)
    else:
        raise SystemExit("victim?")
    side = [torch.cuda.Stream() for _ in range(int(os.environ.get("SIDE", "4")))]
    ref = run().clone()
    torch.cuda.synchronize()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    worst = torch.zeros((), device=dev)
    for it in range(iters):
        if side and it % 8 == 0:
            for st in side:
                with torch.cuda.stream(st):
                    torch.mm(ma, ma)
        if heavy == "conv":
            big.forward(xb, wb, bb, 1.0, 256, 256)
        elif heavy == "mm":
            torch.mm(ma, ma)
        d = (run() - ref).abs().max()
        bad += (d > 0).long()
        worst = torch.maximum(worst, d)
    torch.cuda.synchronize()
    print("%-9s heavy=%-4s rank %d mismatching launches %5d of %d worst abs diff %.3e (max |ref| %.3e)" % (
        victim, heavy, rank, int(bad), iters, float(worst), float(ref.abs().max())), flush=True)


if __name__ == "__main__":
    nproc, iters, victim, heavy = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    if nproc == 1:
        work(0, iters, victim, heavy)
    else:
        mp.spawn(work, args=(iters, victim, heavy), nprocs=nproc, join=True)
