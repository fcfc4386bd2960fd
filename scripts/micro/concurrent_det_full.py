"""Full-size check of the rule found in round 2 (a kernel was not deterministic on a GPU shared by two processes,
profiles/r02_two_process_determinism.txt): the benchmark's workload (filters 32, 256x256, full UDA) at batch B for two
steps, solo and then in TWO independent processes at once, parameters compared bit for bit.
usage: concurrent_det_full.py [B=8] [procs=2]"""
import os, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch, torch.multiprocessing as mp


def run(b):
    import bench as BN
    dev = torch.device("cuda", 0)
    wl = dict(BN.WORKLOADS["full_uda"])
    wl["pn"] = {"drop": 0.0}                 # (dropout draws from the device RNG: not part of this comparison)
    tr = BN.build_trainer(wl, dev, 0)
    batch = BN.synth_device_batch(b, 256, 4, 1, dev)
    for _ in range(2):
        tr.step(*batch)
    torch.cuda.synchronize()
    return [o.p.detach().cpu().clone() for o in [tr.opt_gen, tr.opt_d1, tr.opt_d2, tr.opt_d4]]


def work(rank, b, tmp):
    sys.path.insert(0, root)
    torch.save(run(b), os.path.join(tmp, "p%d.pt" % rank))


if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    solo = run(b)
    tmp = tempfile.mkdtemp()
    mp.spawn(work, args=(b, tmp), nprocs=procs, join=True)
    bad = 0
    for r in range(procs):
        got = torch.load(os.path.join(tmp, "p%d.pt" % r))
        for i, nm in enumerate(("seg", "d1", "d2", "d4")):
            d = (solo[i] - got[i]).abs()
            if float(d.max()) > 0:
                bad += 1
                print("process %d %s differs from the solo run: max %.3e in %d elements" % (r, nm, float(d.max()), int((d > 0).sum())))
    print("B=%d, %d concurrent processes: %s" % (b, procs, "bit-identical to the solo run" if not bad else "%d buffers differ" % bad))
