"""debug aid: tests/test_dist_gpu.py's comparison with details (which parameters differ between the two-rank run on
equal shards and the single-process run, and by how much)"""
import os, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, torch.multiprocessing as mp
import test_dist_gpu as T

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    nets = T._build(0, dev)
    names = [k for k, _ in nets[0].named_parameters()]
    sizes = [p.numel() for _, p in nets[0].named_parameters()]
    single = T._run(nets, [9, 10], dev)
    torch.cuda.synchronize()
    tmp = tempfile.mkdtemp()
    mp.spawn(T._worker, args=(2, T._free_port(), tmp), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp, "rank0.pt")); r1 = torch.load(os.path.join(tmp, "rank1.pt"))
    for i, nm in enumerate(("seg", "d1", "d2", "d4")):
        d = (r0["same"][i] - single[i]).abs()
        print(nm, "max diff vs single", float(d.max()), "n diff", int((d > 0).sum()), "| r0 vs r1", float((r0["same"][i] - r1["same"][i]).abs().max()))
    d = (r0["same"][0] - single[0]).abs()
    off = 0
    for k, n in zip(names, sizes):
        n_al = (n + 63) // 64 * 64
        seg = d[off:off + n]
        if float(seg.max()) > 0:
            print("  %-40s max %.3e  rel %.3e  (%d of %d)" % (k, float(seg.max()), float(seg.max() / single[0][off:off + n].abs().max().clamp_min(1e-30)), int((seg > 0).sum()), n))
        off += n_al
