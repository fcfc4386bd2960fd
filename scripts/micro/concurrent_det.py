"""debug aid: is the small two-step trajectory bit-identical when TWO independent processes run it at the same time on
one GPU (no process group at all)?"""
import os, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, torch.multiprocessing as mp
import test_dist_gpu as T


def work(rank, tmp):
    sys.path.insert(0, root)
    dev = torch.device("cuda", 0)
    out = T._run(T._build(0, dev), [9, 10], dev)
    torch.save(out, os.path.join(tmp, "p%d.pt" % rank))


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    solo = T._run(T._build(0, dev), [9, 10], dev)
    torch.cuda.synchronize()
    tmp = tempfile.mkdtemp()
    mp.spawn(work, args=(tmp,), nprocs=2, join=True)
    a, b = torch.load(os.path.join(tmp, "p0.pt")), torch.load(os.path.join(tmp, "p1.pt"))
    for i, nm in enumerate(("seg", "d1", "d2", "d4")):
        print(nm, "solo vs p0", float((solo[i] - a[i]).abs().max()), int(((solo[i] - a[i]) != 0).sum()), "| solo vs p1",
              float((solo[i] - b[i]).abs().max()), "| p0 vs p1", float((a[i] - b[i]).abs().max()))
