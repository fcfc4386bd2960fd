import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.full((1000,), float(rank + 1), device="cuda:0")
    h = dist.all_reduce(t, async_op=True)
    h.wait()
    torch.cuda.synchronize()
    print(rank, float(t[0]), flush=True)
    b = torch.full((10,), float(rank), device="cuda:0"); dist.broadcast(b, src=0); print(rank, "bcast", float(b[0]), flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2, 29533), nprocs=2, join=True)
