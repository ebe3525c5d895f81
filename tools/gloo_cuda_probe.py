import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, port):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    t = torch.full((1000,), float(rank + 1), device="cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        wk = dist.all_reduce(t, async_op=True)
    wk.wait()
    torch.cuda.synchronize()
    print(rank, t[:3].tolist(), flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(29533,), nprocs=2, join=True)
