import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=dev)
t = torch.ones(25_000_000, device=dev)
dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print("nccl(RCCL) 1-rank all_reduce ok", float(t.sum()), dist.get_backend())
dist.destroy_process_group()
