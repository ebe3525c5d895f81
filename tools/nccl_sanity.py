"""RCCL sanity on the one-GPU box: a 1-rank "nccl" group, then the model's exchange code path
(bucketed early all-reduces from the comm stream + the remaining pieces) with world_size forced to 2
so that the overlapped branch runs; a 1-rank sum is the identity, so the trajectory must equal the
non-overlapped one bit for bit."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=dev)
t = torch.ones(25_000_000, device=dev)
dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
print("nccl(RCCL) 1-rank all_reduce ok", float(t.sum()), dist.get_backend())

import bench
from vit_som_amd import ViTSOM

def run(overlap, steps=6):
    from vit_som_amd.tuning import hooks
    hooks.set(overlap_allreduce=bool(overlap))
    torch.manual_seed(0)
    m = ViTSOM(bench.c3_config(128), device="cuda")
    m.set_schedule(50000, 1000)
    m.set_distributed(1, 0); m.world_size = 2            # force the N > 1 branches on a 1-rank group
    (opt,), _ = m.configure_optimizers()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(128, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 10, (128,), generator=g).cuda()
    losses, used = [], 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        loss = m.train_step_fused(x, y)
        used += len(m._works)
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    return losses, used, m.arena.params.clone(), (time.perf_counter() - t0) / steps

l0, u0, p0, t0 = run(False)
l1, u1, p1, t1 = run(True)
print("no overlap :", ["%.6f" % v for v in l0], "early pieces", u0, "%.2f ms/step" % (1e3 * t0))
print("overlap    :", ["%.6f" % v for v in l1], "early pieces", u1, "%.2f ms/step" % (1e3 * t1))
assert u0 == 0 and u1 >= 6 * 4, (u0, u1)
assert l0 == l1 and torch.equal(p0, p1), "overlapped exchange changed the result"
print("overlapped exchange: identical trajectory")
dist.destroy_process_group()
