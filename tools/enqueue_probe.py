"""Is the step GPU-bound or launch-bound?  Times the host-side enqueue of K steps (no sync) against
the wall time until the GPU drains."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vit_som_amd import ViTSOM

cfg = bench.c3_config(512)
torch.manual_seed(0)
model = ViTSOM(cfg, device="cuda")
model.set_schedule(50000, 10000)
(opt,), _ = model.configure_optimizers()
x = torch.rand(512, 3, 32, 32, device="cuda"); y = torch.randint(0, 10, (512,), device="cuda")
for _ in range(5):
    model.train_step_fused(x, y); opt.step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    model.train_step_fused(x, y); opt.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/K:.2f} ms/step, drained {1e3*(t2-t0)/K:.2f} ms/step")
