"""cProfile of the host side of the training step (where do the ~12 ms of enqueue time go?)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vit_som_amd import ViTSOM
model = ViTSOM(bench.c3_config(512), device="cuda")
model.set_schedule(50000, 10000)
(opt,), _ = model.configure_optimizers()
x = torch.rand(512, 3, 32, 32, device="cuda"); y = torch.randint(0, 10, (512,), device="cuda")
for _ in range(3):
    model.train_step_fused(x, y); opt.step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    model.train_step_fused(x, y); opt.step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
