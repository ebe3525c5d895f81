"""Does the round quantisation of the step's GEMM launches (T = 512 x 65 = 260 row tiles of 128 = 256 + 4) cost anything INSIDE the
step, where two kernels share the chip most of the time?  Per-image time at batch 504 (T = 32760: 256 row tiles, the last one 8 rows
short) against batch 512, in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from vit_som_amd import ViTSOM

def run(B, steps=30):
    torch.manual_seed(0)
    m = ViTSOM(bench.c3_config(B), device="cuda")
    m.set_schedule(50000, 10000)
    (opt,), _ = m.configure_optimizers()
    x = torch.randn(B, 3, 32, 32, device="cuda"); y = torch.zeros(B, dtype=torch.int64, device="cuda")
    for _ in range(6):
        m.train_step_fused(x, y); opt.step()
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            m.train_step_fused(x, y); opt.step()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps)
    return best * 1e3

run(64, 5)
for rep in range(2):
    for B in (512, 504, 496, 520):
        ms = run(B)
        print(f"batch {B}: {ms:.3f} ms/step = {ms / B * 1e3:.3f} us per image", flush=True)
