"""In-step duration of the BMU distance pass (bench.py's event timer) against how many encoder blocks run as two
half-batch chains: the denser the forward, the slower the kernels right after it (power management)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vit_som_amd import ViTSOM, ops
model = ViTSOM(bench.c3_config(512), device="cuda")
model.set_schedule(50000, 10000); model._it = 1000
(opt,), _ = model.configure_optimizers()
x = torch.randn(512, 3, 32, 32, device="cuda"); y = torch.zeros(512, dtype=torch.int64, device="cuda")
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        model.train_step_fused(x, y); opt.step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
for r in range(3):
    for val in ("0", "6", "12"):
        from vit_som_amd.tuning import hooks
        hooks.set(fwd_split_blocks=int(val))
        run(3)
        ops.enable_timer("bmu_cosine_dots")
        t = run(20)
        ms, calls = ops.timer_ms("bmu_cosine_dots")
        ops.disable_timers()
        print(f"blocks split={val:>2s}: step {t:.2f} ms   BMU (events) {ms*1e3:.1f} us")
