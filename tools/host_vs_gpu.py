"""Host enqueue time vs GPU time of the training step at small per-GPU batches: for each batch size,
ms per step with the queue kept full (what bench.py reports), the host time to ENQUEUE a step (perf_counter around the calls,
GPU idle-free) and the GPU-side time of one step run alone behind a synchronize (HIP events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vit_som_amd import ViTSOM

def run(B, cfgfn, steps=30):
    torch.manual_seed(0)
    m = ViTSOM(cfgfn(B), device="cuda")
    m.set_schedule(50000, 10000)
    (opt,), _ = m.configure_optimizers()
    x = torch.randn(B, 3, 32, 32, device="cuda"); y = torch.randint(0, 10, (B,), device="cuda")
    for _ in range(5):
        m.train_step_fused(x, y); opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train_step_fused(x, y); opt.step()
    t_enq = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / steps
    # one step alone: GPU time between events (host far ahead is impossible here: the queue is empty at the start)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gpu = []
    for _ in range(5):
        torch.cuda.synchronize(); e0.record()
        m.train_step_fused(x, y); opt.step()
        e1.record(); torch.cuda.synchronize(); gpu.append(e0.elapsed_time(e1))
    return t_all * 1e3, t_enq * 1e3, min(gpu)

def c4_config(B):
    c = bench.c3_config(B)
    c["hyperparameters"]["som"]["map_size"] = [4, 4]
    c["data"]["num_classes"] = 100
    return c

from vit_som_amd.tuning import hooks
run(64, bench.c3_config, steps=10)          # the first model of a process pays one-time costs (code objects, allocator): not measured
for name, fn, Bs in (("c3 (40x40 SOM, clustering)", bench.c3_config, (64, 128, 256, 512)), ("c4 (4x4 SOM, 100 classes)", c4_config, (128,))):
    for B in Bs:
        for tape in (True, False):
            hooks.set(launch_tape=tape)
            ms, enq, alone = run(B, fn)
            print(f"{name:28s} batch {B:4d} launch tape {'on ' if tape else 'off'}: {ms:6.2f} ms/step = {B / ms * 1e3:8.0f} img/s | host enqueue {enq:5.2f} ms/step | "
                  f"one step alone (events) {alone:5.2f} ms", flush=True)
hooks.reset()
