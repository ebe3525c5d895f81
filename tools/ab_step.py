"""A/B timing of the whole training step inside ONE process (box-to-box and clock drift cancel):
alternates environment settings between blocks of steps.  usage: ab_step.py KEY=a,b [steps] [rounds]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vit_som_amd import ViTSOM

key, vals = sys.argv[1].split("=")
vals = vals.split(",")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = bench.c3_config(512)
torch.manual_seed(0)
model = ViTSOM(cfg, device="cuda")
model.set_schedule(50000, 10000)
(opt,), _ = model.configure_optimizers()
x = torch.rand(512, 3, 32, 32, device="cuda"); y = torch.randint(0, 10, (512,), device="cuda")

def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        model.train_step_fused(x, y); opt.step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

res = {v: [] for v in vals}
for v in vals:
    os.environ[key] = v; run(3)
for r in range(rounds):
    for v in vals:
        os.environ[key] = v
        run(2)
        res[v].append(run(steps))
for v in vals:
    print(f"{key}={v}: " + " ".join(f"{t:.2f}" for t in res[v]) + f"  min {min(res[v]):.2f} ms/step")
