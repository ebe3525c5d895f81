"""Does running the step's main chain on a HIGH-priority stream (the weight-gradient / second-chain stream stays normal)
change the step time?  usage: prio_probe.py [steps] [rounds]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vit_som_amd import ViTSOM

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = bench.c3_config(512)
torch.manual_seed(0)
model = ViTSOM(cfg, device="cuda")
model.set_schedule(50000, 10000)
(opt,), _ = model.configure_optimizers()
x = torch.rand(512, 3, 32, 32, device="cuda"); y = torch.randint(0, 10, (512,), device="cuda")
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (least, greatest):", lo, hi)
hi_stream = torch.cuda.Stream(priority=-1)

def run(n, st):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if st is None:
        for _ in range(n):
            model.train_step_fused(x, y); opt.step()
    else:
        with torch.cuda.stream(st):
            for _ in range(n):
                model.train_step_fused(x, y); opt.step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

res = {"default": [], "high": []}
run(3, None); run(3, hi_stream)
for r in range(rounds):
    res["default"].append(run(steps, None))
    res["high"].append(run(steps, hi_stream))
for k, v in res.items():
    print(f"main chain on the {k}-priority stream: " + " ".join(f"{t:.2f}" for t in v) + f"  min {min(v):.2f} ms/step")
