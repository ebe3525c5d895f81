"""End-to-end health check: 1500 training steps of the c3 model on a fixed synthetic set of 8 batches (so the
losses CAN go down), LambdaLR stepped per 'epoch' of 8 batches; prints the loss trajectory and checks that
everything stays finite and that the total loss decreases."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vit_som_amd import ViTSOM
cfg = bench.c3_config(512)
cfg["hyperparameters"]["total_epochs"] = 200
torch.manual_seed(0)
m = ViTSOM(cfg, device="cuda")
m.set_schedule(8 * 512, 8 * 200)
(opt,), (sched,) = m.configure_optimizers()
g = torch.Generator().manual_seed(1)
# smooth images (low-frequency patterns) rather than white noise: something an autoencoder can learn
base = torch.randn(8, 512, 3, 4, 4, generator=g)
data = [torch.nn.functional.interpolate(b, size=32, mode="bilinear", align_corners=False).cuda() for b in base]
y = torch.zeros(512, dtype=torch.int64, device="cuda")
hist = []
for step in range(1500):
    loss = m.train_step_fused(data[step % 8], y)
    opt.step()
    if step % 8 == 7:
        sched.step()
    if step % 100 == 0 or step == 1499:
        v = float(loss); hist.append(v)
        print(f"step {step:5d}  loss {v:.5f}  recon+cls {float(m._last['main']):.5f}  som {float(m._last['som']):.5f}  T {float(m._last['T']):.3f}  lr {opt.param_groups[0]['lr']:.2e}", flush=True)
assert all(math.isfinite(v) for v in hist), "non-finite loss"
assert bool(torch.isfinite(m.arena.params).all()), "non-finite parameter"
assert hist[-1] < 0.6 * hist[0], (hist[0], hist[-1])
print("ok: loss %.4f -> %.4f" % (hist[0], hist[-1]))
