"""A/B timing of the whole training step inside ONE process (box-to-box and clock drift cancel):
alternates one test hook (vit_som_amd/tuning.py) between blocks of steps.  usage: ab_step.py hook=a,b [steps] [rounds]
e.g. ab_step.py fwd_split_blocks=0,6,12   ab_step.py side_stream=0,1   ab_step.py attn_fused=0,1"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vit_som_amd import ViTSOM, ops
from vit_som_amd.tuning import hooks


def apply(key, v):
    if key == "attn_fused":
        ops.set_attention_fused(int(v))
    elif key == "gemm_mode":                         # 0 f32 MFMA, 1 six products everywhere, 2 (default) three in the gradient GEMMs
        ops.set_gemm_mode(int(v))
    else:
        hooks.set(**{key: (None if v == "None" else int(v))})

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
    apply(key, v); run(3)
for r in range(rounds):
    for v in vals:
        apply(key, v)
        run(2)
        res[v].append(run(steps))
for v in vals:
    print(f"{key}={v}: " + " ".join(f"{t:.2f}" for t in res[v]) + f"  min {min(res[v]):.2f} ms/step")
