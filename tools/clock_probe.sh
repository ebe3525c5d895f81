#!/bin/bash
# Reported shader clock / power while a workload loops.  usage: clock_probe.sh [bmu|step]
cd "$(dirname "$0")/.."
MODE=${1:-bmu}
python - "$MODE" <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
mode = sys.argv[1]
if mode == "bmu":
    from vit_som_amd import ops
    B, K, L = 512, 1600, 12288
    X = torch.randn(B, L, device="cuda"); W = torch.randn(K, L, device="cuda"); inx = torch.ones(B, device="cuda"); inw = torch.ones(K, device="cuda")
    dist = torch.empty(B, K, device="cuda"); bmu = torch.empty(B, dtype=torch.int64, device="cuda")
    work = lambda: [ops.bmu_cosine_fwd(X, W, inx, inw, dist, bmu) for _ in range(200)]
else:
    import bench
    from vit_som_amd import ViTSOM
    model = ViTSOM(bench.c3_config(512), device="cuda"); model.set_schedule(50000, 10000)
    (opt,), _ = model.configure_optimizers()
    x = torch.randn(512, 3, 32, 32, device="cuda"); y = torch.zeros(512, dtype=torch.int64, device="cuda")
    def work():
        for _ in range(20):
            model.train_step_fused(x, y); opt.step()
t0 = time.time()
while time.time() - t0 < 14:
    work(); torch.cuda.synchronize()
PY
PID=$!
sleep 7
for i in 1 2 3; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -iE "sclk|power|junction|Temperature \(Sensor (edge|junction)" | head -6
  echo "--"
  sleep 1.5
done
wait $PID
