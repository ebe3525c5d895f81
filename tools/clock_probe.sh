#!/bin/bash
# Shader clock while (a) a pure-MFMA loop and (b) the BMU distance pass run back to back.
cd "$(dirname "$0")/.."
python - <<'PY' &
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from vit_som_amd import ops
B, K, L = 512, 1600, 12288
X = torch.randn(B, L, device="cuda"); W = torch.randn(K, L, device="cuda"); inx = torch.ones(B, device="cuda"); inw = torch.ones(K, device="cuda")
dist = torch.empty(B, K, device="cuda"); bmu = torch.empty(B, dtype=torch.int64, device="cuda")
t0 = time.time()
while time.time() - t0 < 14:
    for _ in range(200): ops.bmu_cosine_fwd(X, W, inx, inw, dist, bmu)
    torch.cuda.synchronize()
PY
PID=$!
sleep 6
for i in 1 2 3 4; do
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -2
  rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1
  sleep 1.5
done
wait $PID
echo "--- idle"
rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1
