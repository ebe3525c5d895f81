import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
dev = "cuda"
B, Kp, L = 512, 1600, 12288
X = torch.randn(B, L, device=dev); W = torch.randn(Kp, L, device=dev); inx = torch.ones(B, device=dev); inw = torch.ones(Kp, device=dev)
dist = torch.empty(B, Kp, device=dev); bmu = torch.empty(B, dtype=torch.int64, device=dev)
f = lambda: ops.bmu_cosine_fwd(X, W, inx, inw, dist, bmu)
for _ in range(3): f()
ops.enable_timer("bmu_cosine_dots")
for _ in range(20): f()
ms, n = ops.timer_ms("bmu_cosine_dots")
print(f"splits={os.environ.get('VSOM_BMU_SPLITS','model')} dots kernel {ms*1e3:7.1f} us  {2.0*B*Kp*L/ms/1e9:6.1f} TF  ({4.0*(B*L+Kp*L+B*Kp)/ms/1e6:7.1f} GB/s algorithmic)", flush=True)
