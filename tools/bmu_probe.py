"""The BMU distance pass alone at the c3 shape: exact-f32 MFMA pass (round 1) vs the three-product bf16 contraction +
exact re-rank (round 2); dots kernel by HIP events, whole pass (norms + dots + finalize) by wall clock."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
dev = "cuda"
B, Kp, L = 512, 1600, 12288
X = torch.randn(B, L, device=dev); W = torch.nn.functional.normalize(torch.rand(Kp, L, device=dev), dim=1)
inx = torch.ones(B, device=dev); inw = torch.ones(Kp, device=dev)
dist = torch.empty(B, Kp, device=dev); bmu = torch.empty(B, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)

def f32_pass():
    ops.row_inv_norm(X, inx); ops.row_inv_norm(W, inw)
    ops.bmu_cosine_fwd(X, W, inx, inw, dist, bmu)

def x3_pass():
    ops.bmu_cosine_x3_fwd(X, W, dist, bmu, inx, inw, cnt)

for name, f in (("f32 MFMA pass (norms + dots + finalize)", f32_pass), ("x3 + re-rank pass (dots + norms/finalize)", x3_pass)):
    for _ in range(3): f()
    ops.enable_timer("bmu_cosine_dots")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20
    ms, n = ops.timer_ms("bmu_cosine_dots")
    ops.disable_timers()
    print(f"{name}: dots kernel {ms*1e3:7.1f} us  {2.0*B*Kp*L/ms/1e9:6.1f} TF f32-eq  ({4.0*(B*L+Kp*L+B*Kp)/ms/1e6:7.1f} GB/s algorithmic); "
          f"whole pass {wall*1e6:7.1f} us", flush=True)
print("rows re-ranked per pass:", int(cnt) / 23)
