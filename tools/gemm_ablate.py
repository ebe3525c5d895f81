import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
dev = "cuda"; T = 33280
def bench(M, N, K):
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    f = lambda: ops.linear_fwd(x, W, b, y)
    for _ in range(3): f()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [f() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    print(f"dbg={os.environ.get('VSOM_GEMM_DBG','0')} M={M} N={N} K={K}: {sorted(ts)[2]*1e3:8.1f} us", flush=True)
for (N, K) in ((576, 32), (576, 192), (192, 768), (768, 192)):
    bench(T, N, K)
