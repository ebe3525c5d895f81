import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
dev = "cuda"
def timeit(f, n=5):
    for _ in range(2): f()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [f() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[2]
for (M, N, K) in [(33280, 576, 192), (33280, 768, 192), (33280, 192, 768), (33280, 192, 192), (33280, 1024, 1024), (4096, 4096, 4096), (33280*4, 768, 192), (8192, 2048, 2048)]:
    dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev); dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    ms = timeit(lambda: ops.linear_bwd_weight(dy, x, dW, db))
    print(f"splits_target={os.environ.get('VSOM_TN_BLOCKS','512')} bwd_w M={M} N={N} K={K}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF", flush=True)
