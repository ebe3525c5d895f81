import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
dev = "cuda"; T = 33280
def timeit(f, n=5):
    for _ in range(2): f()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [f() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[2]
def bench(kind, M, N, K):
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); y2 = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev)
    gg = torch.randn(M, K, device=dev); dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev); R = torch.randn(M, N, device=dev)
    f = {"fwd": lambda: ops.linear_fwd(x, W, b, y), "fwd_gelu": lambda: ops.linear_gelu_fwd(x, W, b, y, y2),
         "fwd_res": lambda: ops.linear_residual_fwd(x, W, b, R, M, y), "bwd_in": lambda: ops.linear_bwd_input(dy, W, dx),
         "bwd_in_gelu": lambda: ops.linear_bwd_input(dy, W, dx, gelu_grad=gg), "bwd_w": lambda: ops.linear_bwd_weight(dy, x, dW, db)}[kind]
    ms = timeit(f)
    print(f"cfg={os.environ.get('VSOM_GEMM_CFG','auto')} {kind:12s} M={M} N={N} K={K}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF", flush=True)
for s in [("fwd", T, 576, 192), ("fwd", T, 192, 192), ("fwd_gelu", T, 768, 192), ("fwd_res", T, 192, 768), ("bwd_in", T, 576, 192), ("bwd_in", T, 768, 192),
          ("bwd_in", T, 192, 192), ("bwd_in_gelu", T, 192, 768), ("bwd_w", T, 576, 192), ("bwd_w", T, 768, 192), ("bwd_w", T, 192, 768), ("bwd_w", T, 192, 192),
          ("fwd", T, 96, 192), ("fwd", T, 288, 96), ("fwd", T, 384, 96), ("fwd", T, 48, 96), ("fwd", 4096, 4096, 4096)]:
    bench(*s)
# SOM shapes
B, Kp, L = 512, 1600, 12288
X = torch.randn(B, L, device=dev); W = torch.randn(Kp, L, device=dev); inx = torch.ones(B, device=dev); inw = torch.ones(Kp, device=dev)
dist = torch.empty(B, Kp, device=dev); bmu = torch.empty(B, dtype=torch.int64, device=dev)
ms = timeit(lambda: ops.bmu_cosine_fwd(X, W, inx, inw, dist, bmu)); print(f"cfg={os.environ.get('VSOM_GEMM_CFG','auto')} bmu_fwd(total): {ms*1e3:8.1f} us {2.0*B*Kp*L/ms/1e9:6.1f} TF", flush=True)
coef = torch.randn(B, Kp, device=dev); rd = torch.randn(B, device=dev); cd = torch.randn(Kp, device=dev); gW = torch.empty(Kp, L, device=dev); gX = torch.zeros(B, L, device=dev)
ms = timeit(lambda: ops.som_bwd(X, W, coef, rd, cd, gW, gX, True)); print(f"cfg={os.environ.get('VSOM_GEMM_CFG','auto')} som_bwd(gW+gX): {ms*1e3:8.1f} us {4.0*B*Kp*L/ms/1e9:6.1f} TF", flush=True)
