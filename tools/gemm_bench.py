"""Micro-benchmark of the f32-MFMA GEMM entries by shape (interleaved rounds, HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops

dev = "cuda"
T = 33280
shapes = [  # (kind, M, N, K)
    ("fwd", T, 576, 192), ("fwd", T, 192, 192), ("fwd_gelu", T, 768, 192), ("fwd_res", T, 192, 768),
    ("bwd_in", T, 576, 192), ("bwd_in", T, 768, 192), ("bwd_in_gelu", T, 192, 768), ("bwd_w", T, 576, 192),
    ("bwd_w", T, 768, 192), ("bwd_w", T, 192, 768), ("fwd", 4096, 4096, 4096), ("bwd_in", 4096, 4096, 4096),
    ("fwd", 8192, 1024, 1024),
]
def make(kind, M, N, K):
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); y2 = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev)
    dx = torch.empty(M, K, device=dev); gg = torch.randn(M, K, device=dev); dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    R = torch.randn(M, N, device=dev)
    if kind == "fwd": return lambda: ops.linear_fwd(x, W, b, y)
    if kind == "fwd_gelu": return lambda: ops.linear_gelu_fwd(x, W, b, y, y2)
    if kind == "fwd_res": return lambda: ops.linear_residual_fwd(x, W, b, R, M, y)
    if kind == "bwd_in": return lambda: ops.linear_bwd_input(dy, W, dx)
    if kind == "bwd_in_gelu": return lambda: ops.linear_bwd_input(dy, W, dx, gelu_grad=gg)
    if kind == "bwd_w": return lambda: ops.linear_bwd_weight(dy, x, dW, db)
fns = [(s, make(*s)) for s in shapes]
for _, f in fns: f()
torch.cuda.synchronize()
res = {s: [] for s, _ in fns}
for r in range(5):
    for s, f in fns:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): f()
        e1.record(); torch.cuda.synchronize()
        res[s].append(e0.elapsed_time(e1) / 3)
for s, _ in fns:
    kind, M, N, K = s
    ms = sorted(res[s])[len(res[s]) // 2]
    print(f"{kind:12s} M={M:6d} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
