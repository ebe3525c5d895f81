"""How much do the step's GEMM shapes lose to the partial last round of workgroups?  Times the
forward Linear at M = 33280 (the c3 token count, 260 row tiles) against M = 32768 (256 row tiles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops

def t_ms(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for mode in (ops.GEMM_SPLIT_BF16, ops.GEMM_F32):
    ops.set_gemm_mode(mode)
    for N, K in ((192, 192), (576, 192), (768, 192), (192, 768)):
        row = []
        for M in (32768, 33280, 2 * 32768, 3 * 33280):
            x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
            y = torch.empty(M, N, device="cuda")
            ms = t_ms(lambda: ops.linear_fwd(x, W, b, y))
            row.append(f"M={M}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF")
        print(("x6 " if mode else "f32"), f"N={N} K={K} | " + " | ".join(row))
