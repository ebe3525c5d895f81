"""Weight-gradient GEMM with six products (exact three-piece split) against three (two-piece RNE split): time alone at the
c3 shapes and relative Frobenius error against fp64, on activation-like (N(0,1)) and gradient-like (N(0,1) * 1e-6, wide
per-row scales) operands.  (vsom_set_gemm_mode: VSOM_GEMM_SPLIT_BF16 vs VSOM_GEMM_SPLIT_BF16_GRAD3.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
T, E, H4 = 512 * 65, 192, 768
dev = "cuda"
torch.manual_seed(0)
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rowscale = torch.exp(torch.randn(T, 1, device=dev) * 2.0)
for name, N, K in (("qkv", 576, 192), ("proj", 192, 192), ("fc1", 768, 192), ("fc2", 192, 768)):
    dY = torch.randn(T, N, device=dev) * 1e-6 * rowscale
    X = torch.randn(T, K, device=dev)
    ref = dY.double().t() @ X.double()
    dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    out = []
    for prods in (6, 3):
        ops.set_gemm_mode(ops.GEMM_SPLIT_BF16 if prods == 6 else ops.GEMM_SPLIT_BF16_GRAD3)
        ops.linear_bwd_weight(dY, X, dW, db); torch.cuda.synchronize()
        err = float((dW.double() - ref).norm() / ref.norm())
        worst = float(((dW.double() - ref).abs() / (ref.abs() + 1e-30 * ref.abs().max())).median())
        out.append((t_us(lambda: ops.linear_bwd_weight(dY, X, dW, db)), err, worst))
    print(f"dW {name:5s} [{N},{K}] over T: six products {out[0][0]:6.1f} us rel err {out[0][1]:.2e} (median elementwise {out[0][2]:.1e}) | "
          f"three products {out[1][0]:6.1f} us rel err {out[1][1]:.2e} (median elementwise {out[1][2]:.1e})", flush=True)
