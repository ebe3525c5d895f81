"""One Linear-shaped GEMM through the C-ABI, 30 launches (for rocprofv3 --pmc runs): gemm_one.py M N K [fwd|res|dw]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "fwd"
dev = "cuda"
r = lambda *s: torch.randn(*s, device=dev) * 0.05
x, W, b, y, res = r(M, K), r(N, K), r(N), r(M, N), r(M, N)
dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
fn = {"fwd": lambda: ops.linear_fwd(x, W, b, y), "res": lambda: ops.linear_residual_fwd(x, W, b, res, M, y),
      "dw": lambda: ops.linear_bwd_weight(y, x, dW, db)}[kind]
for _ in range(30): fn()
torch.cuda.synchronize()
