"""The twelve GEMMs of one encoder layer at the c3 shapes, each alone: time, f32-equivalent TFLOP/s, HBM GB/s,
and the two floors (bytes at 5 TB/s; the GEMM's bf16 products -- six in the forward, three in the gradient GEMMs of the default
mode -- at 2500 TFLOP/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
T, E, H4 = 512 * 65, 192, 768
dev = "cuda"
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
r = lambda *s: torch.randn(*s, device=dev) * 0.05
x, x4 = r(T, E), r(T, H4)
Wqkv, Wp, W1, W2 = r(3 * E, E), r(E, E), r(H4, E), r(E, H4)
b3, b1, b4 = r(3 * E), r(E), r(H4)
y3, y1, ya, yg, res = r(T, 3 * E), r(T, E), r(T, H4), r(T, H4), r(T, E)
dW = {k: torch.empty_like(v) for k, v in dict(qkv=Wqkv, p=Wp, f1=W1, f2=W2).items()}
db = {k: torch.empty(v.shape[0], device=dev) for k, v in dict(qkv=Wqkv, p=Wp, f1=W1, f2=W2).items()}
Wt = {k: v.t().contiguous() for k, v in dict(qkv=Wqkv, p=Wp, f1=W1, f2=W2).items()}
rows = [
    ("fwd qkv   [T,192]x[576,192]", lambda: ops.linear_fwd(x, Wqkv, b3, y3), T, 576, 192, 4 * (T * 192 + T * 576)),
    ("fwd proj  +residual", lambda: ops.linear_residual_fwd(x, Wp, b1, res, T, y1), T, 192, 192, 4 * (3 * T * 192)),
    ("fwd fc1   +gelu (2 outputs)", lambda: ops.linear_gelu_fwd(x, W1, b4, yg, ya), T, 768, 192, 4 * (T * 192 + 2 * T * 768)),
    ("fwd fc2   +residual", lambda: ops.linear_residual_fwd(x4, W2, b1, res, T, y1), T, 192, 768, 4 * (T * 768 + 2 * T * 192)),
    ("dX fc2    x gelu'", lambda: ops.linear_bwd_input_t(y1, Wt["f2"], ya, gelu_grad=yg), T, 768, 192, 4 * (T * 192 + 2 * T * 768)),
    ("dX fc1", lambda: ops.linear_bwd_input_t(x4, Wt["f1"], y1), T, 192, 768, 4 * (T * 768 + T * 192)),
    ("dX proj", lambda: ops.linear_bwd_input_t(x, Wt["p"], y1), T, 192, 192, 4 * (2 * T * 192)),
    ("dX qkv", lambda: ops.linear_bwd_input_t(y3, Wt["qkv"], y1), T, 192, 576, 4 * (T * 576 + T * 192)),
    ("dW fc2    [192,768] over T", lambda: ops.linear_bwd_weight(y1, x4, dW["f2"], db["f2"]), 192, 768, T, 4 * (T * 192 + T * 768)),
    ("dW fc1    [768,192] over T", lambda: ops.linear_bwd_weight(x4, x, dW["f1"], db["f1"]), 768, 192, T, 4 * (T * 192 + T * 768)),
    ("dW proj   [192,192] over T", lambda: ops.linear_bwd_weight(y1, x, dW["p"], db["p"]), 192, 192, T, 4 * (2 * T * 192)),
    ("dW qkv    [576,192] over T", lambda: ops.linear_bwd_weight(y3, x, dW["qkv"], db["qkv"]), 576, 192, T, 4 * (T * 576 + T * 192)),
]
tot = 0.0
grad3 = ops.get_gemm_mode() == ops.GEMM_SPLIT_BF16_GRAD3
for name, fn, M, N, K, nbytes in rows:
    us = t_us(fn)
    tot += us
    fl = 2.0 * M * N * K
    prod = 3 if (grad3 and not name.startswith("fwd")) else 6
    print(f"{name:30s} {us:7.1f} us  {fl/us/1e6:6.1f} TF  {nbytes/us/1e3:7.1f} GB/s   floors: hbm {nbytes/5e6:6.1f} us, mfma {prod*fl/2.5e9:6.1f} us   products {prod}")
print(f"sum {tot:.0f} us per layer (x12 encoder layers = {tot*12/1e3:.2f} ms)")
