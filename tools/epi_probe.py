"""What the epilogue variants cost at the step's shapes: the same GEMM with bias only, with bias + residual, with the
two GELU outputs."""
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
W1, W2, Wp = r(H4, E), r(E, H4), r(E, E)
b1, b4 = r(E), r(H4)
y1, ya, yg, res = r(T, E), r(T, H4), r(T, H4), r(T, E)
for name, fn in [
    ("fc2 shape, bias only          ", lambda: ops.linear_fwd(x4, W2, b1, y1)),
    ("fc2 shape, bias + residual    ", lambda: ops.linear_residual_fwd(x4, W2, b1, res, T, y1)),
    ("proj shape, bias only         ", lambda: ops.linear_fwd(x, Wp, b1, y1)),
    ("proj shape, bias + residual   ", lambda: ops.linear_residual_fwd(x, Wp, b1, res, T, y1)),
    ("fc1 shape, bias only (1 out)  ", lambda: ops.linear_fwd(x, W1, b4, ya)),
    ("fc1 shape, bias + gelu (2 out)", lambda: ops.linear_gelu_fwd(x, W1, b4, yg, ya)),
]:
    print(f"{name} {t_us(fn):7.1f} us")
