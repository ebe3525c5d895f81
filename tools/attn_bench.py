import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
dev = "cuda"
def timeit(f, n=10):
    for _ in range(3): f()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [f() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[2]
for (B, N, H, hd) in [(512, 65, 3, 64), (512, 65, 3, 32), (128, 197, 2, 8), (256, 257, 3, 64)]:
    E = H * hd
    qkv = torch.randn(B, N, 3 * E, device=dev); out = torch.empty(B, N, E, device=dev); lse = torch.empty(B, H, N, device=dev)
    dout = torch.randn(B, N, E, device=dev); dqkv = torch.empty_like(qkv); delta = torch.empty(B, H, N, device=dev)
    tf = timeit(lambda: ops.attention_fwd(qkv, out, lse, B, N, H, hd))
    tb = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, dqkv, delta, B, N, H, hd))
    fl = 4.0 * B * H * N * N * hd
    print(f"B={B} N={N} H={H} hd={hd}: fwd {tf*1e3:7.1f} us ({fl/tf/1e9:5.1f} TF useful)  bwd {tb*1e3:7.1f} us ({3.5*fl/tb/1e9:5.1f} TF useful, 7 products)", flush=True)
