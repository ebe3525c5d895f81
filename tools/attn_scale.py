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
N, H, hd = 65, 3, 64
for B in (16, 43, 86, 128, 171, 256, 342, 512, 1024):
    E = H * hd
    qkv = torch.randn(B, N, 3 * E, device=dev); out = torch.empty(B, N, E, device=dev); lse = torch.empty(B, H, N, device=dev)
    tf = timeit(lambda: ops.attention_fwd(qkv, out, lse, B, N, H, hd))
    print(f"B={B:5d} blocks={B*H:5d} ({B*H/256:4.2f}/CU): fwd {tf*1e3:7.1f} us", flush=True)
