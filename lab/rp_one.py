"""One GEMM of the row-panel engine in a loop (for rocprofv3 --pmc runs):  python lab/rp_one.py qkv|fc2|proj [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T, E, H4 = int(os.environ.get("RP_T", 512 * 65)), 192, 768
dev = "cuda"
torch.manual_seed(0)
shapes = {"qkv": (E, 3 * E, 0), "proj": (E, E, 0), "fc1": (E, H4, 0), "fc2": (H4, E, 0), "dxfc2": (E, H4, 1), "dxqkv": (3 * E, E, 1)}
K, N, tr = shapes[which]
X = torch.randn(T, K, device=dev)
W = torch.randn((K, N) if tr else (N, K), device=dev) * 0.05      # nn.Linear weight [out, in]; the input gradient reduces over `out`
kind = ops.weight_image_kind(N, K)
img = torch.empty(ops.weight_image_bytes(N, K), dtype=torch.uint8, device=dev)
table = torch.tensor([[0, 0, N, K, tr, kind]], dtype=torch.int64, device=dev)
ops.weight_images_prepare(W.view(-1), img, table, ((N + 31) // 32) * ((K + 31) // 32 * 2))
bias = torch.randn(N, device=dev)
out = torch.empty(T, N, device=dev)
for _ in range(3): ops.linear_planes(X, img, bias, out, epilogue=ops.EPI_BIAS)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): ops.linear_planes(X, img, bias, out, epilogue=ops.EPI_BIAS)
e1.record(); torch.cuda.synchronize()
print(which, "kind", kind, f"{e0.elapsed_time(e1) / iters * 1e3:.1f} us")
