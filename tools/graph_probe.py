"""Feasibility / timing probe: the whole training step (three HIP streams, ~400 launches) captured into ONE hipGraph
through torch.cuda.CUDAGraph and replayed (per-step scalars frozen at capture time: timing only)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vit_som_amd import ViTSOM

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
m = ViTSOM(bench.c3_config(B), device="cuda")
m.set_schedule(50000, 48500); m._it = 1000
(opt,), _ = m.configure_optimizers()
x = torch.randn(B, 3, 32, 32, device="cuda"); y = torch.zeros(B, dtype=torch.int64, device="cuda")

def step():
    m.train_step_fused(x, y)
    opt.step()

def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3

for _ in range(5): step()
print("eager: host enqueue %.2f ms/step, wall %.2f ms/step" % timeit(step))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()          # warm-up on the capture stream (scratch buffers keyed by stream)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step()
torch.cuda.synchronize()
print("captured")
for _ in range(3): g.replay()
print("graph: host enqueue %.3f ms/step, wall %.2f ms/step" % timeit(g.replay))

def single(fn, n=8):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    ts.sort()
    return ts[len(ts) // 2]
print("one step from an idle GPU, eager : host %.2f ms, until done %.2f ms" % single(step))
print("one step from an idle GPU, graph : host %.2f ms, until done %.2f ms" % single(g.replay))
