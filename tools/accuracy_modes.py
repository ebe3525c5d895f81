"""Accuracy of the training step at real CIFAR-10 layer shapes (E=192, 12+2 layers, N=65, L=12288, 40x40 map)
in both GEMM modes, against the fp64 evaluation of the CPU oracle: loss and the worst / median relative
gradient error over all parameter tensors.  (Oracle = tests-only; this is a measurement aid, not product.)"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import vitsom_oracle as O
import vit_som_amd
from vit_som_amd import ops

B = 32
cfg = bench.c3_config(B)
d = O.Dims(cfg)
P = O.init_params(cfg, seed=3)
x, y = O.synthetic_batch(d, B, seed=1)
it, n_train, est = 1000, 50000, 9700
P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
torch.set_num_threads(os.cpu_count())
total64, parts64, G64 = O.loss_and_grads(P64, x.double(), y, d, it, n_train, est)
print(f"fp64 oracle loss {float(total64):.9f}")
for mode, name in ((ops.GEMM_SPLIT_BF16_GRAD3, "split-bf16, gradient GEMMs 3 products (default)"), (ops.GEMM_SPLIT_BF16, "split-bf16, 6 products everywhere              "),
                   (ops.GEMM_F32, "f32 MFMA                                       ")):
    ops.set_gemm_mode(mode)
    m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device="cuda")
    m.load_state_dict({k: v for k, v in P.items()}, strict=False)
    m._it = it; m.set_schedule(n_train, est)
    loss = m.train_step_fused(x.cuda(), y.cuda())
    errs = []
    for n, _ in m.named_parameters():
        if n in G64:
            g = m._grad_views[n].double().cpu(); r = G64[n]
            errs.append(float((g - r).norm() / (r.norm() + 1e-300)))
    errs.sort()
    print(f"{name}: loss {float(loss):.9f} (diff {abs(float(loss)-float(total64)):.2e})  grad rel err: median {errs[len(errs)//2]:.2e}  worst {errs[-1]:.2e}  over {len(errs)} tensors; bmu equal fp64: {bool(torch.equal(m._ctx[2].bmu.cpu(), parts64['bmu']))}")
