"""ViT-SOM training-step benchmark (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full optimisation step of the hot path on one synthetic batch already resident
in HBM: forward (ViT autoencoder + SOM BMU pass) + losses + backward + gradient all-reduce (RCCL,
N > 1) + fused AdamW.  Workload = BASELINE config c3: CIFAR-10 shapes, 40x40 SOM, per-GPU batch
512 (weak scaling: the reference's batch_size is per rank).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA peak (no xf32 on gfx950)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)


def c3_config(batch):
    return {
        "hyperparameters": {
            "model_arch": "vit_som", "total_epochs": 500, "batch_size": batch, "gamma": 0.01,
            "som": {"map_size": [40, 40], "Tmax": 4, "Tmin": 0.1, "distance_fcn": "cosine", "topology": "square",
                    "use_reduced": False},
            "vit": {"patch_size": 4, "emb_dim": 192, "depth": 12, "dec_emb_dim": 96, "dec_depth": 2, "heads": 3,
                    "mlp_ratio": 4, "qkv_bias": True, "qk_norm": False, "proj_drop": 0, "attn_drop": 0, "drop_path": 0.1,
                    "global_pool": False},
            "optimizer": {"type": "adamw", "lr": 0.0005, "min_lr": 0.000001, "beta_1": 0.9, "beta_2": 0.999,
                          "scheduler": "cosine_annealing", "warmup_epochs": 25, "weight_decay": 0.05, "layer_decay": 0.75,
                          "smoothing": 0.1},
        },
        "data": {"dataset": "synthetic-cifar10", "num_classes": 0, "num_channels": 3, "input_size": 32, "num_workers": 0},
    }


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_budget=25.0):
    """the CPU oracle (oracle/vitsom_oracle.py: fwd + bwd + AdamW in plain torch CPU ops, the
    restatement of the reference's step) timed on this box's host cores, at torch matmul precision 'highest' AND
    'medium' (the reference's default, vit_som.py:23), best of 5 steps after 2 warm-ups each -- on a BOUNDED sample of
    the workload: batch 64 instead of 512 (a batch-512 step takes ~3.5 s on 16 cores; the per-image cost is flat in
    the batch at these sizes)."""
    from oracle import vitsom_oracle as O
    Bc = 64
    cfg = c3_config(Bc)
    # one GPU's share of the host is 16 cores; more threads only oversubscribe them
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    n_train, est = 50000, 100000
    res, t_start = {}, time.perf_counter()
    for prec in ("highest", "medium"):
        torch.set_float32_matmul_precision(prec)
        step = O.CPUStep(cfg, seed=0)
        x, y = O.synthetic_batch(step.d, Bc, seed=0)
        for _ in range(2):
            step.step(x, y, n_train, est)
        best = float("inf")
        for _ in range(5):
            t0 = time.perf_counter()
            step.step(x, y, n_train, est)
            best = min(best, time.perf_counter() - t0)
            if time.perf_counter() - t_start > seconds_budget:
                break
        res[prec] = round(Bc / best, 2)
    torch.set_float32_matmul_precision("highest")
    # the bench's own batch once: one warm-up + two timed steps at batch 512 (about 3.5 s each on 16 cores)
    v512 = None
    if time.perf_counter() - t_start < seconds_budget:
        step = O.CPUStep(c3_config(512), seed=0)
        x, y = O.synthetic_batch(step.d, 512, seed=0)
        step.step(x, y, n_train, est)
        best = float("inf")
        for _ in range(2):
            t0 = time.perf_counter()
            step.step(x, y, n_train, est)
            best = min(best, time.perf_counter() - t0)
        v512 = round(512 / best, 2)
        del step, x, y
    return {"value": v512 if v512 is not None else res["highest"], "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": _cpu_model(), "value_batch_64": res["highest"], "value_batch_64_medium_precision": res["medium"],
            "sample": f"same c3 workload, fwd+bwd+AdamW, fp32 ('highest' matmul precision).  'value': the bench's own batch 512, best of 2 "
                      f"steps after 1 warm-up" + ("" if v512 is not None else " -- NOT MEASURED (time budget), the batch-64 figure stands in") +
                      f"; 'value_batch_{Bc}' / '..._medium_precision' (the reference's default "
                      f"torch.set_float32_matmul_precision('medium')): batch {Bc}, best of 5 steps after 2 warm-ups per precision",
            "seconds": round(time.perf_counter() - t_start, 1)}


def secondary_kernels(ops, B, dev):
    """SURVEY.md 8(d) also asks for the MFMA utilisation of the attention / Linear GEMMs: time the
    encoder's qkv Linear and the attention forward on their own (outside the timed region), with
    the peak each is quoted against."""
    T, E, H, N = B * 65, 192, 3, 65
    x = torch.randn(T, E, device=dev); W = torch.randn(3 * E, E, device=dev) * 0.05; b = torch.zeros(3 * E, device=dev)
    qkv = torch.empty(T, 3 * E, device=dev); ao = torch.empty(T, E, device=dev); lse = torch.empty(B * H * N, device=dev)

    def ms(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    t_lin = ms(lambda: ops.linear_fwd(x, W, b, qkv))
    t_att = ms(lambda: ops.attention_fwd(qkv, ao, lse, B, N, H, E // H))
    dao = torch.randn(T, E, device=dev); dqkv = torch.empty(T, 3 * E, device=dev); delta = torch.empty(B * H * N, device=dev)
    t_attb = ms(lambda: ops.attention_bwd(qkv, ao, dao, lse, dqkv, delta, B, N, H, E // H))
    dW = torch.empty(3 * E, E, device=dev); db = torch.empty(3 * E, device=dev)
    dY = torch.randn(T, 3 * E, device=dev)
    t_dw = ms(lambda: ops.linear_bwd_weight(dY, x, dW, db))
    split = ops.get_gemm_mode() != ops.GEMM_F32
    grad3 = ops.get_gemm_mode() == ops.GEMM_SPLIT_BF16_GRAD3
    lin_tf = 2.0 * T * 3 * E * E / (t_lin * 1e-3) / 1e12
    att_tf = 4.0 * B * H * N * N * (E // H) / (t_att * 1e-3) / 1e12
    attb_tf = 10.0 * B * H * N * N * (E // H) / (t_attb * 1e-3) / 1e12        # five N x N x hd products
    dw_tf = 2.0 * T * 3 * E * E / (t_dw * 1e-3) / 1e12
    lin_peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if split else F32_MFMA_PEAK_TFLOPS
    return {
        "linear_qkv_fwd": {"shape": [T, 3 * E, E], "ms": round(t_lin, 4), "achieved_f32_equiv_TFLOPs": round(lin_tf, 1),
                           "peak_TFLOPs": round(lin_peak, 1), "frac": round(lin_tf / lin_peak, 3),
                           "peak_basis": ("dense bf16 MFMA 2500 TF / 6 products per fp32 product" if split else "f32 MFMA"),
                           "hbm_GBps": round(4.0 * (T * E + 3 * E * E + T * 3 * E) / (t_lin * 1e-3) / 1e9, 1)},
        "attention_fwd": {"shape": {"images": B, "heads": H, "tokens": N, "head_dim": E // H}, "ms": round(t_att, 4),
                          "achieved_TFLOPs": round(att_tf, 1), "peak_TFLOPs": F32_MFMA_PEAK_TFLOPS,
                          "frac": round(att_tf / F32_MFMA_PEAK_TFLOPS, 3), "peak_basis": "f32 MFMA (16x16x4)",
                          "hbm_GBps": round(4.0 * (T * 3 * E + T * E) / (t_att * 1e-3) / 1e9, 1)},
        "attention_bwd": {"shape": {"images": B, "heads": H, "tokens": N, "head_dim": E // H}, "ms": round(t_attb, 4),
                          "achieved_TFLOPs": round(attb_tf, 1), "peak_TFLOPs": F32_MFMA_PEAK_TFLOPS,
                          "frac": round(attb_tf / F32_MFMA_PEAK_TFLOPS, 3),
                          "peak_basis": ("quoted against the f32 MFMA peak (16x16x4) for continuity; in this mode the seven products run on bf16 "
                                         "MFMA from the two-piece split (matrix pipe busy ~4 of the ~62 us: the kernel is staging / VALU / latency "
                                         "bound, DESIGN.md section 4); 10 B H N^2 hd FLOP") if grad3 else "f32 MFMA (16x16x4); 10 B H N^2 hd FLOP",
                          "hbm_GBps": round(4.0 * (2 * T * 3 * E + 2 * T * E) / (t_attb * 1e-3) / 1e9, 1)},
        "linear_qkv_bwd_weight": {"shape": [3 * E, E, T], "ms": round(t_dw, 4), "achieved_f32_equiv_TFLOPs": round(dw_tf, 1),
                                  "peak_TFLOPs": round(lin_peak * (2.0 if grad3 else 1.0), 1),
                                  "frac": round(dw_tf / (lin_peak * (2.0 if grad3 else 1.0)), 3),
                                  "peak_basis": ("dense bf16 MFMA 2500 TF / 3 products per fp32 product (two-piece split)" if grad3 else
                                                 "dense bf16 MFMA 2500 TF / 6 products" if split else "f32 MFMA"),
                                  "note": "dW = dY^T X over the token rows incl. the fixed-order slab reduction and the bias gradient",
                                  "hbm_GBps": round(4.0 * (T * 3 * E + T * E + 3 * E * E) / (t_dw * 1e-3) / 1e9, 1)},
    }


def bmu_roofline(ops, B, world, bmu_ms, bmu_calls):
    """Roofline object of the dominant kernel of the BMU distance pass (the kernel BASELINE.json's metric names).
    achieved = SURVEY.md 8(d)'s algorithmic FLOPs (2 B L K) / the kernel's average launch time, measured live with
    HIP events on the stream it runs on; traffic = HBM bytes per launch from the tracked rocprofv3 PMC table."""
    K, L = 1600, 12288
    flops, nbytes = 2.0 * B * L * K, 4.0 * (B * L + K * L + B * K)       # 20.13 GFLOP, 107.1 MB at B = 512
    t_s = bmu_ms * 1e-3
    split = ops.get_gemm_mode() != ops.GEMM_F32
    # three bf16 products per fp32 product (two-piece split + exact re-rank) vs the exact-f32 MFMA engine
    peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if split else F32_MFMA_PEAK_TFLOPS
    from vit_som_amd.tuning import hooks
    planes = split and hooks.bmu_planes and ops.bmu_planes_supported(B, K, L)
    traffic = None
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r03_bmu_hbm_traffic.json")))
        if B == t["batch"] and world == 1 and split and ("planes" in t["kernel"]) == planes:
            traffic = t["read_bytes"] + t["write_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    if planes:
        kernel = ("bmu_x3_planes_kernel (BMU distance pass: X[B,L] . W[K,L]^T on bf16 MFMA, three products of a two-piece split; "
                  "operands pre-split into plane images by planes_kernel -- X ~12 us and W ~33 us per step, separate launches NOT "
                  "in this kernel's time -- fed by LDS-DMA; split over L; exact fp64 re-rank in bmu_x3_finalize_kernel)")
    elif split:
        kernel = ("bmu_x3_kernel<2,3,4,2> (BMU distance pass: X[B,L] . W[K,L]^T on bf16 MFMA from a two-piece split, three "
                  "products, row norms fused in, split over L; exact fp64 re-rank in bmu_x3_finalize_kernel)")
    else:
        kernel = "gemm_f32_kernel<true,true,1,2,4,1,6,true> (BMU distance pass on f32 MFMA)"
    return {"kernel": kernel,
            "bound": "mfma", "achieved": round(flops / t_s / 1e12, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(flops / t_s / 1e12 / peak, 4),
            "peak_basis": "dense bf16 MFMA 2500 TF / 3 products per fp32 product" if split else "dense f32 MFMA",
            "traffic": traffic, "avg_launch_ms": round(bmu_ms, 4), "launches_timed": bmu_calls,
            "algorithmic_flops": flops, "algorithmic_bytes": nbytes,
            "hbm_view": {"achieved_GBps": round(nbytes / t_s / 1e9, 1), "peak_GBps": HBM_PEAK_GBS,
                         "frac": round(nbytes / t_s / 1e9 / HBM_PEAK_GBS, 4)}}


def launch_ranks(args, script=None, argv=None) -> int:
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment), rendezvous on
    127.0.0.1.  Rank 0 inherits stdout, so its JSON line is this command's output."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] +
                                      (sys.argv[1:] if argv is None else list(argv)), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies must not leave the others waiting in the rendezvous / a collective until their time-out
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            rc = max(rc, abs(code))
        if rc and live:
            for p in live:                         # exactly the children started above
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            live = []
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch (BASELINE c3: 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher -- it starts one fresh child per
        # GPU (before making any GPU call itself) and relays rank 0's JSON line
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VSOM_DIST_BACKEND", "nccl")          # nccl == RCCL on ROCm; gloo only to rehearse
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # the CPU leg runs BEFORE anything touches the GPU (rank 0, N = 1 only)
    cpu_res = cpu_baseline() if (world == 1 and rank == 0 and not args.no_cpu_baseline) else None

    import vit_som_amd
    from vit_som_amd import ops

    B = args.batch
    cfg = c3_config(B)
    torch.manual_seed(0)                                # identical replicas on every rank
    model = vit_som_amd.ViTSOM(cfg, device=dev)
    model.set_distributed(world, rank)
    model.broadcast_parameters()                        # replicas are seed-identical; this makes it explicit (N > 1)
    n_train = 50000
    model.set_schedule(n_train, (n_train // (B * world)) * cfg["hyperparameters"]["total_epochs"])
    model._it = 1000                                    # mid-ramp: gamma_t > 0 so the SOM gradients are live
    (opt,), _ = model.configure_optimizers()
    g = torch.Generator().manual_seed(1234 + rank)      # a different shard of synthetic images per rank
    x = torch.randn(B, 3, 32, 32, generator=g).to(dev)
    y = torch.zeros(B, dtype=torch.int64, device=dev)

    def step():
        loss = model.train_step_fused(x, y)
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    # The BMU distance kernel inside the step, by HIP events on the stream it is launched on.  The timed region above runs
    # the product path -- the step's launches re-issued from C by the launch tape (model.py) -- where the host cannot place
    # events around one kernel; so the same steps go on for a few more iterations through the host-driven path (same
    # kernels, same buffers, same streams) with an event pair around every launch of the kernel.
    from vit_som_amd.tuning import hooks
    hooks.set(launch_tape=False)
    ops.enable_timer("bmu_cosine_dots")
    for _ in range(max(5, min(args.steps, 20))):
        step()
    barrier()
    bmu_ms, bmu_calls = ops.timer_ms("bmu_cosine_dots")
    ops.disable_timers()
    hooks.reset()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    final_loss = float(loss)

    if rank == 0:
        K, L = 1600, 12288
        bmu_bytes = 4.0 * (B * L + K * L + B * K)       # 107.1 MB at B=512
        bmu_flops = 2.0 * B * L * K                     # 20.13 GFLOP
        t_s = bmu_ms * 1e-3
        out = {
            "metric": "images/sec/node ViT-SOM 40x40 CIFAR-10 bs512 (training step: fwd+bwd+all-reduce+AdamW)",
            "value": round(world * B * args.steps / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            # fp32 storage / accumulation / results; the Linear GEMMs reach the bf16 matrix cores through an exact
            # 3-piece split of every fp32 operand (config.gemm_arithmetic), everything else is plain f32
            "dtype": ("f32" if ops.get_gemm_mode() == ops.GEMM_F32 else
                      "f32 (Linear GEMMs on bf16 MFMA: forward from an exact 3xbf16 operand split, six products; " +
                      ("weight / input gradients from a 2xbf16 split, three products, rel. error 4e-6)"
                       if ops.get_gemm_mode() == ops.GEMM_SPLIT_BF16_GRAD3 else "gradients likewise)")),
            "data": "synthetic",
            "config": {"workload": "c3: vit_som CIFAR-10 shapes (3x32x32, patch 4, E=192, 3 heads, depth 12 + 2-layer "
                                   "decoder), 40x40 cosine SOM on the flattened patch tokens (L=12288), clustering loss "
                                   "L1(recon)+gamma*SOM, AdamW, random-init weights",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "gemm_arithmetic": ("fp32 in/out; forward nn.Linear GEMMs on bf16 MFMA from an EXACT 3-piece bf16 split of "
                                           "each fp32 operand (6 products, fp32 accumulate; error <= fp32 MFMA's); " +
                                           ("their weight- and input-gradient GEMMs from a 2-piece round-to-nearest split (3 "
                                            "products; gradients of the whole step within 1.3e-5 relative of fp64, bar 1e-4); "
                                            if ops.get_gemm_mode() == ops.GEMM_SPLIT_BF16_GRAD3 else "gradient GEMMs likewise; ") +
                                           "BMU distances from a 2-piece split (3 products, |err| <= 4.6e-5) with an exact "
                                           "fp64 re-rank of the near-minimum prototypes") if ops.get_gemm_mode() != ops.GEMM_F32
                                          else "f32 MFMA everywhere",
                       "final_loss": round(final_loss, 6)},
            "roofline": bmu_roofline(ops, B, world, bmu_ms, bmu_calls),
        }
        out["secondary"] = secondary_kernels(ops, B, dev)
        if cpu_res is not None:
            out["cpu_baseline"] = cpu_res
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
