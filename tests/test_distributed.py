"""The N > 1 (data-parallel) path: one process per rank, one sum all-reduce over the flat
gradient arena, AdamW divides by world_size.  CPU: world_size-2 gloo rehearsal of the exchange
and the schedules.  GPU: two ranks sharing the one MI355X of the test box (gloo transport) must
reproduce the single-process step on the concatenated batch."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import golden_params, load_golden, rel_err


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _cpu_worker(rank, world, port, out):
    import vit_som_amd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    z, cfg = load_golden("ref_cluster_tiny")
    torch.manual_seed(0)
    m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device="cpu")
    m.set_distributed(world, rank)
    m.set_schedule(int(z["n_train"]), int(z["est_steps"]))
    # replicas are identical without any broadcast (same seed): compare against rank 0
    flat = m.arena.params.clone()
    dist.broadcast(flat, src=0)
    same_init = bool(torch.equal(flat, m.arena.params))
    # the exchange: ONE all-reduce over the whole arena (ViT grads and prototype accumulators)
    m.arena.grads.copy_(torch.arange(m.arena.numel, dtype=torch.float32) * (rank + 1))
    m.allreduce_gradients()
    expect = torch.arange(m.arena.numel, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok_sum = bool(torch.equal(m.arena.grads, expect))
    # prototypes' accumulator is part of the same buffer
    off, n, _ = m.arena.offsets["som_layer.prototypes"]
    ok_proto = bool(torch.equal(m._grad_views["som_layer.prototypes"].flatten(), expect[off:off + n]))
    # overlapped form: every bucket the backward issues early (prototypes, decoder, encoder block groups) starts
    # asynchronously in backward order, allreduce_gradients() covers the gaps exactly once
    m.arena.grads.copy_(torch.arange(m.arena.numel, dtype=torch.float32) * (rank + 1))
    m._grads_reduced = False
    m._exchange_reset()
    buckets = m._exchange_buckets()
    for name in ["som", "decoder"] + sorted((k for k in buckets if k.startswith("enc")), key=lambda k: -int(k[3:])):
        m._reduce_early(*buckets[name])
    started = len(m._works) == len(buckets) and len(buckets) >= 2
    spans = sorted(buckets.values())
    disjoint = all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    m.allreduce_gradients()
    ok_overlap = started and disjoint and bool(torch.equal(m.arena.grads, expect)) and not m._works
    m.allreduce_gradients()                                   # idempotent until the next backward
    ok_overlap = ok_overlap and bool(torch.equal(m.arena.grads, expect))
    hp = cfg["hyperparameters"]
    ok_T = abs(m.som_layer.total_iterations() - (int(z["n_train"]) / (hp["batch_size"] * world)) * hp["total_epochs"]) < 1e-9
    if rank == 0:
        torch.save({"same_init": same_init, "ok_sum": ok_sum, "ok_proto": ok_proto, "ok_T": ok_T,
                    "ok_overlap": ok_overlap}, out)
    res = torch.tensor([float(same_init and ok_sum and ok_proto and ok_T and ok_overlap)])
    dist.all_reduce(res, op=dist.ReduceOp.MIN)
    assert float(res) == 1.0
    dist.destroy_process_group()


def test_gloo_world2_exchange_and_schedules(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_cpu_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r == {"same_init": True, "ok_sum": True, "ok_proto": True, "ok_T": True, "ok_overlap": True}


def _gpu_worker(rank, world, port, out):
    import vit_som_amd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    z, cfg = load_golden("ref_cluster_tiny")
    cfg = copy.deepcopy(cfg)
    P = golden_params(z)
    x = torch.cat([torch.from_numpy(z["x"]), torch.from_numpy(z["x1"])])      # global batch of 12
    y = torch.cat([torch.from_numpy(z["y"]), torch.from_numpy(z["y1"])])
    per = x.shape[0] // world
    cfg["hyperparameters"]["batch_size"] = per                                 # reference batch_size is per rank
    m = vit_som_amd.ViTSOM(cfg, device="cuda:0")
    m.load_state_dict(P)
    m.set_distributed(world, rank)
    m.set_schedule(120, 40)
    (opt,), _ = m.configure_optimizers()
    xs, ys = x[rank * per:(rank + 1) * per].cuda(), y[rank * per:(rank + 1) * per].cuda()
    losses, grads = [], None
    overlapped = 0
    for s in range(2):
        if s == 0:
            # through the autograd bridge (training_step -> loss.backward()), the path Lightning drives: the
            # bucketed all-reduces start INSIDE backward(), each rank on its own shard of the batch
            loss = m.training_step((xs, ys), 0)
            loss.backward()
            overlapped = len(m._works)
            losses.append(float(loss))
            m.allreduce_gradients()                       # explicit call; optimizer.step() must not reduce twice
            grads = (m.arena.grads / world).cpu()
        else:
            losses.append(float(m.train_step_fused(xs, ys)))
        opt.step()
    torch.cuda.synchronize()
    assert overlapped >= 2, "the early (overlapped) all-reduce pieces were not issued"
    if rank == 0:
        torch.save({"params": m.arena.params.cpu(), "loss": losses, "grads": grads, "lr": opt.param_groups[0]["lr"]}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_equal_single_process_on_concatenated_batch(tmp_path):
    import vit_som_amd
    out = str(tmp_path / "dp.pt")
    mp.spawn(_gpu_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    dp = torch.load(out)
    # single process, global batch
    z, cfg = load_golden("ref_cluster_tiny")
    cfg = copy.deepcopy(cfg)
    x = torch.cat([torch.from_numpy(z["x"]), torch.from_numpy(z["x1"])]).cuda()
    y = torch.cat([torch.from_numpy(z["y"]), torch.from_numpy(z["y1"])]).cuda()
    cfg["hyperparameters"]["batch_size"] = x.shape[0]
    m = vit_som_amd.ViTSOM(cfg, device="cuda:0")
    m.load_state_dict(golden_params(z))
    m.set_schedule(120, 40)
    (opt,), _ = m.configure_optimizers()
    # same lr as the 2-rank run (lr scales with the per-rank batch_size in the reference formula)
    for g in opt.param_groups:
        g["lr"] = g["lr"] / 2
    ref_grads = None
    for s in range(2):
        m.train_step_fused(x, y)
        if s == 0:
            ref_grads = m.arena.grads.cpu()
        opt.step()
    torch.cuda.synchronize()
    ref = m.arena.params.cpu()
    assert abs(opt.param_groups[0]["lr"] - dp["lr"]) < 1e-15
    # the exchange itself: mean of the per-rank gradients == single-process gradient (sum order differs)
    err = rel_err(dp["grads"], ref_grads)
    print(f"2-rank mean gradient vs single process: relative error {err:.3e}")
    assert err < 1e-6
    # Adam's first steps move every weight by ~lr regardless of gradient scale, so compare the
    # UPDATE (param - initial) relatively: summation order differs between 1 and 2 ranks
    init = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device="cuda:0")
    init.load_state_dict(golden_params(z))
    p0 = init.arena.params.cpu()
    assert rel_err(dp["params"] - p0, ref - p0) < 2e-3
    assert torch.allclose(dp["params"], ref, atol=0.25 * dp["lr"])      # within a quarter of one Adam step


@pytest.mark.gpu
def test_vsom_comm_rccl_one_rank_bucketed_exchange_keeps_the_trajectory():
    """The library's own RCCL communicator (vsom_comm_init / _allreduce_sum / _destroy behind the C-ABI): on a 1-rank
    communicator a sum all-reduce is the identity, so the whole bucketed exchange (prototype accumulators first, decoder,
    encoder block groups, remainder -- each enqueued on the exchange stream behind the events of the streams that wrote
    it) with the N > 1 branches forced must leave the 6-step trajectory bit-identical to the plain single-GPU one."""
    import vit_som_amd
    from oracle.gen_golden import make_config
    from vit_som_amd import ops
    from vit_som_amd.tuning import hooks
    hooks.set(launch_tape=False)            # this test counts the exchange calls the HOST issues (the taped form: next test)
    cfg = make_config(3, 32, 4, 192, 6, 3, 96, 2, (12, 12), 0, 64)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(64, 3, 32, 32, generator=g).cuda()
    y = torch.zeros(64, dtype=torch.int64).cuda()

    def run(use_comm):
        torch.manual_seed(0)
        m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device="cuda:0")
        m.set_schedule(50000, 1000)
        calls = []
        if use_comm:
            m.set_distributed(1, 0, backend="rccl")
            assert ops.comm_info() == (1, 0)
            from vit_som_amd.model import _vsom_comm_selftest
            assert _vsom_comm_selftest(1, x.device)   # what set_distributed runs at world > 1 before it trusts the communicator
            m.world_size = 2                          # force the N > 1 branches; AdamW's 1/world is undone below
            orig = ops.comm_allreduce_sum

            def counted(t):
                calls.append(t.numel())
                return orig(t)
            ops.comm_allreduce_sum = counted
        try:
            (opt,), _ = m.configure_optimizers()
            losses = []
            for _ in range(6):
                losses.append(m.train_step_fused(x, y))
                if use_comm:
                    m.allreduce_gradients()
                    m.world_size = 1                  # 1-rank sum: no averaging to undo in AdamW
                opt.step()
                if use_comm:
                    m.world_size = 2
            torch.cuda.synchronize()
        finally:
            if use_comm:
                ops.comm_allreduce_sum = orig
        return [float(v) for v in losses], m.arena.params.clone(), calls, m.arena.numel

    try:
        l0, p0, _, _ = run(False)
        try:
            l1, p1, calls, numel = run(True)
        finally:
            ops.comm_destroy()
        assert ops.comm_info() == (0, -1)
        per_step = len(calls) // 6
        assert per_step >= 4 and sum(calls) == 6 * numel, (per_step, sum(calls), numel)      # every float reduced exactly once per step
        assert l0 == l1 and torch.equal(p0, p1)
        # the same exchange recorded on a launch tape: the collectives of the backward are re-issued from C with the step's
        # other launches, the remainder by allreduce_gradients(); same trajectory
        hooks.set(launch_tape=True)
        try:
            l2, p2, calls2, _ = run(True)
        finally:
            ops.comm_destroy()
        assert l0 == l2 and torch.equal(p0, p2)
        assert len(calls2) < len(calls)          # steps 4-6 issued only the remainder piece from the host
    finally:
        hooks.reset()


def _train_worker(rank, world, port, out, mode):
    import vit_som_amd.train as T
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      VSOM_DIST_BACKEND="gloo")
    z, cfg = load_golden("ref_cls_tiny" if mode == "cls" else "ref_cluster_tiny")
    cfg = copy.deepcopy(cfg)
    cfg["hyperparameters"]["total_epochs"] = 2
    logs = []
    res = T.main(cfg, n_runs=1, make_loaders=lambda c, r, w: T.synthetic_loaders(c, r, w, n_train=96, n_val=24, n_test=24),
                 model_states_dir=os.path.join(os.path.dirname(out), "states"), log=logs.append)
    torch.save({k: v for k, v in res.items() if k not in ("run_duration", "inference_time")}, out + f".{rank}")
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["cls", "cluster"])
def test_train_driver_two_ranks_report_whole_set_metrics(tmp_path, mode):
    """vit_som_amd.train.main at world_size 2 (two ranks on the one GPU of the test box, gloo transport): every rank
    evaluates the reloaded checkpoint on ITS shard, the contingency tables are summed over the ranks, so both ranks
    report the same metrics -- those of the whole set, as the single-process run on the same data does."""
    out = str(tmp_path / "m.pt")
    mp.spawn(_train_worker, args=(2, _free_port(), out, mode), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0 == r1, (r0, r1)
    keys = ("accuracy", "precision", "recall", "f1") if mode == "cls" else ("purity", "nmi")
    assert all(len(r0[k]) == 1 and 0.0 <= r0[k][0] <= 1.0 for k in keys)
    # single process on the same samples: per-rank batch b at world 2 == global batch 2 b (reference batch_size is per rank)
    import vit_som_amd.train as T
    z, cfg = load_golden("ref_cls_tiny" if mode == "cls" else "ref_cluster_tiny")
    cfg = copy.deepcopy(cfg)
    cfg["hyperparameters"]["total_epochs"] = 2
    cfg["hyperparameters"]["batch_size"] *= 2                     # global batch; lr follows batch_size in the reference formula
    cfg["hyperparameters"]["optimizer"]["lr"] /= 2
    for k_ in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k_, None)
    single = T.main(cfg, n_runs=1, make_loaders=lambda c, r, w: T.synthetic_loaders(c, r, w, n_train=96, n_val=24, n_test=24),
                    model_states_dir=str(tmp_path / "states1"), log=lambda *_: None)
    # same samples per step, same schedules; only the summation order differs (a near-tie may move a sample or two)
    for k in keys:
        assert abs(single[k][0] - r0[k][0]) <= 0.1, (k, single[k], r0[k])


def test_bench_launcher_stops_the_other_ranks_when_one_dies(tmp_path):
    """bench.py --gpus N as its own launcher: a rank that exits non-zero ends the job (the others are terminated, not
    left waiting in the rendezvous) and its code is the launcher's."""
    import importlib.util
    import time
    import types
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == os.environ['RANK']\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                      "time.sleep(120)\n")
    t0 = time.time()
    rc = bench.launch_ranks(types.SimpleNamespace(gpus=3), script=str(script), argv=[])
    assert rc != 0 and time.time() - t0 < 60
    ok = tmp_path / "ok.py"
    ok.write_text("import os\nprint(os.environ['RANK'])\n")
    assert bench.launch_ranks(types.SimpleNamespace(gpus=2), script=str(ok), argv=[]) == 0
