"""End-to-end GPU parity of the ViTSOM host class (all HIP kernels chained) against
(1) golden vectors produced by the reference itself and (2) the CPU oracle at larger shapes."""
import copy

import pytest
import torch

from helpers import REF_CASES, golden_params, load_golden, rel_err

pytestmark = pytest.mark.gpu
COSINE_CASES = REF_CASES      # cosine/square, and the euclidean + hexa variant
DEV = "cuda"


def build(cfg, params=None):
    import vit_som_amd
    m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device=DEV)
    if params is not None:
        m.load_state_dict(params)
    return m


def bmu_ok(bmu, dist_ref, eps=2e-6):
    """exact outside fp32 near-ties of the reference distances (SURVEY hard part 3)."""
    ref = dist_ref.argmin(1)
    srt = dist_ref.sort(1).values
    gap = srt[:, 1] - srt[:, 0]
    chosen = dist_ref.gather(1, bmu.view(-1, 1)).squeeze(1)
    return bool(((bmu == ref) | ((gap <= eps) & (chosen - srt[:, 0] <= eps))).all())


@pytest.mark.parametrize("name", COSINE_CASES)
def test_forward_matches_reference_golden(name):
    z, cfg = load_golden(name)
    m = build(cfg, golden_params(z))
    x = torch.from_numpy(z["x"]).to(DEV)
    cls, recon, logits, dist, bmu = m(x)
    # north_star tolerance: 1e-4 fp32 on logits/outputs; measured ~1e-6
    assert torch.allclose(cls.cpu(), torch.from_numpy(z["fwd/cls"]), atol=1e-4)
    assert float((cls.cpu() - torch.from_numpy(z["fwd/cls"])).abs().max()) < 2e-5
    assert torch.allclose(recon.cpu(), torch.from_numpy(z["fwd/recon"]), atol=2e-5)
    assert torch.allclose(dist.cpu(), torch.from_numpy(z["fwd/dist"]), atol=1e-5)
    assert bmu.dtype == torch.int64
    assert torch.equal(bmu.cpu(), torch.from_numpy(z["fwd/bmu"]))        # BMU indices bit-exact
    if m.classification:
        assert torch.allclose(logits.cpu(), torch.from_numpy(z["fwd/logits"]), atol=2e-5)
    else:
        assert logits is None
    # sub-module surface used by tools/evaluation.py
    c2, p2, r2 = m.vit(x)
    assert torch.equal(c2, cls) and torch.equal(r2, recon) and p2.shape == (x.shape[0], m.vit.patch_embed.num_patches, m.vit.embed_dim)
    d2, b2 = m.som_layer(p2)
    assert torch.equal(b2, bmu) and torch.allclose(d2, dist, atol=1e-6)


@pytest.mark.parametrize("name", COSINE_CASES)
def test_training_step_grads_match_reference_golden(name):
    z, cfg = load_golden(name)
    m = build(cfg, golden_params(z))
    m.set_schedule(int(z["n_train"]), int(z["est_steps"]))
    assert m._it == int(z["iteration"])
    x, y = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"]).to(DEV)
    loss = m.training_step((x, y), 0)
    assert abs(float(loss) - float(z["train/loss"])) < 2e-5
    assert abs(float(m.som_layer.current_temperature) - float(z["train/T"])) < 1e-6 * float(z["train/T"])
    loss.backward()
    for n, p in m.named_parameters():
        if not p.requires_grad:
            continue
        ref = torch.from_numpy(z["grad/" + n])
        g = p.grad.cpu()
        assert g.shape == ref.shape
        # relative (prototype gradients are ~1e-7 in magnitude), 1e-4 bar
        assert rel_err(g, ref) < 1e-4 or float((g - ref).abs().max()) < 1e-9, (n, rel_err(g, ref))
    assert int(m.iteration) == int(z["iteration"]) + 1


@pytest.mark.parametrize("name", COSINE_CASES)
def test_three_step_trajectory_matches_reference_golden(name):
    """reference configure_optimizers() + 3 optimizer steps vs FusedAdamW on the flat arena."""
    z, cfg = load_golden(name)
    m = build(cfg, golden_params(z))
    m.set_schedule(int(z["n_train"]), int(z["est_steps"]))
    (opt,), (sched,) = m.configure_optimizers()
    assert abs(opt.param_groups[0]["lr"] - float(z["opt/lr"])) < 1e-12
    batches = [(z["x"], z["y"]), (z["x1"], z["y1"]), (z["x2"], z["y2"])]
    for s, (x, y) in enumerate(batches):
        opt.zero_grad()
        if s == 1:      # exercise the fused (autograd-free) entry too: same kernels
            loss = m.train_step_fused(torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV))
        else:
            loss = m.training_step((torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV)), s)
            loss.backward()
        opt.step()
        if s == 0:
            sd = m.state_dict()
            for k in sd:
                if ("after1/" + k) in z.files and sd[k].is_floating_point():
                    assert torch.allclose(sd[k].cpu(), torch.from_numpy(z["after1/" + k]), atol=2e-6), k
        else:
            assert abs(float(loss) - float(z[f"train/loss{s}"])) < 5e-5
    sd = m.state_dict()
    for k in sd:
        if sd[k].is_floating_point():
            assert torch.allclose(sd[k].cpu(), torch.from_numpy(z["after3/" + k]), atol=1e-5), k
    assert int(sd["iteration"]) == int(z["after3/iteration"])
    # validation_step after training (full gamma, last temperature)
    v = m.validation_step((torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"]).to(DEV)), 0)
    assert abs(float(v) - float(z["val/loss"])) < 5e-5


def cifar_like_cfg(num_classes, map_size, B, depth=3):
    from oracle.gen_golden import make_config
    return make_config(3, 32, 4, 192, depth, 3, 96, 2, map_size, num_classes, B, gamma=0.01, Tmax=4.0, Tmin=0.1)


@pytest.mark.parametrize("num_classes,map_size", [(0, (24, 24)), (10, (4, 4))])
def test_cifar_shapes_against_oracle(num_classes, map_size):
    """Real CIFAR-10 layer shapes (E=192, 3 heads, N=65, L=12288), reduced depth/batch so the CPU
    oracle finishes in seconds; forward outputs, loss and every gradient."""
    from oracle import vitsom_oracle as O
    B = 16
    cfg = cifar_like_cfg(num_classes, map_size, B)
    d = O.Dims(cfg)
    P = O.init_params(cfg, seed=3)
    g = torch.Generator().manual_seed(5)
    for k in O.trainable_keys(P):
        if P[k].ndim == 1:
            P[k] = P[k] + 0.05 * torch.randn(P[k].shape, generator=g)
    x, y = O.synthetic_batch(d, B, seed=1)
    it, n_train, est = 40, 50000, 200
    total, parts, G = O.loss_and_grads(P, x, y, d, it, n_train, est)
    m = build(cfg, P)
    m._it = it
    m.set_schedule(n_train, est)
    cls, recon, logits, dist, bmu = m(x.to(DEV))
    assert torch.allclose(cls.cpu(), parts["cls"], atol=1e-4)
    assert torch.allclose(dist.cpu(), parts["dist"], atol=1e-5)
    assert bmu_ok(bmu.cpu(), parts["dist"].double())
    assert torch.allclose(recon.cpu(), parts["recon"], atol=1e-4)
    if num_classes:
        assert torch.allclose(logits.cpu(), parts["logits"], atol=1e-4)
    loss = m.training_step((x.to(DEV), y.to(DEV)), 0)
    assert abs(float(loss) - float(total)) < 1e-4
    loss.backward()
    same_bmu = torch.equal(bmu.cpu(), parts["bmu"])
    for n, p in m.named_parameters():
        if p.requires_grad:
            if not same_bmu and n == "som_layer.prototypes":
                continue
            e = rel_err(p.grad.cpu(), G[n])
            assert e < 1e-4 or float((p.grad.cpu() - G[n]).abs().max()) < 1e-9, (n, e)


def test_mnist_shapes_against_oracle():
    """c1 shapes: 28x28x1, p=2 -> N=197, E=16 (hd 8), decoder E=4 (hd 2), 24x24 map."""
    from oracle import vitsom_oracle as O
    from oracle.gen_golden import make_config
    B = 8
    cfg = make_config(1, 28, 2, 16, 4, 2, 4, 2, (24, 24), 0, B, gamma=0.005, Tmax=20.0, Tmin=0.001)
    d = O.Dims(cfg)
    P = O.init_params(cfg, seed=4)
    x, y = O.synthetic_batch(d, B, seed=2)
    it, n_train, est = 25, 60000, 100
    total, parts, G = O.loss_and_grads(P, x, y, d, it, n_train, est)
    m = build(cfg, P)
    m._it = it
    m.set_schedule(n_train, est)
    loss = m.training_step((x.to(DEV), y.to(DEV)), 0)
    assert abs(float(loss) - float(total)) < 1e-4
    s = m._ctx[2]
    assert torch.allclose(s.dist.cpu(), parts["dist"], atol=1e-5)
    assert bmu_ok(s.bmu.cpu(), parts["dist"].double())
    loss.backward()
    if torch.equal(s.bmu.cpu(), parts["bmu"]):
        for n, p in m.named_parameters():
            if p.requires_grad:
                e = rel_err(p.grad.cpu(), G[n])
                assert e < 2e-4 or float((p.grad.cpu() - G[n]).abs().max()) < 1e-9, (n, e)


def test_steps_are_deterministic():
    z, cfg = load_golden("ref_cluster_tiny")
    outs = []
    for _ in range(2):
        m = build(cfg, golden_params(z))
        m.set_schedule(60, 40)
        x, y = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"]).to(DEV)
        m.train_step_fused(x, y)
        outs.append(m.arena.grads.clone())
    assert torch.equal(outs[0], outs[1])


def test_stream_level_concurrency_is_bitwise_identical():
    """Weight-gradient GEMMs and the SOM backward run on side streams, and the forward runs as two
    half-batch chains on two streams (model.py).  Arithmetic and summation order are unchanged, so 8
    training steps at CIFAR layer shapes must end in bit-identical parameters with every combination
    of the switches -- any missed ordering edge would show up here."""
    import vit_som_amd
    from oracle.gen_golden import make_config
    from vit_som_amd import ops
    from vit_som_amd.tuning import hooks
    cfg = make_config(3, 32, 4, 192, 4, 3, 96, 2, (12, 12), 0, 96)
    # six products everywhere: the forms of the attention backward are bit-identical there (in the default mode the fused
    # form runs its products on the two-piece split); then the default mode with the attention form fixed
    for mode, flip_attention in ((ops.GEMM_SPLIT_BF16, True), (ops.GEMM_SPLIT_BF16_GRAD3, False)):
        finals = []
        prev = ops.get_gemm_mode()
        ops.set_gemm_mode(mode)
        try:
            _run_switch_combinations(cfg, finals, hooks, ops, vit_som_amd, flip_attention)
        finally:
            hooks.reset()
            ops.set_attention_fused(True)
            ops.set_gemm_mode(prev)
        assert all(torch.equal(finals[0], f) for f in finals[1:])


def _run_switch_combinations(cfg, finals, hooks, ops, vit_som_amd, flip_attention=True):
    # (last column: the LayerNorm backwards' column reductions in one launch per exchange piece, or one launch each)
    # and the BMU pass on the SOM stream under the decoder forward, or on the main chain)
    for side, split, nblk, fused, lnb in (("0", "0", None, True, True), ("1", "1", None, True, True), ("1", "0", None, False, False),
                                          ("0", "1", 2, True, False), ("1", "1", 1, False, True), ("1", "1", None, True, False)):
        hooks.set(side_stream=side == "1", fwd_split=split == "1", fwd_split_blocks=nblk, ln_reduce_batched=lnb, bmu_overlap=lnb)
        ops.set_attention_fused(fused if flip_attention else True)
        torch.manual_seed(0)
        m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device=DEV)
        m.set_schedule(5000, 500)
        m._it = 100
        (opt,), _ = m.configure_optimizers()
        g = torch.Generator().manual_seed(5)
        for _ in range(8):
            x = torch.randn(96, 3, 32, 32, generator=g).to(DEV)
            y = torch.zeros(96, dtype=torch.int64, device=DEV)
            m.train_step_fused(x, y)
            opt.step()
        assert (m.vit._side is not None) == (side == "1")
        jobs = m.vit._acts[96].__dict__.get("ln_jobs")
        assert (jobs is not None and jobs.n == jobs.flushed == 2 * (4 + 2) + 2) == lnb      # every LayerNorm of the step went through the batch
        assert (m.vit.__dict__.get("_fwd_side") is not None) == (split == "1")
        assert split == "0" or m.vit._fwd_side is m._side_stream          # the second chain borrows the backward's side stream
        finals.append(m.arena.params.clone())


def test_use_reduced_cls_token_som():
    """use_reduced=True: SOM on the CLS token (L = E), vit_som.py:70-71."""
    from oracle import vitsom_oracle as O
    z, cfg = load_golden("ref_cls_tiny")
    cfg = copy.deepcopy(cfg)
    cfg["hyperparameters"]["som"]["use_reduced"] = True
    d = O.Dims(cfg)
    P = O.init_params(cfg, seed=7)
    x, y = O.synthetic_batch(d, 5, seed=3)
    total, parts, G = O.loss_and_grads(P, x, y, d, 6, 50, 20)
    m = build(cfg, P)
    m._it = 6
    m.set_schedule(50, 20)
    loss = m.training_step((x.to(DEV), y.to(DEV)), 0)
    loss.backward()
    assert abs(float(loss) - float(total)) < 2e-5
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert rel_err(p.grad.cpu(), G[n]) < 1e-4 or float((p.grad.cpu() - G[n]).abs().max()) < 1e-9, n


def _shape_case(cfg, B, seed, it=30, n_train=100000, est=400, grad_tol=2e-4):
    from oracle import vitsom_oracle as O
    d = O.Dims(cfg)
    P = O.init_params(cfg, seed=seed)
    x, y = O.synthetic_batch(d, B, seed=seed + 1)
    total, parts, G = O.loss_and_grads(P, x, y, d, it, n_train, est)
    m = build(cfg, P)
    m._it = it
    m.set_schedule(n_train, est)
    loss = m.training_step((x.to(DEV), y.to(DEV)), 0)
    s = m._ctx[2]
    assert abs(float(loss) - float(total)) < 1e-4
    assert torch.allclose(s.dist.cpu(), parts["dist"], atol=2e-5)
    assert bmu_ok(s.bmu.cpu(), parts["dist"].double())
    if m.classification:
        assert torch.allclose(m._ctx[1].logits.cpu(), parts["logits"], atol=1e-4)
    loss.backward()
    if torch.equal(s.bmu.cpu(), parts["bmu"]):
        for n, p in m.named_parameters():
            if p.requires_grad:
                e = rel_err(p.grad.cpu(), G[n])
                assert e < grad_tol or float((p.grad.cpu() - G[n]).abs().max()) < 1e-9, (n, e)
    return m


def test_c4_shapes_cifar100_classification_head():
    """BASELINE config c4: vit_som-cls, CIFAR-100 (100 classes), 4x4 SOM (K=16, HBM-bound BMU shape)."""
    from oracle.gen_golden import make_config
    cfg = make_config(3, 32, 4, 192, 2, 3, 96, 2, (4, 4), 100, 24, gamma=0.01, Tmax=4.0, Tmin=0.1)
    _shape_case(cfg, 24, seed=21)


def test_c5_shapes_tiny_imagenet_40x40():
    """BASELINE config c5: Tiny-ImageNet 64x64 -> N=257 tokens, L=49152, 40x40 SOM (the largest shapes)."""
    from oracle.gen_golden import make_config
    cfg = make_config(3, 64, 4, 192, 1, 3, 96, 1, (40, 40), 0, 4, gamma=0.01, Tmax=4.0, Tmin=0.1)
    m = _shape_case(cfg, 4, seed=31, grad_tol=3e-4)
    assert m.som_layer.latent_dim == 49152 and m.vit.patch_embed.num_patches == 256


def test_full_size_c3_step_properties():
    """BASELINE c3 at its FULL size (CIFAR-10 shapes, 12 + 2 layers, 40x40 SOM, batch 512: the bench
    workload), where the CPU oracle would take minutes: size-independent properties instead.
      * BMU policy: bmu == argmin (first minimum) of the distances the step itself produced, bit for bit;
      * distances against an fp64 evaluation (on the device, plain torch) of 1 - xhat . what from the SOM input;
      * determinism: a second, identically seeded model gives a bit-identical loss and gradient arena;
      * linearity of the backward pass in the loss seed: backward(2 * loss) == 2 * backward(loss), bit for bit
        (powers of two commute with fp32 rounding);
      * the weighted SOM loss equals mean(h * dist) recomputed in fp64 from the kernel's own (bmu, T)."""
    import vit_som_amd
    import bench
    cfg = bench.c3_config(512)
    arenas, losses = [], []
    g = torch.Generator().manual_seed(11)
    x = torch.randn(512, 3, 32, 32, generator=g).to(DEV)
    y = torch.zeros(512, dtype=torch.int64, device=DEV)
    for rep in range(2):
        torch.manual_seed(0)
        m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device=DEV)
        m.set_schedule(50000, 9700)
        m._it = 1000
        loss = m.training_step((x, y), 0)
        (loss * (2.0 if rep else 1.0)).backward()
        arenas.append(m.arena.grads.clone())
        losses.append(float(loss.detach()))
    assert losses[0] == losses[1]
    assert torch.equal(arenas[1], 2.0 * arenas[0])                     # deterministic AND linear in the seed
    a, s = m._ctx[1], m._ctx[2]
    assert torch.equal(s.bmu, s.dist.argmin(dim=1))
    X = m._som_input(a).double()
    W = m.som_layer.prototypes.detach().double()
    ref = 1.0 - torch.nn.functional.normalize(X, dim=1) @ torch.nn.functional.normalize(W, dim=1).T
    assert float((s.dist.double() - ref).abs().max()) < 2e-6
    gap = ref.topk(2, dim=1, largest=False).values
    sure = (gap[:, 1] - gap[:, 0]) > 4e-6                             # rows whose fp64 winner is unambiguous in fp32
    assert int(sure.sum()) > 400 and torch.equal(s.bmu[sure], ref.argmin(dim=1)[sure])
    T = float(m.som_layer.current_temperature)
    pos = m.som_layer.grid_positions.double()
    h = torch.exp(-(pos[None, :, :] - pos[s.bmu][:, None, :]).pow(2).sum(-1) / (2 * T * T))
    assert abs(float(m._last["som"]) - float((h * s.dist.double()).mean())) < 1e-7


@pytest.mark.parametrize("name,chans,img,p,E,depth,heads,DE,map_size,B", [
    ("c1", 1, 28, 2, 16, 4, 2, 4, (24, 24), 128),         # BASELINE configs[0]: MNIST shapes, N = 197 tokens, L = 3136
    ("c2", 3, 32, 4, 192, 12, 3, 96, (24, 24), 512)])     # BASELINE configs[1]: CIFAR-10 shapes, 24x24 SOM, batch 512
def test_full_size_c1_c2_step_properties(name, chans, img, p, E, depth, heads, DE, map_size, B):
    """BASELINE c1 and c2 at their FULL size (depth, map, batch) through the same size-independent properties as c3:
    determinism + bitwise linearity of the backward in the loss seed, BMU == first argmin of the step's own distances,
    distances against fp64 on the device, BMU == fp64 argmin on every row whose winner is unambiguous in fp32, the SOM
    loss recomputed in fp64, and no row of the cosine pass left to an un-re-ranked tie."""
    import vit_som_amd
    from oracle.gen_golden import make_config
    cfg = make_config(chans, img, p, E, depth, heads, DE, 2, map_size, 0, B, gamma=0.01, Tmax=4.0, Tmin=0.1)
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, chans, img, img, generator=g).to(DEV)
    y = torch.zeros(B, dtype=torch.int64, device=DEV)
    arenas, losses = [], []
    for rep in range(2):
        torch.manual_seed(0)
        m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device=DEV)
        m.set_schedule(60000, 9000)
        m._it = 800
        loss = m.training_step((x, y), 0)
        (loss * (2.0 if rep else 1.0)).backward()
        arenas.append(m.arena.grads.clone())
        losses.append(float(loss.detach()))
    # deterministic AND linear in the seed, bit for bit wherever doubling is exact (gradients of prototypes far from every
    # BMU reach the fp32 denormal range, ~1e-41, where a factor of two rounds -- and sums that contain such addends)
    assert losses[0] == losses[1]
    big = arenas[0].abs() > 1e-24                       # no denormal addend can reach the last bit of these
    assert int(big.sum()) > 0.5 * arenas[0].numel() and torch.equal(arenas[1][big], 2.0 * arenas[0][big])
    assert float((arenas[1] - 2.0 * arenas[0])[~big].abs().max()) < 1e-30
    a, s = m._ctx[1], m._ctx[2]
    assert s.dist.shape == (B, map_size[0] * map_size[1]) and torch.equal(s.bmu, s.dist.argmin(dim=1))
    X = m._som_input(a).double()
    W = m.som_layer.prototypes.detach().double()
    ref = 1.0 - torch.nn.functional.normalize(X, dim=1) @ torch.nn.functional.normalize(W, dim=1).T
    assert float((s.dist.double() - ref).abs().max()) < 2e-6
    gap = ref.topk(2, dim=1, largest=False).values
    sure = (gap[:, 1] - gap[:, 0]) > 4e-6
    assert int(sure.sum()) > 0.75 * B and torch.equal(s.bmu[sure], ref.argmin(dim=1)[sure])
    T = float(m.som_layer.current_temperature)
    pos = m.som_layer.grid_positions.double()
    h = torch.exp(-(pos[None, :, :] - pos[s.bmu][:, None, :]).pow(2).sum(-1) / (2 * T * T))
    assert abs(float(m._last["som"]) - float((h * s.dist.double()).mean())) < 1e-7
    for n_, p_ in m.named_parameters():
        if p_.requires_grad:
            assert torch.isfinite(p_.grad).all(), n_


@pytest.mark.parametrize("name,img,classes,map_size,B", [("c4", 32, 100, (4, 4), 128), ("c5", 64, 200, (40, 40), 256)])
def test_full_size_c4_c5_step_properties(name, img, classes, map_size, B):
    """BASELINE c4 (CIFAR-100 shapes, 4x4 SOM, 1024 global = 128 per GPU) and c5 (Tiny-ImageNet 64x64 -> 257
    tokens, L = 49152, 40x40 SOM, 256 per GPU) at full depth and per-GPU batch, classification mode:
    determinism, the BMU policy, distances against fp64, logits against an fp64 head, zero decoder gradients."""
    import vit_som_amd
    from oracle.gen_golden import make_config
    cfg = make_config(3, img, 4, 192, 12, 3, 96, 2, map_size, classes, B, gamma=0.01, Tmax=4.0, Tmin=0.1)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, 3, img, img, generator=g).to(DEV)
    y = torch.randint(0, classes, (B,), generator=g).to(DEV)
    arenas, losses = [], []
    for rep in range(2):
        torch.manual_seed(0)
        m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device=DEV)
        m.set_schedule(100000, 5000)
        m._it = 500
        losses.append(float(m.train_step_fused(x, y)))
        arenas.append(m.arena.grads.clone())
        if rep == 0:
            del m
    assert losses[0] == losses[1] and torch.equal(arenas[0], arenas[1])
    a, s = m._ctx[1], m._ctx[2]
    assert torch.equal(s.bmu, s.dist.argmin(dim=1))
    X = m._som_input(a).double()
    W = m.som_layer.prototypes.detach().double()
    ref = 1.0 - torch.nn.functional.normalize(X, dim=1) @ torch.nn.functional.normalize(W, dim=1).T
    assert float((s.dist.double() - ref).abs().max()) < 2e-6
    gap = ref.topk(2, dim=1, largest=False).values
    sure = (gap[:, 1] - gap[:, 0]) > 4e-6                             # rows whose fp64 winner is unambiguous in fp32
    assert int(sure.sum()) > 0.75 * B and torch.equal(s.bmu[sure], ref.argmin(dim=1)[sure])
    cls_tok = m._cls_view(a.xe, a).double()
    logits_ref = cls_tok @ m.cls_head.weight.detach().double().T + m.cls_head.bias.detach().double()
    assert float((a.logits.double() - logits_ref).abs().max()) < 1e-5
    for n in m._decoder_param_names():
        assert float(m._grad_views[n].abs().max()) == 0.0
    assert all(bool(torch.isfinite(v).all()) for v in m._grad_views.values())


@pytest.mark.parametrize("B", [1, 3, 7])
def test_small_and_odd_batches(B):
    from oracle import vitsom_oracle as O
    z, cfg = load_golden("ref_cluster_tiny")
    P = golden_params(z)
    d = O.Dims(cfg)
    x, y = O.synthetic_batch(d, B, seed=B)
    total, parts, G = O.loss_and_grads(P, x, y, d, 7, 60, 40)
    m = build(cfg, P)
    m.set_schedule(60, 40)
    loss = m.training_step((x.to(DEV), y.to(DEV)), 0)
    loss.backward()
    assert abs(float(loss) - float(total)) < 2e-5
    assert torch.equal(m._ctx[2].bmu.cpu(), parts["bmu"])
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert rel_err(p.grad.cpu(), G[n]) < 1e-4 or float((p.grad.cpu() - G[n]).abs().max()) < 1e-9, n


def test_bad_inputs_fail_loudly():
    from vit_som_amd._lib import VsomError
    z, cfg = load_golden("ref_cluster_tiny")
    m = build(cfg, golden_params(z))
    with pytest.raises((VsomError, ValueError, RuntimeError)):
        m(torch.empty(0, 1, 8, 8, device=DEV))                       # empty batch
    with pytest.raises(ValueError):
        m(torch.zeros(2, 3, 8, 8, device=DEV))                       # wrong channel count
    with pytest.raises(ValueError):
        m(torch.zeros(2, 1, 16, 16, device=DEV))                     # wrong image size


@pytest.mark.parametrize("name", ["ref_cluster_tiny", "ref_hexa_euclid_tiny", "ref_manhattan_tiny"])
def test_submodules_are_differentiable_on_their_own(name):
    """A reference user composes model.vit(x) / model.som_layer(z) / som_loss(w, d) in a loss of their own (SURVEY 8(b):
    every op differentiable on the torch side): gradients of such a composition against the CPU oracle under torch
    autograd, with weights that are NOT the neighbourhood of the BMU (som_loss must honour its `weights` argument)."""
    from oracle import vitsom_oracle as O
    z, cfg = load_golden(name)
    P = golden_params(z)
    d = O.Dims(cfg)
    x = torch.from_numpy(z["x"])
    B, K = x.shape[0], P["som_layer.prototypes"].shape[0]
    w_any = torch.rand(B, K, generator=torch.Generator().manual_seed(3)) + 0.1          # arbitrary positive weights

    def compose(vit_fn, som_fn, loss_fn, xin, wts):
        cls, patches, recon = vit_fn(xin)
        dist, bmu = som_fn(patches.flatten(1))
        return 0.7 * loss_fn(wts, dist) + 0.3 * (recon - xin).abs().mean() + 0.1 * cls.pow(2).mean() + 0.05 * patches.mean(), bmu

    # oracle (CPU autograd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items() if k in O.trainable_keys(P)}
    Pl = dict(P); Pl.update(leaves)
    ref, bmu_ref = compose(lambda t: O.vit_forward(Pl, t, d), lambda t: O.som_forward(t, Pl["som_layer.prototypes"], d.distance_fcn),
                           O.som_loss, x, w_any)
    ref.backward()
    # the build
    m = build(cfg, P)
    m.zero_grad(set_to_none=True)
    out, bmu = compose(m.vit, m.som_layer, m.som_layer.som_loss, x.to(DEV), w_any.to(DEV))
    out.backward()
    assert abs(float(out) - float(ref)) < 2e-5
    assert torch.equal(bmu.cpu(), bmu_ref)
    worst = 0.0
    for n, p in m.named_parameters():
        if not p.requires_grad or n.startswith("cls_head"):
            continue
        assert p.grad is not None, n
        worst = max(worst, rel_err(p.grad.cpu(), leaves[n].grad))
    assert worst < 1e-4, worst
    # som_loss on its own: value and both gradients for arbitrary weights
    dd = torch.rand(B, K, generator=torch.Generator().manual_seed(4)).to(DEV).requires_grad_(True)
    ww = w_any.to(DEV).clone().requires_grad_(True)
    l = m.som_layer.som_loss(ww, dd)
    l.backward()
    assert abs(float(l) - float((w_any * dd.detach().cpu()).mean())) < 1e-6
    assert torch.allclose(dd.grad.cpu(), w_any / (B * K), atol=1e-9) and torch.allclose(ww.grad, dd.detach() / (B * K), atol=1e-9)


def test_optimizer_step_with_closure_runs_backward():
    """Lightning's automatic optimization calls optimizer.step(closure) with a closure that runs training_step +
    backward: the closure must execute with autograd enabled (torch.optim.AdamW does the same)."""
    z, cfg = load_golden("ref_cluster_tiny")
    m = build(cfg, golden_params(z))
    m.set_schedule(int(z["n_train"]), int(z["est_steps"]))
    (opt,), _ = m.configure_optimizers()
    x, y = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"]).to(DEV)
    before = m.arena.params.clone()

    def closure():
        loss = m.training_step((x, y), 0)
        loss.backward()
        return loss

    loss = opt.step(closure)
    assert loss is not None and torch.isfinite(loss) and not torch.equal(before, m.arena.params)


@pytest.mark.gpu
def test_iteration_buffer_advances_with_the_training_step_only():
    """vit_som.py:104 `self.iteration += 1`: the device buffer moves inside the step's own loss kernel (no ATen
    launch); validation leaves it alone; state_dict round-trips it."""
    from oracle.gen_golden import make_config
    torch.manual_seed(0)
    cfg = make_config(3, 32, 4, 192, 2, 3, 96, 2, (4, 4), 10, 8, gamma=0.01, Tmax=4.0, Tmin=0.1)
    model = build(cfg)
    model.set_schedule(64, 10)
    (opt,), _ = model.configure_optimizers()
    x = torch.randn(8, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (8,), device=DEV)
    for _ in range(3):
        model.training_step((x, y), 0).backward()
        opt.step()
    model.validation_step((x, y), 0)
    assert int(model.iteration) == 3 and model.state_dict()["iteration"].item() == 3
    model.train_step_fused(x, y)
    assert int(model.iteration) == 4


DECODER_CASES = ["ref_decoder_cluster_tiny", "ref_decoder_mnistlike_tiny", "ref_decoder_cls_tiny"]


@pytest.mark.parametrize("name", DECODER_CASES)
def test_forward_decoder_features_and_attention_maps_match_reference_golden(name):
    """ViTAutoencoder.forward_decoder on an arbitrary token tensor (vit.py:182-200; caller tools/evaluation.py:209-222),
    forward_features (vit.py:155-179) and the attention maps of return_attn(s)=True (vit.py:41-42) against outputs of the
    reference itself."""
    z, cfg = load_golden(name)
    m = build(cfg, golden_params(z))
    vit = m.vit
    tokens, x = torch.from_numpy(z["tokens"]).to(DEV), torch.from_numpy(z["x"]).to(DEV)
    patches, none = vit.forward_decoder(tokens, return_attn=False)
    assert none is None and patches.shape == z["dec/patches"].shape
    assert torch.allclose(patches.cpu(), torch.from_numpy(z["dec/patches"]), atol=2e-5)
    assert torch.allclose(vit.unpatchify(patches).cpu(), torch.from_numpy(z["dec/recon"]), atol=2e-5)
    p2, maps = vit.forward_decoder(tokens, return_attn=True)
    assert torch.equal(p2, patches) and len(maps) == len(vit.decoder_blocks)
    for i, a_ in enumerate(maps):
        ref = torch.from_numpy(z[f"dec/attn{i}"])
        assert a_.shape == ref.shape and torch.allclose(a_.cpu(), ref, atol=2e-6)
        assert torch.allclose(a_.sum(-1).cpu(), torch.ones(ref.shape[:-1]), atol=1e-5)
    cls, nothing = vit.forward_features(x)
    assert nothing is None and torch.allclose(cls.cpu(), torch.from_numpy(z["ff/cls"]), atol=2e-5)
    cls2, emaps = vit.forward_features(x, return_attns=True)
    assert torch.equal(cls2, cls) and len(emaps) == int(z["ff/n_attn"])
    out = vit(x, return_attns=True)
    assert len(out) == 4 and torch.allclose(out[2].cpu(), torch.from_numpy(z["fwd/recon"]), atol=2e-5)
    for i, (a_, b_) in enumerate(zip(emaps, out[3])):
        ref = torch.from_numpy(z[f"fwd/attn{i}"])
        assert torch.allclose(a_.cpu(), ref, atol=2e-6) and torch.equal(a_, b_)
    # decoder fed with the encoder's own tokens reproduces forward()'s reconstruction bit for bit (same kernels)
    cls3, ptok, recon = vit(x)
    full = torch.cat([cls3.unsqueeze(1), ptok], dim=1)
    again, _ = vit.forward_decoder(full)
    assert torch.equal(vit.unpatchify(again), recon)
    # a second cached batch size did not evict the first one's buffers
    assert set(vit._acts) == {x.shape[0], tokens.shape[0]} or x.shape[0] == tokens.shape[0]
    with pytest.raises(ValueError):
        vit.forward_decoder(tokens[:, :-1])


def test_decoder_and_features_are_differentiable_on_their_own():
    """forward_decoder / forward_features under torch autograd against the oracle under CPU autograd: parameter
    gradients and the gradient w.r.t. the token tensor forward_decoder was fed."""
    from oracle import vitsom_oracle as O
    z, cfg = load_golden("ref_decoder_mnistlike_tiny")
    P = golden_params(z)
    d = O.Dims(cfg)
    tokens, x = torch.from_numpy(z["tokens"]), torch.from_numpy(z["x"])
    wts = torch.randn(tuple(z["dec/patches"].shape), generator=torch.Generator().manual_seed(9))
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items() if k in O.trainable_keys(P)}
    Pl = dict(P); Pl.update(leaves)
    tk = tokens.clone().requires_grad_(True)
    ref = (O.vit_forward_decoder(Pl, tk, d)[0] * wts).sum() + O.vit_forward_features(Pl, x, d)[0].pow(2).sum()
    ref.backward()
    m = build(cfg, P)
    m.zero_grad(set_to_none=True)
    tg = tokens.to(DEV).requires_grad_(True)
    l1 = (m.vit.forward_decoder(tg)[0] * wts.to(DEV)).sum()
    l1.backward()                                  # before the next forward: the activation buffers are reused
    g_dec = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    l2 = m.vit.forward_features(x.to(DEV))[0].pow(2).sum()
    l2.backward()
    assert abs(float(l1) + float(l2) - float(ref)) < 1e-4 * max(1.0, abs(float(ref)))
    assert rel_err(tg.grad.cpu(), tk.grad) < 1e-4
    worst = 0.0
    for n, p in m.named_parameters():
        if not n.startswith("vit.") or not p.requires_grad:
            continue
        g = (p.grad if p.grad is not None else 0) + g_dec.get(n, 0)
        r = leaves[n].grad
        if r is None or float(r.abs().max()) == 0.0:
            assert float(torch.as_tensor(g).abs().max()) == 0.0, n
            continue
        worst = max(worst, rel_err(g.cpu(), r))
    assert worst < 1e-4, worst


def test_stale_or_repeated_backward_is_refused():
    """The activation buffers are reused per batch size: a backward after they were rewritten (by ANY forward: no_grad
    calls and the fused training step included) raises instead of returning gradients of the wrong batch; one
    training_step gives one backward; an input image that requires grad is refused."""
    z, cfg = load_golden("ref_cluster_tiny")
    m = build(cfg, golden_params(z))
    m.set_schedule(int(z["n_train"]), int(z["est_steps"]))
    x, y = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"]).to(DEV)
    out = m.vit(x)
    with torch.no_grad():
        m.vit(x)                                   # rewrites the buffers without autograd
    with pytest.raises(RuntimeError, match="rewritten"):
        out[0].sum().backward()
    out = m.vit(x)
    m.train_step_fused(x, y)                       # the fused step writes the same buffers
    with pytest.raises(RuntimeError, match="rewritten"):
        out[2].sum().backward()
    loss = m.training_step((x, y), 0)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="twice"):
        loss.backward()
    with pytest.raises(RuntimeError, match="input image"):
        m.vit(x.clone().requires_grad_(True))


def test_loss_parts_are_plain_tensors_with_a_lifetime():
    """model._last is an ordinary dict of tensors ('total', 'main', 'som'; vit_som.py:95-102 logs them): .get / `in` /
    iteration see every key, and a step's values stay valid while the next steps run."""
    z, cfg = load_golden("ref_cluster_tiny")
    m = build(cfg, golden_params(z))
    m.set_schedule(int(z["n_train"]), int(z["est_steps"]))
    x, y = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"]).to(DEV)
    l0 = m.train_step_fused(x, y)
    first = dict(m._last)
    assert type(m._last) is dict and {"total", "main", "som", "gamma_t", "T"} <= set(first) and m._last.get("main") is not None
    v0 = {k: float(first[k]) for k in ("total", "main", "som")}
    assert abs(v0["total"] - (v0["main"] + first["gamma_t"] * v0["som"])) < 1e-6 and abs(v0["total"] - float(z["train/loss"])) < 2e-5
    x1, y1 = torch.from_numpy(z["x1"]).to(DEV), torch.from_numpy(z["y1"]).to(DEV)
    for _ in range(3):
        m.train_step_fused(x1, y1)
        m.validation_step((x1, y1), 0)
    assert {k: float(first[k]) for k in ("total", "main", "som")} == v0 and float(l0) == v0["total"]


@pytest.mark.parametrize("mode,B", [("cluster", 64), ("cls", 24), ("cluster", 192)])      # 192: BMU pass on plane images
def test_launch_tape_replays_the_step_bit_for_bit(mode, B):
    """The launch tape (vsom_tape_*): after two host-driven steps the third is recorded while it runs and every later
    step re-issues those launches from C.  8 steps through train_step_fused AND through training_step().backward() (with a
    loss seed other than 1: segment 2) must leave bit-identical parameters, losses and logged terms with the tape on and off;
    a validation step in between (host-driven, same buffers) must not disturb it."""
    import vit_som_amd
    from oracle.gen_golden import make_config
    from vit_som_amd import ops
    from vit_som_amd.tuning import hooks
    cfg = make_config(3, 32, 4, 192, 3, 3, 96, 2, (8, 8), 10 if mode == "cls" else 0, B, gamma=0.02, Tmax=4.0, Tmin=0.1)
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(B, 3, 32, 32, generator=g).to(DEV) for _ in range(8)]
    ys = [torch.randint(0, 10, (B,), generator=g).to(DEV) for _ in range(8)]

    def run(tape_on):
        hooks.set(launch_tape=tape_on)
        try:
            torch.manual_seed(0)
            m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device=DEV)
            m.set_schedule(4000, 400)
            m._it = 40
            (opt,), _ = m.configure_optimizers()
            out = []
            for i in range(8):
                if i % 3 == 2:                                      # the autograd bridge, seed 0.5
                    loss = m.training_step((xs[i], ys[i]), i)
                    (0.5 * loss).backward()
                else:
                    loss = m.train_step_fused(xs[i], ys[i])
                out.append((float(loss), float(m._last["main"]), float(m._last["som"])))
                opt.step()
                if i == 4:
                    m.validation_step((xs[0], ys[0]), 0)
            torch.cuda.synchronize()
            a = m.vit._acts[B]
            return out, m.arena.params.clone(), m.arena.grads.clone(), a.__dict__.get("tape"), int(m.iteration)
        finally:
            hooks.reset()

    o0, p0, g0, t0, it0 = run(False)
    o1, p1, g1, t1, it1 = run(True)
    assert t0 is None and t1 is not None and t1.id > 0
    nops = [ops.tape_segment_ops(t1.id, k) for k in range(4)]
    assert nops[0] > 20 and nops[1] >= 1 and nops[2] == 4 and nops[3] > 60, nops
    assert o0 == o1 and it0 == it1 == 8            # the device-side iteration buffer advanced once per training step, replayed or not
    assert torch.equal(g0, g1) and torch.equal(p0, p1)


def test_bmu_plane_images_follow_the_prototypes():
    """The cosine BMU pass on pre-split plane images (B >= 192): the prototypes' image is rewritten by the optimizer step,
    the step results equal the in-loop-split path's (losses to fp32 rounding of the row norms, same BMUs), and every other
    change of the prototypes -- an in-place torch op, load_state_dict, invalidate_planes() after a raw write -- is seen."""
    import vit_som_amd
    from oracle.gen_golden import make_config
    from vit_som_amd import ops
    from vit_som_amd.tuning import hooks
    B = 192
    cfg = make_config(3, 32, 4, 192, 3, 3, 96, 2, (8, 8), 0, B, gamma=0.02, Tmax=4.0, Tmin=0.1)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(B, 3, 32, 32, generator=g).to(DEV) for _ in range(6)]
    y = torch.zeros(B, dtype=torch.int64, device=DEV)

    def run(planes_on, adamw_planes=False):
        hooks.set(bmu_planes=planes_on, adamw_planes=adamw_planes)
        try:
            torch.manual_seed(0)
            m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device=DEV)
            m.set_schedule(4000, 400)
            m._it = 40
            (opt,), _ = m.configure_optimizers()
            out = []
            for i in range(6):
                loss = m.train_step_fused(xs[i], y)
                out.append((float(loss), m._ctx[2].bmu.clone(), m._ctx[2].dist.clone()))
                opt.step()
            return m, out
        finally:
            hooks.reset()

    m0, o0 = run(False)
    m1, o1 = run(True)                                               # default: re-split on the SOM stream every training step
    m2, o2 = run(True, adamw_planes=True)                            # the optimizer writes the image
    som = m1.som_layer
    assert m0.som_layer._wplanes is None and som._wplanes is not None and som._planes_used
    assert som._wplanes_stamp != som._w_stamp()                      # the optimizer step went past the image ...
    som2 = m2.som_layer
    assert som2._wplanes_stamp == som2._w_stamp()                    # ... here it left it current,
    fresh = ops.bmu_planes_alloc(*som2.prototypes.shape, DEV)
    ops.bmu_planes_from(som2.prototypes.detach(), fresh)
    assert torch.equal(fresh, som2._wplanes)                         # and it is the image of the updated prototypes
    assert torch.equal(m1.arena.params, m2.arena.params)             # same arithmetic either way
    assert [l for l, _, _ in o1] == [l for l, _, _ in o2]
    for i, ((l0, b0, d0), (l1, b1, d1)) in enumerate(zip(o0, o1)):
        assert abs(l0 - l1) <= 2e-6 * abs(l0), (i, l0, l1)
        assert float((d0 - d1).abs().max()) < 1e-6, i
        assert int((b0 != b1).sum()) <= 1, i                         # a last-bit norm difference may flip an exact-to-fp32 tie

    x = xs[0]
    with torch.no_grad():
        fwd = m1.forward(x)
    # in-place torch op
    with torch.no_grad():
        som.prototypes[5].copy_(som.prototypes[9])
        som.prototypes.mul_(1.5)
    ref_d = 1 - torch.nn.functional.normalize(m1._som_input(m1.vit._acts[B]).double(), dim=1) @ \
        torch.nn.functional.normalize(som.prototypes.detach().double(), dim=1).T
    s = som._buffers_for(B, x.device)
    som._distances_into(m1._som_input(m1.vit._acts[B]), s)
    assert som._wplanes_stamp == som._w_stamp()
    assert float((s.dist.double() - ref_d).abs().max()) < 1e-5
    assert torch.equal(s.bmu, s.dist.argmin(1))
    tie = s.bmu.cpu()
    assert not bool((tie == 9).any())                                # rows 5 and 9 are identical now: the lower index wins
    # raw write + invalidate_planes()
    som.prototypes.data[7] = som.prototypes.data[2]                  # bypasses the version counter of the Parameter
    som.invalidate_planes()
    som._distances_into(m1._som_input(m1.vit._acts[B]), s)
    assert not bool((s.bmu == 7).any())
    # load_state_dict
    sd = {k: v.clone() for k, v in m0.state_dict().items()}
    m1.load_state_dict(sd)
    assert som._wplanes_stamp != som._w_stamp()
    with torch.no_grad():
        out1 = m1.forward(x)
        out0 = m0.forward(x)
    assert len(fwd) == len(out1)
    for a0, a1 in zip(out0, out1):
        if torch.is_tensor(a0) and a0.is_floating_point():
            assert float((a0 - a1).abs().max()) < 1e-6
