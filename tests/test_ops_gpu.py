"""GPU parity tests, one C-ABI entry at a time, against the CPU oracle / torch fp32 (fp64 where
summation order matters).  Tolerances are stated per test; index outputs are exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from vit_som_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def O():
    from oracle import vitsom_oracle
    return vitsom_oracle


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def dev(t):
    return t.to(DEV)


# fp32 GEMM tolerance: relative Frobenius error vs an fp64 reference -- the SAME bound for the
# exact-f32 MFMA engine and the split-bf16 engine (both are fp32-accurate)
GEMM_TOL = 2e-6
# the default mode's gradient GEMMs (vsom_linear_bwd_weight / _bwd_input_t): two-piece split, three products -- worst case
# 3 * 2^-16 = 4.6e-5 per product, measured 4e-6 on random operands; the model-level gradient bar stays 1e-4
GRAD3_TOL = 2e-5


@pytest.fixture(params=["grad3", "split_bf16", "f32"])
def gemm_mode(request, ops):
    """Run a test under every arithmetic mode of the nn.Linear-shaped GEMMs (grad3 = the default)."""
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode({"grad3": ops.GEMM_SPLIT_BF16_GRAD3, "split_bf16": ops.GEMM_SPLIT_BF16, "f32": ops.GEMM_F32}[request.param])
    yield request.param
    ops.set_gemm_mode(prev)


def grad_tol(mode):
    return GRAD3_TOL if mode == "grad3" else GEMM_TOL


@pytest.mark.parametrize("M,N,K,ldpad", [(70, 50, 48, 0), (300, 192, 192, 0), (257, 576, 192, 8), (33, 10, 24, 0),
                                          (130, 96, 4, 0), (64, 768, 192, 0), (5, 3, 2, 0), (128, 128, 33, 3)])
def test_linear_fwd(ops, gemm_mode, M, N, K, ldpad):
    xfull = rnd(M, K + ldpad, seed=1)
    x = xfull[:, :K]
    W, b = rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    ref = (x.double() @ W.double().T + b.double())
    xd = dev(xfull)[:, :K]
    out = torch.empty(M, N, device=DEV)
    ops.linear_fwd(xd, dev(W), dev(b), out)
    assert rel_err(out.cpu(), ref) < GEMM_TOL
    out2 = torch.empty(M, N, device=DEV)
    ops.linear_fwd(xd, dev(W), None, out2)
    assert rel_err(out2.cpu(), ref - b.double()) < GEMM_TOL


@pytest.mark.parametrize("M,N,K", [(130, 768, 192), (37, 64, 16), (70, 16, 4), (300, 64, 8)])
def test_linear_gelu_fwd(ops, gemm_mode, M, N, K):
    x, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3)
    pre_ref = (x.double() @ W.double().T + b.double()).requires_grad_(True)
    act_ref = F.gelu(pre_ref)
    act_ref.sum().backward()
    grad, act = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.linear_gelu_fwd(dev(x), dev(W), dev(b), grad, act)
    # single-exponential erf (A&S 7.1.26): absolute error <= ~2e-7 * max(1, |x|)
    # (fp32 GEMM rounding scales with |pre|, hence the relative part)
    assert float(((act.cpu().double() - act_ref.detach()).abs() / (1 + act_ref.detach().abs())).max()) < 5e-6
    assert float((grad.cpu().double() - pre_ref.grad).abs().max()) < 5e-6


@pytest.mark.parametrize("M,N,K,rmod", [(130, 192, 768, 130), (130, 96, 192, 65), (68, 4, 16, 17)])
def test_linear_residual_fwd(ops, gemm_mode, M, N, K, rmod):
    x, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    R = rnd(rmod, N, seed=4)
    ref = x.double() @ W.double().T + b.double() + R.double().repeat(M // rmod, 1)
    out = torch.empty(M, N, device=DEV)
    ops.linear_residual_fwd(dev(x), dev(W), dev(b), dev(R), rmod, out)
    assert rel_err(out.cpu(), ref) < GEMM_TOL


@pytest.mark.parametrize("M,N,K", [(130, 576, 192), (257, 192, 768), (33, 10, 24), (70, 12, 4), (300, 768, 192)])
def test_linear_bwd_input(ops, M, N, K):
    dy, W = rnd(M, N, seed=1), rnd(N, K, seed=2, scale=0.1)
    ref = dy.double() @ W.double()
    dx = torch.empty(M, K, device=DEV)
    ops.linear_bwd_input(dev(dy), dev(W), dx)
    assert rel_err(dx.cpu(), ref) < GEMM_TOL
    base = rnd(M, K, seed=5)
    dx2 = dev(base).clone()
    ops.linear_bwd_input(dev(dy), dev(W), dx2, accumulate=True)
    assert rel_err(dx2.cpu(), ref + base.double()) < GEMM_TOL
    gg = rnd(M, K, seed=6)
    dx3 = torch.empty(M, K, device=DEV)
    ops.linear_bwd_input(dev(dy), dev(W), dx3, gelu_grad=dev(gg))
    assert rel_err(dx3.cpu(), ref * gg.double()) < GEMM_TOL


@pytest.mark.parametrize("M,N,K", [(130, 576, 192), (257, 192, 768), (33, 12, 24), (70, 12, 4), (300, 768, 192), (4160, 768, 192)])
def test_linear_bwd_input_t(ops, gemm_mode, M, N, K):
    """dX = dY W from the transposed weight copy (vsom_transpose_many + vsom_linear_bwd_input_t)."""
    dy, W = rnd(M, N, seed=1), rnd(N, K, seed=2, scale=0.1)
    ref = dy.double() @ W.double()
    Wd = dev(W)
    Wt = torch.empty(K, N, device=DEV)
    table = torch.tensor([[0, 0, N, K]], dtype=torch.int64, device=DEV)
    ops.transpose_many(Wd, Wt.view(-1), table, N, K)
    assert torch.equal(Wt.cpu(), W.T.contiguous())
    tol = grad_tol(gemm_mode)
    dx = torch.empty(M, K, device=DEV)
    ops.linear_bwd_input_t(dev(dy), Wt, dx)
    assert rel_err(dx.cpu(), ref) < tol
    base = rnd(M, K, seed=5)
    dx2 = dev(base).clone()
    ops.linear_bwd_input_t(dev(dy), Wt, dx2, accumulate=True)
    assert rel_err(dx2.cpu(), ref + base.double()) < tol
    gg = rnd(M, K, seed=6)
    dx3 = torch.empty(M, K, device=DEV)
    ops.linear_bwd_input_t(dev(dy), Wt, dx3, gelu_grad=dev(gg))
    assert rel_err(dx3.cpu(), ref * gg.double()) < tol


def test_transpose_many_batched(ops):
    shapes = [(192, 576), (37, 5), (768, 192), (4, 4), (1, 33)]
    src = rnd(sum(r * c for r, c in shapes) + 64, seed=3)
    rows, so, do = [], 0, 0
    for r, c in shapes:
        rows.append([so, do, r, c]); so += r * c; do += r * c
    srcd, dst = dev(src), torch.zeros(do, device=DEV)
    ops.transpose_many(srcd, dst, torch.tensor(rows, dtype=torch.int64, device=DEV), 768, 576)
    for s0, d0, r, c in rows:
        assert torch.equal(dst[d0:d0 + r * c].view(c, r).cpu(), src[s0:s0 + r * c].view(r, c).T)


def test_split_bf16_wide_dynamic_range(ops):
    """The split keeps fp32's exponent range and all 24 significand bits: operands spanning 60
    orders of magnitude, and values with only low-order mantissa bits set, come out fp32-accurate."""
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(ops.GEMM_SPLIT_BF16)
    try:
        M, N, K = 96, 64, 128
        x = rnd(M, K, seed=1) * torch.logspace(-30, 30, K).float()[None, :]
        W = rnd(N, K, seed=2) * torch.logspace(25, -25, K).float()[None, :]
        out = torch.empty(M, N, device=DEV)
        ops.linear_fwd(dev(x), dev(W), None, out)
        assert rel_err(out.cpu(), x.double() @ W.double().T) < GEMM_TOL
        # 1 + 2^-23 and friends: the information sits in the last mantissa bits
        x2 = (1.0 + torch.arange(M * K).view(M, K).float() % 7 * 2.0 ** -23)
        W2 = torch.where(torch.arange(N * K).view(N, K) % 2 == 0, 1.0, -1.0).float()
        ops.linear_fwd(dev(x2), dev(W2), None, out)
        ref = x2.double() @ W2.double().T
        assert float((out.cpu().double() - ref).abs().max()) < 2e-6      # |ref| ~ 1e-6 .. 1e-5: cancellation survives
    finally:
        ops.set_gemm_mode(prev)


def test_gemm_mode_switch(ops):
    prev = ops.get_gemm_mode()
    assert prev == ops.GEMM_SPLIT_BF16_GRAD3                     # the default
    from vit_som_amd._lib import VsomError
    with pytest.raises(VsomError):
        ops.set_gemm_mode(7)
    assert ops.get_gemm_mode() == prev


@pytest.mark.parametrize("M,N,K", [(1000, 576, 192), (4160, 192, 768), (70, 10, 24), (33, 4, 16), (650, 48, 96),
                                    (2080, 768, 192), (650, 288, 96), (1301, 96, 384), (31, 192, 64), (33280, 192, 192),
                                    (33280, 576, 192)])
def test_linear_bwd_weight(ops, gemm_mode, M, N, K):
    dy, x = rnd(M, N, seed=1), rnd(M, K, seed=2)
    dW_ref, db_ref = dy.double().T @ x.double(), dy.double().sum(0)
    dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    ops.linear_bwd_weight(dev(dy), dev(x), dW, db)
    assert rel_err(dW.cpu(), dW_ref) < grad_tol(gemm_mode)
    assert rel_err(db.cpu(), db_ref) < GEMM_TOL                  # the bias gradient is a plain fp32 column sum in every mode
    # deterministic: bitwise identical on a second run
    dW2, db2 = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    ops.linear_bwd_weight(dev(dy), dev(x), dW2, db2)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)


@pytest.mark.parametrize("B,C,S,p,E", [(6, 3, 32, 4, 192), (4, 1, 28, 2, 16), (3, 3, 8, 4, 24)])
def test_patch_embed(ops, O, gemm_mode, B, C, S, p, E):
    n = (S // p) ** 2
    img = rnd(B, C, S, S, seed=1)
    W, b = rnd(E, C, p, p, seed=2, scale=0.2), rnd(E, seed=3)
    cls, pos = rnd(1, 1, E, seed=4), O.sincos_pos_embed(E, S // p)
    P = {"vit.patch_embed.proj.weight": W, "vit.patch_embed.proj.bias": b}
    ref = torch.cat(((cls + pos[:, :1]).expand(B, -1, -1), O.patch_embed(P, img, p) + pos[:, 1:]), dim=1)
    tokens = torch.empty(B, n + 1, E, device=DEV)
    xp = torch.empty(B * n, C * p * p, device=DEV)
    ops.patch_embed_fwd(dev(img), dev(W.reshape(E, -1).contiguous()), dev(b), dev(pos[0]), dev(cls.reshape(E)), tokens, xp, p)
    assert torch.allclose(tokens.cpu(), ref, atol=2e-6)
    # backward
    dt = rnd(B, n + 1, E, seed=5)
    Wl = W.clone().requires_grad_(True); bl = b.clone().requires_grad_(True); cl = cls.clone().requires_grad_(True)
    Pl = {"vit.patch_embed.proj.weight": Wl, "vit.patch_embed.proj.bias": bl}
    out = torch.cat(((cl + pos[:, :1]).expand(B, -1, -1), O.patch_embed(Pl, img, p) + pos[:, 1:]), dim=1)
    out.backward(dt)
    dW, db, dc = torch.empty(E, C * p * p, device=DEV), torch.empty(E, device=DEV), torch.empty(E, device=DEV)
    ops.patch_embed_bwd(dev(dt), xp, dW, db, dc, B, C, S, p, E)
    assert rel_err(dW.cpu(), Wl.grad.reshape(E, -1)) < 1e-5
    assert rel_err(db.cpu(), bl.grad) < 1e-5
    assert rel_err(dc.cpu(), cl.grad.reshape(E)) < 1e-5


@pytest.mark.parametrize("rows,cols", [(130, 192), (70, 96), (37, 16), (20, 4), (9, 768), (4163, 192), (77, 64), (50, 128), (33, 256)])
def test_layernorm(ops, rows, cols):
    x = rnd(rows, cols, seed=1) * 2 + 0.5
    g, b = 1 + 0.1 * rnd(cols, seed=2), 0.1 * rnd(cols, seed=3)
    xl, gl, bl = (t.clone().double().requires_grad_(True) for t in (x, g, b))
    ref = F.layer_norm(xl, (cols,), gl, bl, 1e-6)
    y = torch.empty(rows, cols, device=DEV); mean = torch.empty(rows, device=DEV); rstd = torch.empty(rows, device=DEV)
    ops.layernorm_fwd(dev(x), dev(g), dev(b), y, mean, rstd, 1e-6)
    assert torch.allclose(y.cpu().double(), ref, atol=3e-6)
    dy, resid = rnd(rows, cols, seed=4), rnd(rows, cols, seed=5)
    ref.backward(dy.double())
    dx = torch.empty(rows, cols, device=DEV); dg = torch.empty(cols, device=DEV); db = torch.empty(cols, device=DEV)
    ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(g), dev(resid), dx, dg, db)
    assert rel_err(dx.cpu(), xl.grad + resid.double()) < 5e-6
    assert rel_err(dg.cpu(), gl.grad) < 5e-6 and rel_err(db.cpu(), bl.grad) < 5e-6
    dx2 = torch.empty(rows, cols, device=DEV)
    ops.layernorm_bwd(dev(dy), dev(x), mean, rstd, dev(g), None, dx2, dg, db)
    assert rel_err(dx2.cpu(), xl.grad) < 5e-6


def attn_ref(qkv, B, N, H, hd):
    q, k, v = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    a = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    return (a @ v).transpose(1, 2).reshape(B, N, H * hd)


@pytest.mark.parametrize("B,N,H,hd", [(4, 65, 3, 64), (3, 17, 2, 8), (2, 37, 2, 2), (3, 5, 3, 4), (2, 65, 3, 32),
                                       (2, 197, 2, 8), (1, 257, 3, 64), (2, 16, 1, 16), (2, 33, 2, 3)])
def test_attention(ops, gemm_mode, B, N, H, hd):
    """Forward exact-f32 in every mode (3e-6 abs); backward 5e-6 relative, except that in the default mode the hd = 64
    short-sequence backward runs its products on the two-piece bf16 split like the gradient GEMMs (measured 8e-6)."""
    E = H * hd
    qkv = rnd(B, N, 3 * E, seed=1)
    q64 = qkv.double().requires_grad_(True)
    ref = attn_ref(q64, B, N, H, hd)
    out = torch.empty(B, N, E, device=DEV); lse = torch.empty(B, H, N, device=DEV)
    ops.attention_fwd(dev(qkv), out, lse, B, N, H, hd)
    assert torch.allclose(out.cpu().double(), ref, atol=3e-6), float((out.cpu().double() - ref).abs().max())
    dout = rnd(B, N, E, seed=2)
    ref.backward(dout.double())
    dqkv = torch.full((B, N, 3 * E), float("nan"), device=DEV); delta = torch.empty(B, H, N, device=DEV)
    ops.attention_bwd(dev(qkv), out, dev(dout), lse, dqkv, delta, B, N, H, hd)
    assert rel_err(dqkv.cpu(), q64.grad) < (GRAD3_TOL if gemm_mode == "grad3" and hd == 64 else 5e-6)


@pytest.mark.parametrize("B,N,H,hd", [(4, 65, 3, 64), (3, 64, 2, 64), (2, 50, 3, 64), (3, 33, 2, 64), (2, 17, 1, 64), (2, 65, 3, 32),
                                       (2, 49, 2, 16), (2, 197, 2, 8)])
def test_attention_backward_forms_give_the_same_bits(ops, gemm_mode, B, N, H, hd):
    """The short-sequence backward exists as two launches (hook 0), as one launch whose dK/dV phase re-uses the P and dS
    blocks of the dQ phase (1, the default where the shape allows) and as one launch that recomputes them (2): same
    arithmetic in the same order, so the gradients must be identical bit for bit -- incl. a ragged last tile (N = 50)
    and shapes on which the default falls back to the other forms.  (In the default GEMM mode form 1 runs its products on
    the two-piece bf16 split at hd = 64: equal to the others within that mode's tolerance, and still deterministic.)"""
    E = H * hd
    qkv, dout = dev(rnd(B, N, 3 * E, seed=11)), dev(rnd(B, N, E, seed=12))
    out = torch.empty(B, N, E, device=DEV); lse = torch.empty(B, H, N, device=DEV)
    ops.attention_fwd(qkv, out, lse, B, N, H, hd)
    got = []
    try:
        for mode in (0, 1, 2):
            ops.set_attention_fused(mode)
            dqkv = torch.full((B, N, 3 * E), float("nan"), device=DEV); delta = torch.full((B, H, N), float("nan"), device=DEV)
            ops.attention_bwd(qkv, out, dout, lse, dqkv, delta, B, N, H, hd)
            got.append((dqkv, delta))
    finally:
        ops.set_attention_fused(1)
    split_form = gemm_mode == "grad3" and hd == 64 and N <= 65
    for i, (dqkv, delta) in enumerate(got[1:], 1):
        if split_form and i == 1:
            # (the split form takes D = sum_j p_ij dP_ij from its own blocks instead of dO . O: same number, other rounding)
            assert rel_err(dqkv.cpu(), got[0][0].cpu().double()) < GRAD3_TOL and rel_err(delta.cpu(), got[0][1].cpu().double()) < GRAD3_TOL
        else:
            assert torch.equal(dqkv, got[0][0]) and torch.equal(delta, got[0][1])
    if split_form:                                   # the split form repeats bit for bit
        ops.set_attention_fused(1)
        again = torch.full((B, N, 3 * E), float("nan"), device=DEV)
        ops.attention_bwd(qkv, out, dout, lse, again, torch.empty(B, H, N, device=DEV), B, N, H, hd)
        assert torch.equal(again, got[1][0])


def test_attention_large_scores(ops):
    """online-softmax rescale path: a key in the SECOND 64-key chunk dominates."""
    B, N, H, hd = 1, 130, 1, 16
    qkv = rnd(B, N, 3 * hd, seed=3)
    qkv[0, 100, hd:2 * hd] *= 30.0
    q64 = qkv.double()
    ref = attn_ref(q64, B, N, H, hd)
    out = torch.empty(B, N, hd, device=DEV); lse = torch.empty(B, H, N, device=DEV)
    ops.attention_fwd(dev(qkv), out, lse, B, N, H, hd)
    assert torch.allclose(out.cpu().double(), ref, atol=1e-5)


def bmu_policy_ok(bmu, dist_ref64, eps=2e-6):
    """BMU exact wherever the fp64 reference's top-2 gap exceeds eps; inside a near-tie either
    of the tied prototypes is accepted (SURVEY.md hard part 3)."""
    ref = dist_ref64.argmin(1)
    srt = dist_ref64.sort(1).values
    gap = srt[:, 1] - srt[:, 0] if dist_ref64.shape[1] > 1 else torch.ones(dist_ref64.shape[0])
    chosen = dist_ref64.gather(1, bmu.view(-1, 1)).squeeze(1)
    exact = (bmu == ref)
    near = (chosen - srt[:, 0]) <= eps
    return bool(((exact) | (near & (gap <= eps))).all()), int((~exact).sum())


@pytest.mark.parametrize("B,K,L", [(70, 15, 256), (64, 576, 3136), (128, 16, 12288), (33, 100, 48), (5, 7, 20)])
def test_bmu_cosine(ops, O, B, K, L):
    x = rnd(B, L, seed=1)
    W = F.normalize(torch.rand(K, L, generator=torch.Generator().manual_seed(2)), dim=1)
    dist_ref, bmu_ref = O.som_forward(x, W)
    d64 = 1 - F.normalize(x.double(), dim=1) @ F.normalize(W.double(), dim=1).T
    xd, Wd = dev(x), dev(W)
    inx, inw = torch.empty(B, device=DEV), torch.empty(K, device=DEV)
    ops.row_inv_norm(xd, inx); ops.row_inv_norm(Wd, inw)
    assert torch.allclose(inx.cpu(), 1 / x.norm(dim=1), rtol=2e-6)
    dist, bmu = torch.empty(B, K, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
    ops.bmu_cosine_fwd(xd, Wd, inx, inw, dist, bmu)
    # distances: 1e-5 abs vs the fp32 oracle (north_star bar 1e-4); 2e-6 vs fp64
    assert torch.allclose(dist.cpu(), dist_ref, atol=1e-5)
    assert float((dist.cpu().double() - d64).abs().max()) < 2e-6
    # bmu is exactly the first argmin of the distances this kernel returned ...
    assert torch.equal(bmu.cpu(), dist.cpu().argmin(1))
    # ... and equals the reference BMU outside fp32 near-ties
    ok, nmis = bmu_policy_ok(bmu.cpu(), d64)
    assert ok, f"{nmis} BMU mismatches outside near-ties"
    bmu2 = torch.empty(B, dtype=torch.int64, device=DEV)
    ops.bmu_cosine_fwd(xd, Wd, inx, inw, None, bmu2)
    assert torch.equal(bmu, bmu2)


@pytest.mark.parametrize("B,K,L", [(70, 15, 256), (64, 576, 3136), (128, 16, 12288), (33, 100, 48), (5, 7, 20), (512, 1600, 12288),
                                    (130, 2048, 192), (512, 576, 12288), (256, 1600, 49152)])      # ... c3, c2 and c5 per GPU at full size
def test_bmu_cosine_x3_rerank(ops, B, K, L):
    """Reduced-precision contraction + exact re-rank: norms, distances (1e-5 abs vs fp64: the three-product error is
    <= 4.6e-5 in the worst case, ~1e-7 observed), bmu == first argmin of the returned distances, and bmu equal to the
    fp64 argmin outside the fp32 near-tie window -- the same policy as the exact-f32 pass."""
    x = rnd(B, L, seed=1)
    W = F.normalize(torch.rand(K, L, generator=torch.Generator().manual_seed(2)), dim=1)
    d64 = 1 - F.normalize(x.double(), dim=1) @ F.normalize(W.double(), dim=1).T
    xd, Wd = dev(x), dev(W)
    inx, inw = torch.empty(B, device=DEV), torch.empty(K, device=DEV)
    dist, bmu = torch.empty(B, K, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.bmu_cosine_x3_fwd(xd, Wd, dist, bmu, inx, inw, cnt)
    assert torch.allclose(inx.cpu().double(), 1 / x.double().norm(dim=1), rtol=2e-6)
    assert torch.allclose(inw.cpu().double(), 1 / W.double().norm(dim=1), rtol=2e-6)
    err = float((dist.cpu().double() - d64).abs().max())
    assert err < 1e-5, err
    assert torch.equal(bmu.cpu(), dist.cpu().argmin(1))
    ok, nmis = bmu_policy_ok(bmu.cpu(), d64)
    srt = d64.sort(1).values
    near = int(((srt[:, 1] - srt[:, 0]) <= 2e-6).sum()) if K > 1 else 0
    print(f"bmu_x3 B={B} K={K} L={L}: max |dist - fp64| {err:.2e}, rows re-ranked {int(cnt)}, "
          f"rows inside the 2e-6 near-tie window {near}, BMU != fp64 argmin on {nmis} rows")
    assert ok, f"{nmis} BMU mismatches outside near-ties"
    # deterministic, and identical without the distance output
    bmu2 = torch.empty(B, dtype=torch.int64, device=DEV)
    ops.bmu_cosine_x3_fwd(xd, Wd, None, bmu2, inx, inw)
    assert torch.equal(bmu, bmu2)


def test_bmu_cosine_x3_forced_rerank(ops):
    """Prototypes closer to each other than the contraction's error bound: the approximate minimum is ambiguous, the
    exact re-rank must pick the fp64 argmin (gaps of 3e-6 .. 3e-5 here, far outside fp32 noise), and exact ties the
    lowest index."""
    B, K, L = 16, 64, 1024
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, L, generator=g)
    W = torch.rand(K, L, generator=g)
    for i in range(B):                         # three near-copies of x_i at prototypes 3i+1 .. 3i+3, ever so slightly apart
        for j, eps in enumerate((2e-3, 2.5e-3, 3e-3)):
            W[3 * i + 1 + j] = x[i] + eps * torch.randn(L, generator=g)
    W[0] = x[5]; W[63] = x[5]                  # exact tie for sample 5 (identical rows) -> lowest index 0
    d64 = 1 - F.normalize(x.double(), dim=1) @ F.normalize(W.double(), dim=1).T
    inx, inw = torch.empty(B, device=DEV), torch.empty(K, device=DEV)
    dist, bmu = torch.empty(B, K, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.bmu_cosine_x3_fwd(dev(x), dev(W), dist, bmu, inx, inw, cnt)
    assert int(cnt) == B                       # every row had several candidates inside the window
    assert int(bmu[5]) == 0
    ref = d64.argmin(1)
    srt = d64.sort(1).values
    gap = srt[:, 1] - srt[:, 0]
    clear = gap > 2e-7                         # exact (fp64-accumulated) re-rank resolves gaps far below fp32 noise
    assert torch.equal(bmu.cpu()[clear], ref[clear]), (bmu.cpu(), ref, gap)
    assert torch.equal(bmu.cpu(), dist.cpu().argmin(1))
    assert float((dist.cpu().double() - d64).abs().max()) < 1e-5


def test_bmu_cosine_x3_more_candidates_than_one_chunk(ops):
    """More than 256 prototypes inside the re-rank window (collapsed prototypes late in training, or a dead input): the
    candidates are re-ranked 256 at a time, none is dropped -- the BMU is still the fp64 argmin, exact ties go to the lowest
    index, and a zero input row (all distances equal 1) yields index 0."""
    B, K, L = 6, 700, 512
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, L, generator=g)
    base = torch.randn(L, generator=g)
    W = base.unsqueeze(0) + 1e-4 * torch.randn(K, L, generator=g)        # 700 prototypes within ~1e-8 of each other in cosine distance
    W[650] = W[20]                                                        # an exact tie between two far-apart slots
    x[3] = 0.0                                                            # dead input: every distance is exactly 1
    d64 = 1 - F.normalize(x.double(), dim=1) @ F.normalize(W.double(), dim=1).T
    inx, inw = torch.empty(B, device=DEV), torch.empty(K, device=DEV)
    dist, bmu = torch.empty(B, K, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.bmu_cosine_x3_fwd(dev(x), dev(W), dist, bmu, inx, inw, cnt)
    assert int(cnt) == B
    assert torch.equal(bmu.cpu(), dist.cpu().argmin(1))
    assert int(bmu[3]) == 0
    live = torch.tensor([0, 1, 2, 4, 5])
    chosen = d64[live].gather(1, bmu.cpu()[live].view(-1, 1)).squeeze(1)
    assert float((chosen - d64[live].min(1).values).abs().max()) < 1.5e-7              # the exact distances are rounded to fp32 (ulp 6e-8 at 1.0) before they are compared
    assert float((dist.cpu().double() - d64).abs().max()) < 1e-5
    for i in live.tolist():                                                              # the tied pair never resolves to the higher slot
        assert int(bmu[i]) != 650


@pytest.mark.parametrize("B,K,L", [(512, 1600, 12288), (256, 576, 12288), (200, 70, 72), (192, 33, 8), (320, 1601, 1000)])
def test_bmu_cosine_x3_planes_equal_the_in_loop_split(ops, B, K, L):
    """The pre-split ("planes") form of the cosine BMU pass: partial-dot slabs bit-identical to the in-loop-split kernel's
    (same products, same order, same reduction split), norms to the last bits, hence distances to ~1e-7 and the same BMUs;
    ragged shapes (rows and prototypes that do not fill 32-row blocks, L that does not fill a 32-deep k-tile)."""
    if ops.get_gemm_mode() == ops.GEMM_F32:
        pytest.skip("the cosine BMU pass runs on the exact-f32 engine in this mode")
    assert ops.bmu_planes_supported(B, K, L)
    xd = dev(rnd(B, L + 8, seed=3))[:, :L]                                 # a row stride that is not L
    x = xd.cpu()
    W = F.normalize(torch.rand(K, L, generator=torch.Generator().manual_seed(4)), dim=1)
    Wd = dev(W)
    lib = ops.lib
    from vit_som_amd._lib import ptr, stream
    # slabs, both ways
    nb0, nb1 = lib.vsom_bmu_cosine_x3_workspace_bytes(B, K, L), lib.vsom_bmu_cosine_x3_planes_workspace_bytes(B, K, L)
    ws0, ws1 = torch.zeros(nb0, dtype=torch.uint8, device=DEV), torch.zeros(nb1, dtype=torch.uint8, device=DEV)
    xp, wp = ops.bmu_planes_alloc(B, L, DEV), ops.bmu_planes_alloc(K, L, DEV)
    ops.bmu_planes_from(xd, xp); ops.bmu_planes_from(Wd, wp)
    assert lib.vsom_bmu_cosine_x3_dots(ptr(xd), xd.stride(0), ptr(Wd), B, K, L, ptr(ws0), nb0, stream()) == 0
    assert lib.vsom_bmu_cosine_x3_planes_dots(ptr(xp), ptr(wp), B, K, L, ptr(ws1), nb1, stream()) == 0
    nslab = (nb1 - 16) // 4
    assert torch.equal(ws0.view(torch.float32)[:nslab], ws1.view(torch.float32)[:nslab])
    # whole pass
    out = []
    for planes in (False, True):
        inx, inw = torch.empty(B, device=DEV), torch.empty(K, device=DEV)
        dist, bmu = torch.empty(B, K, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
        cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
        if planes:
            ops.bmu_cosine_x3_planes_fwd(xd, Wd, xp, wp, dist, bmu, inx, inw, cnt)
        else:
            ops.bmu_cosine_x3_fwd(xd, Wd, dist, bmu, inx, inw, cnt)
        out.append((dist.cpu(), bmu.cpu(), inx.cpu(), inw.cpu(), int(cnt)))
    (d0, b0, ix0, iw0, c0), (d1, b1, ix1, iw1, c1) = out
    assert torch.allclose(ix0, ix1, rtol=1e-6) and torch.allclose(iw0, iw1, rtol=1e-6)
    assert float((d0 - d1).abs().max()) < 5e-7
    assert torch.equal(b1, d1.argmin(1))
    d64 = 1 - F.normalize(x.double(), dim=1) @ F.normalize(W.double(), dim=1).T
    assert float((d1.double() - d64).abs().max()) < 1e-5
    ok, nmis = bmu_policy_ok(b1, d64)
    assert ok, f"{nmis} BMU mismatches outside near-ties"
    same = int((b0 == b1).sum())
    print(f"bmu planes B={B} K={K} L={L}: slabs bit-identical, max |dist planes - dist in-loop| {float((d0 - d1).abs().max()):.1e}, "
          f"same BMU on {same}/{B} rows, re-ranked {c0} / {c1}")
    assert same >= B - 1


def test_adamw_step_planes_is_the_flat_step_plus_the_split(ops):
    """vsom_adamw_step_planes: parameters and moments bitwise those of the flat kernel over the whole arena, and the plane
    buffer of the slice bytewise the one vsom_bmu_planes_from writes from the UPDATED values."""
    R, L = 70, 136                                   # 9520 elements: the slice ends inside a 256-chunk (padding follows)
    lead, tail = 256 * 3, 256 * 2
    padded = (R * L + 255) // 256 * 256
    n = lead + padded + tail
    g0 = torch.Generator().manual_seed(0)
    p, gr = torch.randn(n, generator=g0), torch.randn(n, generator=g0)
    p[lead + R * L:lead + padded] = 0; gr[lead + R * L:lead + padded] = 0
    wd = torch.rand(n // 256, generator=g0) * 0.05
    pa, pb = dev(p).clone(), dev(p).clone()
    ma, va, mb, vb = (torch.zeros(n, device=DEV) for _ in range(4))
    planes, ref = ops.bmu_planes_alloc(R, L, DEV), ops.bmu_planes_alloc(R, L, DEV)
    planes.zero_(); ref.zero_()
    for step in range(1, 4):
        for adamw in (True, False):
            ops.adamw_step(pa, dev(gr), ma, va, dev(wd), 3e-3, 0.9, 0.999, 1e-8, step, grad_scale=0.5, adamw=adamw)
            ops.adamw_step(pb, dev(gr), mb, vb, dev(wd), 3e-3, 0.9, 0.999, 1e-8, step, grad_scale=0.5, adamw=adamw,
                           planes=(lead, R, L, planes))
            assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), (step, adamw)
            ops.bmu_planes_from(pb[lead:lead + R * L].view(R, L), ref)
            assert torch.equal(planes, ref), (step, adamw)
    # argument checks: a slice that is not chunk-aligned, a short plane buffer
    with pytest.raises(Exception):
        ops.adamw_step(pb, dev(gr), mb, vb, dev(wd), 3e-3, 0.9, 0.999, 1e-8, 1, planes=(lead + 4, R, L, planes))
    with pytest.raises(Exception):
        ops.adamw_step(pb, dev(gr), mb, vb, dev(wd), 3e-3, 0.9, 0.999, 1e-8, 1, planes=(lead, R, L, planes[:1024]))


def test_bmu_exact_ties_pick_lowest_index(ops):
    B, K, L = 8, 40, 64
    x = rnd(B, L, seed=1)
    W = torch.rand(K, L, generator=torch.Generator().manual_seed(2))
    W[7] = x[0]; W[3] = x[0]; W[30] = x[0]          # three identical prototypes: exact tie for sample 0
    xd, Wd = dev(x), dev(W)
    inx, inw = torch.empty(B, device=DEV), torch.empty(K, device=DEV)
    ops.row_inv_norm(xd, inx); ops.row_inv_norm(Wd, inw)
    dist, bmu = torch.empty(B, K, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
    ops.bmu_cosine_fwd(xd, Wd, inx, inw, dist, bmu)
    assert int(bmu[0]) == 3
    assert torch.equal(bmu.cpu(), dist.cpu().argmin(1))


@pytest.mark.parametrize("B,K,L,map_size,topo", [(70, 15, 256, (3, 5), "square"), (64, 576, 3136, (24, 24), "square"),
                                                  (33, 12, 48, (4, 3), "hexa"),
                                                  (128, 576, 3072, (24, 24), "square")])     # the last: weight-gradient-engine form
def test_som_neigh_loss_and_bwd(ops, O, gemm_mode, B, K, L, map_size, topo):
    Nrow = L + 8
    xfull = rnd(B, Nrow, seed=1)
    x = xfull[:, 8:]
    W = F.normalize(torch.rand(K, L, generator=torch.Generator().manual_seed(2)), dim=1)
    grid = O.grid_positions(map_size, topo)
    T, gam = 1.7, 0.37
    xl, Wl = x.clone().double().requires_grad_(True), W.clone().double().requires_grad_(True)
    d_ref = O.som_distances(xl, Wl)
    bmu_ref = d_ref.argmin(1)
    h_ref = O.neighbourhood(bmu_ref, grid.double(), T)
    loss_ref = O.som_loss(h_ref, d_ref)
    (gam * loss_ref).backward()

    xd = dev(xfull)[:, 8:]; Wd = dev(W)
    inx, inw = torch.empty(B, device=DEV), torch.empty(K, device=DEV)
    ops.row_inv_norm(xd, inx); ops.row_inv_norm(Wd, inw)
    dist, bmu = torch.empty(B, K, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
    ops.bmu_cosine_fwd(xd, Wd, inx, inw, dist, bmu)
    assert torch.equal(bmu.cpu(), bmu_ref)
    h = torch.empty(B, K, device=DEV); loss = torch.zeros(1, device=DEV)
    coef = torch.empty(B, K, device=DEV); rd = torch.empty(B, device=DEV); cd = torch.empty(K, device=DEV)
    ops.som_neigh_loss(dist, bmu, dev(grid), T, loss, h=h, inv_nx=inx, inv_nw=inw, grad_scale=gam / (B * K), coef=coef,
                       row_dot=rd, col_dot=cd)
    assert torch.allclose(h.cpu().double(), h_ref, atol=2e-6)
    assert abs(float(loss) / (B * K) - float(loss_ref.detach())) < 1e-6
    gW = torch.empty(K, L, device=DEV)
    base = rnd(B, Nrow, seed=9) * 1e-5          # same magnitude as the gradient: no fp32 cancellation
    gXfull = dev(base).clone()
    ops.som_bwd(xd, Wd, coef, rd, cd, gW, gXfull[:, 8:], accumulate_gx=True)
    # gradients compared relatively (they carry 1/(B K))
    assert rel_err(gW.cpu(), Wl.grad) < 2e-5
    assert rel_err(gXfull.cpu()[:, 8:].double() - base[:, 8:].double(), xl.grad) < 2e-5
    assert torch.equal(gXfull.cpu()[:, :8], base[:, :8])
    # forward-only form
    loss2 = torch.zeros(1, device=DEV)
    ops.som_neigh_loss(dist, bmu, dev(grid), T, loss2)
    assert float(loss2) == float(loss)


@pytest.mark.parametrize("B,C,S,p", [(6, 3, 32, 4), (4, 1, 28, 2), (3, 3, 8, 4)])
def test_l1_unpatchify(ops, O, B, C, S, p):
    n = (S // p) ** 2
    pred = rnd(B, n + 1, p * p * C, seed=1)
    img = rnd(B, C, S, S, seed=2)
    pl = pred.clone().requires_grad_(True)
    recon_ref = O.unpatchify(pl[:, 1:, :], p)
    loss_ref = F.l1_loss(recon_ref, img)
    loss_ref.backward()
    recon = torch.empty(B, C, S, S, device=DEV); loss = torch.zeros(1, device=DEV); dpred = torch.empty_like(pred, device=DEV)
    ops.l1_unpatchify(dev(pred), dev(img), loss, recon=recon, dpred=dpred, grad_scale=1.0 / img.numel(), p=p)
    assert torch.equal(recon.cpu(), recon_ref.detach())
    assert abs(float(loss) / img.numel() - float(loss_ref)) < 1e-6
    assert torch.allclose(dpred.cpu(), pl.grad, atol=1e-9)


@pytest.mark.parametrize("B,C", [(64, 10), (33, 100), (5, 200), (7, 5)])
def test_cross_entropy_ls(ops, B, C):
    z = rnd(B, C, seed=1) * 3
    y = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(2))
    zl = z.clone().double().requires_grad_(True)
    ref = F.cross_entropy(zl, y, label_smoothing=0.1)
    ref.backward()
    loss = torch.zeros(1, device=DEV); dz = torch.empty(B, C, device=DEV)
    ops.cross_entropy_ls(dev(z), dev(y), 0.1, loss, dlogits=dz, grad_scale=1.0 / B)
    assert abs(float(loss) / B - float(ref)) < 2e-6
    assert rel_err(dz.cpu(), zl.grad) < 5e-6


def test_layernorm_bwd_in_two_halves_gives_the_same_bits(ops):
    """vsom_layernorm_bwd_partial + ONE vsom_layernorm_bwd_finish_many for several LayerNorms of different widths, flushed in
    two groups: dX, dgamma, dbeta bit for bit those of vsom_layernorm_bwd; a second pass reuses the job table."""
    shapes = [(33280, 192), (33280, 96), (8320, 192), (4160, 64)]
    data = []
    for i, (rows, cols) in enumerate(shapes):
        assert ops.layernorm_bwd_deferrable(rows, cols)
        x, dy, res = dev(rnd(rows, cols, seed=3 * i)), dev(rnd(rows, cols, seed=3 * i + 1)), dev(rnd(rows, cols, seed=3 * i + 2))
        gam = dev(rnd(cols, seed=50 + i))
        y, mean, rstd = torch.empty_like(x), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
        ops.layernorm_fwd(x, gam, torch.zeros(cols, device=DEV), y, mean, rstd, 1e-6)
        ref = [torch.empty_like(x), torch.empty(cols, device=DEV), torch.empty(cols, device=DEV)]
        ops.layernorm_bwd(dy, x, mean, rstd, gam, res if i % 2 else None, *ref)
        data.append((dy, x, mean, rstd, gam, res if i % 2 else None, ref))
    jobs = ops.LayerNormJobs(DEV)
    for rep in range(2):
        outs = [[torch.full_like(d[1], 7.0), torch.full((d[1].shape[1],), 7.0, device=DEV), torch.full((d[1].shape[1],), 7.0, device=DEV)]
                for d in data] if rep == 0 else outs
        for o in outs:
            for t in o:
                t.fill_(7.0)
        jobs.begin()
        for k, (d, o) in enumerate(zip(data, outs)):
            jobs.bwd(*d[:6], *o)
            if k == 1:
                jobs.flush()
        jobs.flush()
        assert jobs.n == jobs.flushed == len(shapes)
        for d, o in zip(data, outs):
            for a, b in zip(d[6], o):
                assert torch.equal(a, b), rep
    assert not ops.layernorm_bwd_deferrable(100, 192)             # too few workgroups for the wide reducer: the one-call form


def test_adamw_step(ops):
    n = 256 * 5
    g0 = torch.Generator().manual_seed(0)
    p, gr = torch.randn(n, generator=g0), torch.randn(n, generator=g0)
    wd_chunk = torch.tensor([0.05, 0.0, 0.01, 0.05, 0.0])
    leaves = [p[256 * i:256 * (i + 1)].clone().requires_grad_(True) for i in range(5)]
    opt = torch.optim.AdamW([{"params": [l], "weight_decay": float(w)} for l, w in zip(leaves, wd_chunk)], lr=3e-3,
                            betas=(0.9, 0.999))
    pd, gd = dev(p).clone(), dev(gr)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 5):
        for i, l in enumerate(leaves):
            l.grad = gr[256 * i:256 * (i + 1)].clone() * 0.5       # grad_scale = 0.5
        opt.step()
        ops.adamw_step(pd, gd, m, v, dev(wd_chunk), 3e-3, 0.9, 0.999, 1e-8, step, grad_scale=0.5)
        ref = torch.cat([l.detach() for l in leaves])
        assert torch.allclose(pd.cpu(), ref, atol=2e-7), step


def test_fill_and_reduce(ops):
    t = torch.empty(1000, device=DEV)
    ops.fill(t, 2.5)
    assert bool((t == 2.5).all())
    for ns, n in [(7, 1003), (40, 192), (33, 5000), (64, 147456), (3, 2)]:
        slabs = rnd(ns, n, seed=1)
        out = torch.empty(n, device=DEV)
        ops.reduce_slabs(dev(slabs), out)
        assert torch.allclose(out.cpu(), slabs.double().sum(0).float(), atol=2e-5), (ns, n)


def test_errors_are_loud(ops):
    from vit_som_amd._lib import VsomError
    x = torch.empty(4, 2048, device=DEV)
    with pytest.raises(VsomError):
        ops.layernorm_fwd(x, x[0], x[0], torch.empty_like(x), torch.empty(4, device=DEV), torch.empty(4, device=DEV))
    with pytest.raises(ValueError):
        ops.linear_fwd(torch.empty(4, 4), torch.empty(4, 4), None, torch.empty(4, 4))   # CPU tensors


@pytest.mark.parametrize("B,K,L,map_size,topo", [(64, 100, 3136, (10, 10), "square"), (33, 12, 48, (4, 3), "hexa"),
                                                  (64, 64, 384, (8, 8), "square")])
def test_som_euclidean_fwd_and_bwd(ops, O, gemm_mode, B, K, L, map_size, topo):
    """euclidean distance variant (torch.cdist p=2) + its SOM-loss gradients."""
    x = rnd(B, L, seed=1)
    W = torch.rand(K, L, generator=torch.Generator().manual_seed(2))
    grid = O.grid_positions(map_size, topo)
    T, gam = 2.3, 0.5
    xl, Wl = x.clone().double().requires_grad_(True), W.clone().double().requires_grad_(True)
    d_ref = torch.cdist(xl, Wl, p=2)
    bmu_ref = d_ref.argmin(1)
    loss_ref = O.som_loss(O.neighbourhood(bmu_ref, grid.double(), T), d_ref)
    (gam * loss_ref).backward()
    xd, Wd = dev(x), dev(W)
    sx, sw = torch.empty(B, device=DEV), torch.empty(K, device=DEV)
    ops.row_sqnorm(xd, sx); ops.row_sqnorm(Wd, sw)
    assert torch.allclose(sx.cpu(), (x * x).sum(1), rtol=2e-6)
    dist, bmu = torch.empty(B, K, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
    ops.bmu_euclid_fwd(xd, Wd, sx, sw, dist, bmu)
    assert torch.allclose(dist.cpu().double(), d_ref.detach(), atol=1e-4, rtol=1e-6)     # matmul form: |x|^2-scale cancellation
    assert torch.equal(bmu.cpu(), dist.cpu().argmin(1)) and torch.equal(bmu.cpu(), bmu_ref)
    loss = torch.zeros(1, device=DEV)
    coef = torch.empty(B, K, device=DEV); rd = torch.empty(B, device=DEV); cd = torch.empty(K, device=DEV)
    ops.som_neigh_loss(dist, bmu, dev(grid), T, loss, grad_scale=gam / (B * K), coef=coef, row_dot=rd, col_dot=cd,
                       distance=ops.DIST_EUCLIDEAN)
    assert abs(float(loss) / (B * K) - float(loss_ref.detach())) < 1e-5 * float(loss_ref.detach())
    gW, gX = torch.empty(K, L, device=DEV), torch.zeros(B, L, device=DEV)
    ops.som_bwd(xd, Wd, coef, rd, cd, gW, gX, accumulate_gx=False)
    assert rel_err(gW.cpu(), Wl.grad) < 5e-5 and rel_err(gX.cpu(), xl.grad) < 5e-5


@pytest.mark.parametrize("B,K,L,map_size,topo", [(64, 100, 3136, (10, 10), "square"), (33, 12, 48, (4, 3), "hexa"),
                                                  (5, 7, 21, (7, 1), "square"), (130, 70, 200, (10, 7), "square")])
def test_som_manhattan_fwd_and_bwd(ops, O, B, K, L, map_size, topo):
    """manhattan distance variant (torch.cdist p=1, som_layer.py:115-116) + its SOM-loss gradients
    (SURVEY 8(f) N4); ragged shapes exercise the scalar tails (L = 21 is not a multiple of 4)."""
    x = rnd(B, L, seed=1)
    W = torch.rand(K, L, generator=torch.Generator().manual_seed(2))
    x[0, :3] = W[1, :3]                                   # exact zeros of x - w: sign(0) = 0 like torch
    grid = O.grid_positions(map_size, topo)
    T, gam = 2.3, 0.5
    xl, Wl = x.clone().double().requires_grad_(True), W.clone().double().requires_grad_(True)
    d_ref = torch.cdist(xl, Wl, p=1)
    bmu_ref = d_ref.argmin(1)
    loss_ref = O.som_loss(O.neighbourhood(bmu_ref, grid.double(), T), d_ref)
    (gam * loss_ref).backward()
    xd, Wd = dev(x), dev(W)
    dist, bmu = torch.empty(B, K, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
    ops.bmu_manhattan_fwd(xd, Wd, dist, bmu)
    assert rel_err(dist.cpu(), d_ref.detach()) < 2e-6
    assert torch.equal(bmu.cpu(), dist.cpu().argmin(1)) and torch.equal(bmu.cpu(), bmu_ref)
    bmu2 = torch.empty(B, dtype=torch.int64, device=DEV)
    ops.bmu_manhattan_fwd(xd, Wd, None, bmu2)             # index-only form
    assert torch.equal(bmu2, bmu)
    loss = torch.zeros(1, device=DEV)
    coef = torch.empty(B, K, device=DEV)
    ops.som_neigh_loss(dist, bmu, dev(grid), T, loss, grad_scale=gam / (B * K), coef=coef, distance=ops.DIST_MANHATTAN)
    assert abs(float(loss) / (B * K) - float(loss_ref.detach())) < 1e-5 * float(loss_ref.detach())
    gW, gX = torch.empty(K, L, device=DEV), torch.zeros(B, L, device=DEV)
    ops.som_bwd_manhattan(xd, Wd, coef, gW, gX, accumulate_gx=False)
    assert rel_err(gW.cpu(), Wl.grad) < 5e-6 and rel_err(gX.cpu(), xl.grad) < 5e-6
    base = rnd(B, L, seed=9)
    gX2 = dev(base).clone()
    ops.som_bwd_manhattan(xd, Wd, coef, gW, gX2, accumulate_gx=True)
    assert rel_err(gX2.cpu(), xl.grad + base.double()) < 5e-6


@pytest.mark.parametrize("M,N,K", [(128, 500, 784), (37, 64, 16), (70, 10, 2000), (256, 2000, 500)])
def test_linear_relu_fwd(ops, gemm_mode, M, N, K):
    """nn.Linear + nn.ReLU of the DESOM autoencoder (ae.py:44-59): activation and its 0/1 derivative."""
    x, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    pre = x.double() @ W.double().T + b.double()
    der, act = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.linear_relu_fwd(dev(x), dev(W), dev(b), der, act)
    assert rel_err(act.cpu(), pre.clamp_min(0)) < GEMM_TOL
    sure = pre.abs() > 1e-4                                  # away from the kink the mask is exact
    assert torch.equal(der.cpu()[sure], (pre > 0).float()[sure])
    assert set(der.cpu().unique().tolist()) <= {0.0, 1.0}


@pytest.mark.parametrize("n", [7, 1024, 128 * 784 + 3])
def test_l1_loss(ops, n):
    p, t = rnd(n, seed=1), rnd(n, seed=2)
    p[:2] = t[:2]                                            # sign(0) = 0
    loss = torch.zeros(1, device=DEV)
    dp = torch.empty(n, device=DEV)
    ops.l1_loss(dev(p), dev(t), loss, dpred=dp, grad_scale=0.25)
    assert abs(float(loss) - float((p.double() - t.double()).abs().sum())) < 1e-5 * n ** 0.5 + 1e-6
    assert torch.equal(dp.cpu(), 0.25 * torch.sign(p - t))
    loss2 = torch.zeros(1, device=DEV)
    ops.l1_loss(dev(p), dev(t), loss2)
    assert float(loss2) == float(loss)
