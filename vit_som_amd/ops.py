"""Tensor-level wrappers over the C-ABI (one per entry point of include/vitsom_hip.h).

Every wrapper validates device / dtype / inner-stride, passes raw device pointers, row strides
and torch's current HIP stream, and raises ``VsomError`` on a non-zero status.  Nothing here
computes: there is no eager fallback.
"""
from typing import Optional

import torch

from . import _lib as _lib_mod
from ._lib import check, lib, ptr, stream

_scratch = {}
_retired = []

# Optional per-op HIP-event timing (bench.py): name -> list of (start_event, end_event) recorded
# on torch's current stream, which is the stream every kernel here is launched on.
_timers = {}


def enable_timer(name: str):
    _timers[name] = []


def disable_timers():
    _timers.clear()


def timer_ms(name: str):
    """Average milliseconds per timed call (synchronises)."""
    ev = _timers.get(name, [])
    if not ev:
        return None
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / len(ev), len(ev)


def reset_timer(name: str):
    if name in _timers:
        _timers[name] = []


def scratch(nbytes: int, device) -> torch.Tensor:
    """Grow-only byte scratch per (device, stream): reuse is ordered by the stream the kernels run on
    (256-byte aligned by the allocator)."""
    key = (torch.device(device).index or 0, stream() if torch.device(device).type == "cuda" else 0)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            # The old block may still be read by kernels queued on a side stream the caching allocator knows nothing
            # about (launches go to raw stream handles): never hand it back while the process lives (grow-only, a few
            # blocks per stream at most).
            _retired.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


def _f32(t: torch.Tensor, name: str, inner_contig: bool = True):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a HIP device tensor (vit_som_amd has no CPU path)")
    if t.dtype != torch.float32:
        raise ValueError(f"{name}: expected float32, got {t.dtype}")
    if inner_contig and t.dim() >= 1 and t.numel() > 0 and t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost stride must be 1")
    return t


def _rows(t: torch.Tensor) -> int:
    return t.stride(0) if t.dim() == 2 and t.shape[0] > 1 else t.shape[-1]


# ---------------------------------------------------------------- linear
def linear_fwd(x, W, bias, out):
    M, K = x.shape
    N = W.shape[0]
    _f32(x, "x"); _f32(W, "W"); _f32(out, "out")
    assert W.shape[1] == K and W.is_contiguous() and out.shape == (M, N)
    check(lib.vsom_linear_fwd(ptr(x), _rows(x), ptr(W), ptr(bias), ptr(out), _rows(out), M, N, K, stream()), "vsom_linear_fwd")
    return out


def linear_gelu_fwd(x, W, bias, grad, act):
    """grad <- gelu'(x W^T + b), act <- gelu(x W^T + b)."""
    pre = grad
    M, K = x.shape
    N = W.shape[0]
    _f32(x, "x"); _f32(W, "W")
    assert W.is_contiguous() and pre.is_contiguous() and act.is_contiguous() and pre.shape == (M, N) == act.shape
    check(lib.vsom_linear_gelu_fwd(ptr(x), _rows(x), ptr(W), ptr(bias), ptr(pre), ptr(act), M, N, K, stream()), "vsom_linear_gelu_fwd")
    return pre, act


def linear_relu_fwd(x, W, bias, ygrad, yact):
    """yact = relu(x W^T + b); ygrad = 1.0 where the pre-activation is positive (its derivative)."""
    M, K = x.shape
    N = W.shape[0]
    _f32(x, "x"); _f32(W, "W"); _f32(ygrad, "ygrad"); _f32(yact, "yact")
    assert W.is_contiguous() and W.shape[1] == K and ygrad.is_contiguous() and yact.is_contiguous()
    assert ygrad.shape == (M, N) and yact.shape == (M, N)
    check(lib.vsom_linear_relu_fwd(ptr(x), _rows(x), ptr(W), ptr(bias), ptr(ygrad), ptr(yact), M, N, K, stream()),
          "vsom_linear_relu_fwd")
    return yact


def l1_loss(pred, target, loss_sum, dpred=None, grad_scale=0.0):
    """loss_sum[0] = sum |pred - target|; dpred = grad_scale * sign(pred - target)."""
    assert pred.is_contiguous() and target.is_contiguous() and pred.numel() == target.numel()
    _f32(pred, "pred"); _f32(target, "target")
    n = pred.numel()
    ws = scratch(lib.vsom_l1_loss_workspace_bytes(n), pred.device)
    check(lib.vsom_l1_loss(ptr(pred), ptr(target), ptr(loss_sum), ptr(dpred), float(grad_scale), n, ptr(ws), ws.numel(),
                           stream()), "vsom_l1_loss")
    return loss_sum


def linear_residual_fwd(x, W, bias, R, r_mod, out):
    M, K = x.shape
    N = W.shape[0]
    _f32(x, "x"); _f32(W, "W"); _f32(R, "R"); _f32(out, "out")
    assert W.is_contiguous() and out.shape == (M, N) and R.shape[-1] == N
    check(lib.vsom_linear_residual_fwd(ptr(x), _rows(x), ptr(W), ptr(bias), ptr(R), _rows(R), int(r_mod), ptr(out),
                                       _rows(out), M, N, K, stream()), "vsom_linear_residual_fwd")
    return out


def linear_bwd_input(dy, W, dx, accumulate=False, gelu_grad=None):
    gelu_pre = gelu_grad
    M, N = dy.shape
    K = W.shape[1]
    _f32(dy, "dy"); _f32(W, "W"); _f32(dx, "dx")
    assert W.shape[0] == N and W.is_contiguous() and dx.shape == (M, K)
    if gelu_pre is not None:
        assert gelu_pre.is_contiguous() and gelu_pre.shape == (M, K)
    check(lib.vsom_linear_bwd_input(ptr(dy), _rows(dy), ptr(W), ptr(dx), _rows(dx), M, N, K, int(accumulate),
                                    ptr(gelu_pre), stream()), "vsom_linear_bwd_input")
    return dx


def linear_bwd_input_t(dy, Wt, dx, accumulate=False, gelu_grad=None):
    """dX (+)= dY * W from the transposed copy Wt[K,N] (see transpose_many)."""
    M, N = dy.shape
    K = Wt.shape[0]
    _f32(dy, "dy"); _f32(Wt, "Wt"); _f32(dx, "dx")
    assert Wt.shape[1] == N and Wt.is_contiguous() and dx.shape == (M, K)
    if gelu_grad is not None:
        assert gelu_grad.is_contiguous() and gelu_grad.shape == (M, K)
    check(lib.vsom_linear_bwd_input_t(ptr(dy), _rows(dy), ptr(Wt), ptr(dx), _rows(dx), M, N, K, int(accumulate),
                                      ptr(gelu_grad), stream()), "vsom_linear_bwd_input_t")
    return dx


def transpose_many(src_base, dst_base, table, max_rows, max_cols):
    """table: device int64 [count,4] rows {src_off, dst_off, rows, cols} (offsets in floats)."""
    _f32(src_base, "src_base"); _f32(dst_base, "dst_base")
    assert table.dtype == torch.int64 and table.is_contiguous() and table.ndim == 2 and table.shape[1] == 4
    assert table.device == src_base.device == dst_base.device
    check(lib.vsom_transpose_many(ptr(src_base), ptr(dst_base), table.data_ptr(), table.shape[0], int(max_rows),
                                  int(max_cols), stream()), "vsom_transpose_many")
    return dst_base


GEMM_F32, GEMM_SPLIT_BF16, GEMM_SPLIT_BF16_GRAD3 = 0, 1, 2      # include/vitsom_hip.h: VSOM_GEMM_*


def set_gemm_mode(mode: int):
    check(lib.vsom_set_gemm_mode(int(mode)), "vsom_set_gemm_mode")


def get_gemm_mode() -> int:
    return int(lib.vsom_get_gemm_mode())


def linear_bwd_weight(dy, x, dW, db):
    M, N = dy.shape
    K = x.shape[1]
    _f32(dy, "dy"); _f32(x, "x"); _f32(dW, "dW")
    assert x.shape[0] == M and dW.is_contiguous() and dW.numel() == N * K
    nbytes = lib.vsom_linear_bwd_weight_workspace_bytes(M, N, K)
    ws = scratch(nbytes, dy.device)
    check(lib.vsom_linear_bwd_weight(ptr(dy), _rows(dy), ptr(x), _rows(x), ptr(dW), ptr(db), M, N, K, ptr(ws), ws.numel(),
                                     stream()), "vsom_linear_bwd_weight")
    return dW, db


# ---------------------------------------------------------------- patch embedding
def patch_embed_fwd(img, Wpe, bpe, pos, cls_token, tokens, xp_ws, p):
    B, Cc, S, _ = img.shape
    E = Wpe.shape[0]
    for n_, t in (("img", img), ("Wpe", Wpe), ("tokens", tokens), ("xp_ws", xp_ws)):
        _f32(t, n_)
        assert t.is_contiguous(), n_
    check(lib.vsom_patch_embed_fwd(ptr(img), ptr(Wpe), ptr(bpe), ptr(pos), ptr(cls_token), ptr(tokens), ptr(xp_ws), B, Cc, S,
                                   p, E, stream()), "vsom_patch_embed_fwd")
    return tokens


def patch_embed_bwd(dtokens, xp_ws, dWpe, dbpe, dcls, B, Cc, S, p, E):
    assert dtokens.is_contiguous() and xp_ws.is_contiguous()
    nbytes = lib.vsom_patch_embed_bwd_workspace_bytes(B, Cc, S, p, E)
    ws = scratch(nbytes, dtokens.device)
    check(lib.vsom_patch_embed_bwd(ptr(dtokens), ptr(xp_ws), ptr(dWpe), ptr(dbpe), ptr(dcls), B, Cc, S, p, E, ptr(ws),
                                   ws.numel(), stream()), "vsom_patch_embed_bwd")


# ---------------------------------------------------------------- layernorm
def layernorm_fwd(x, gamma, beta, y, mean, rstd, eps=1e-6):
    rows, cols = x.shape
    assert x.is_contiguous() and y.is_contiguous()
    _f32(x, "x")
    check(lib.vsom_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, cols, float(eps), stream()),
          "vsom_layernorm_fwd")
    return y


def layernorm_bwd(dy, x, mean, rstd, gamma, resid, dx, dgamma, dbeta):
    rows, cols = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and dx.is_contiguous() and (resid is None or resid.is_contiguous())
    nbytes = lib.vsom_layernorm_bwd_workspace_bytes(rows, cols)
    ws = scratch(nbytes, x.device)
    check(lib.vsom_layernorm_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(resid), ptr(dx), ptr(dgamma),
                                 ptr(dbeta), rows, cols, ptr(ws), ws.numel(), stream()), "vsom_layernorm_bwd")
    return dx


class LayerNormJobs:
    """The LayerNorm backwards of one pass in deferred form: `bwd` computes dX and leaves the column partials in a buffer
    of its own, `flush` produces dgamma / dbeta for everything since the last flush in ONE launch (bit for bit what
    layernorm_bwd writes).  The job table is fixed after the first pass (same calls in the same order, same buffers);
    a call that does not match it (other shape, other order) starts a new table."""

    def __init__(self, device):
        self.device = device
        self.keys, self.rows_host, self.parts = [], [], []
        self.table = None               # int64 [njobs, 4] on the device, built once the first pass is complete
        self.n = self.flushed = 0
        self.max_cols = 0

    def begin(self):
        self.n = self.flushed = 0

    def bwd(self, dy, x, mean, rstd, gamma, resid, dx, dgamma, dbeta):
        rows, cols = x.shape
        key = (ptr(dgamma), ptr(dbeta), rows, cols)
        i = self.n
        if i < len(self.keys) and self.keys[i] != key:           # the pass changed: forget the table from here on
            del self.keys[i:], self.rows_host[i:], self.parts[i:]
            self.table = None
        if i == len(self.keys):
            nbytes = lib.vsom_layernorm_bwd_workspace_bytes(rows, cols)
            part = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.keys.append(key); self.parts.append(part)
            self.rows_host.append([ptr(part), ptr(dgamma), ptr(dbeta), ((nbytes // (8 * cols)) << 32) | cols])
            self.max_cols = max(self.max_cols, cols)
            self.table = None
        part = self.parts[i]
        check(lib.vsom_layernorm_bwd_partial(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(resid), ptr(dx), rows, cols,
                                             ptr(part), part.numel(), stream()), "vsom_layernorm_bwd_partial")
        self.n += 1
        return dx

    def flush(self):
        if self.n == self.flushed:
            return
        if self.table is None or self.table.shape[0] < self.n:
            self.table = torch.tensor(self.rows_host, dtype=torch.int64, device=self.device)
        check(lib.vsom_layernorm_bwd_finish_many(ptr(self.table), self.flushed, self.n - self.flushed, self.max_cols, stream()),
              "vsom_layernorm_bwd_finish_many")
        self.flushed = self.n


def layernorm_bwd_deferrable(rows: int, cols: int) -> bool:
    return bool(lib.vsom_layernorm_bwd_deferrable(int(rows), int(cols)))


# ---------------------------------------------------------------- attention
def attention_fwd(qkv, out, lse, B, N, H, hd):
    assert qkv.is_contiguous() and out.is_contiguous() and lse.is_contiguous()
    _f32(qkv, "qkv")
    check(lib.vsom_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, H, hd, stream()), "vsom_attention_fwd")
    return out


def attention_bwd(qkv, out, dout, lse, dqkv, delta, B, N, H, hd):
    assert all(t.is_contiguous() for t in (qkv, out, dout, lse, dqkv, delta))
    check(lib.vsom_attention_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(delta), B, N, H, hd, stream()),
          "vsom_attention_bwd")
    return dqkv


def attention_probs(qkv, lse, probs, B, N, H, hd):
    """probs[B,H,N,N] = softmax(q k^T / sqrt(hd)) from qkv and the saved log-sum-exp (attention maps, vit.py:41-42)."""
    assert qkv.is_contiguous() and lse.is_contiguous() and probs.is_contiguous() and probs.shape == (B, H, N, N)
    _f32(qkv, "qkv"); _f32(probs, "probs")
    check(lib.vsom_attention_probs(ptr(qkv), ptr(lse), ptr(probs), B, N, H, hd, stream()), "vsom_attention_probs")
    return probs


# ---------------------------------------------------------------- SOM
def row_inv_norm(x, out, eps=1e-12):
    rows, cols = x.shape
    _f32(x, "x")
    check(lib.vsom_row_inv_norm(ptr(x), _rows(x), rows, cols, float(eps), ptr(out), stream()), "vsom_row_inv_norm")
    return out


DIST_COSINE, DIST_EUCLIDEAN, DIST_MANHATTAN = 0, 1, 2


def row_sqnorm(x, out):
    rows, cols = x.shape
    _f32(x, "x")
    check(lib.vsom_row_sqnorm(ptr(x), _rows(x), rows, cols, ptr(out), stream()), "vsom_row_sqnorm")
    return out


def bmu_euclid_fwd(x, W, sq_x, sq_w, dist: Optional[torch.Tensor], bmu):
    B, L = x.shape
    K = W.shape[0]
    _f32(x, "x"); _f32(W, "W")
    assert W.is_contiguous() and W.shape[1] == L and bmu.dtype == torch.int64 and (dist is None or dist.is_contiguous())
    nbytes = lib.vsom_bmu_cosine_workspace_bytes(B, K, L)
    ws = scratch(nbytes, x.device)
    check(lib.vsom_bmu_euclid_fwd(ptr(x), _rows(x), ptr(W), ptr(sq_x), ptr(sq_w), ptr(dist), ptr(bmu), B, K, L, ptr(ws),
                                  ws.numel(), stream()), "vsom_bmu_euclid_fwd")
    return dist, bmu


def bmu_manhattan_fwd(x, W, dist: Optional[torch.Tensor], bmu):
    B, L = x.shape
    K = W.shape[0]
    _f32(x, "x"); _f32(W, "W")
    assert W.is_contiguous() and W.shape[1] == L and bmu.dtype == torch.int64 and (dist is None or dist.is_contiguous())
    ws = scratch(lib.vsom_bmu_manhattan_workspace_bytes(B, K, L), x.device)
    check(lib.vsom_bmu_manhattan_fwd(ptr(x), _rows(x), ptr(W), ptr(dist), ptr(bmu), B, K, L, ptr(ws), ws.numel(), stream()),
          "vsom_bmu_manhattan_fwd")
    return dist, bmu


def som_bwd_manhattan(x, W, coef, gW, gX, accumulate_gx=True):
    B, L = x.shape
    K = W.shape[0]
    assert W.is_contiguous() and coef.is_contiguous() and gW.is_contiguous()
    check(lib.vsom_som_bwd_manhattan(ptr(x), _rows(x), ptr(W), ptr(coef), ptr(gW), ptr(gX), _rows(gX), int(accumulate_gx),
                                     B, K, L, stream()), "vsom_som_bwd_manhattan")


def bmu_cosine_x3_fwd(x, W, dist: Optional[torch.Tensor], bmu, inv_nx, inv_nw, reranked: Optional[torch.Tensor] = None):
    """Cosine BMU pass (norms + three-product bf16 contraction + exact re-rank) -> dist, bmu, inv_nx, inv_nw.
    `reranked`: optional int32 device scalar counting the rows whose minimum had to be re-ranked."""
    B, L = x.shape
    K = W.shape[0]
    _f32(x, "x"); _f32(W, "W"); _f32(inv_nx, "inv_nx"); _f32(inv_nw, "inv_nw")
    assert W.is_contiguous() and W.shape[1] == L and bmu.dtype == torch.int64 and (dist is None or dist.is_contiguous())
    assert reranked is None or (reranked.dtype == torch.int32 and reranked.is_cuda)
    nbytes = lib.vsom_bmu_cosine_x3_workspace_bytes(B, K, L)
    ws = scratch(nbytes, x.device)
    rec = _timers.get("bmu_cosine_dots")
    if rec is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(_lib_mod.launch_torch_stream())
    check(lib.vsom_bmu_cosine_x3_dots(ptr(x), _rows(x), ptr(W), B, K, L, ptr(ws), ws.numel(), stream()), "vsom_bmu_cosine_x3_dots")
    if rec is not None:
        e1.record(_lib_mod.launch_torch_stream())
        rec.append((e0, e1))
    check(lib.vsom_bmu_cosine_x3_finalize(ptr(x), _rows(x), ptr(W), ptr(ws), ws.numel(), ptr(dist), ptr(bmu), ptr(inv_nx),
                                          ptr(inv_nw), ptr(reranked), B, K, L, stream()), "vsom_bmu_cosine_x3_finalize")
    return dist, bmu


def bmu_planes_supported(B: int, K: int, L: int) -> bool:
    """Does the pre-split ("planes") form of the cosine BMU pass cover this shape?"""
    return bool(lib.vsom_bmu_cosine_x3_planes_supported(int(B), int(K), int(L)))


def bmu_planes_alloc(R: int, L: int, device) -> torch.Tensor:
    """Plane buffer of an operand [R, L] (fragment image + squared-norm partials)."""
    return torch.empty(lib.vsom_bmu_planes_bytes(int(R), int(L)), dtype=torch.uint8, device=device)


def bmu_planes_from(src, planes):
    """Split the fp32 operand `src` [R, L] (rows 16-byte aligned) into its plane buffer."""
    R, L = src.shape
    _f32(src, "src")
    assert src.stride(1) == 1 and planes.dtype == torch.uint8
    check(lib.vsom_bmu_planes_from(ptr(src), _rows(src), R, L, ptr(planes), planes.numel(), stream()), "vsom_bmu_planes_from")
    return planes


def bmu_cosine_x3_planes_fwd(x, W, xplanes, wplanes, dist: Optional[torch.Tensor], bmu, inv_nx, inv_nw,
                             reranked: Optional[torch.Tensor] = None):
    """bmu_cosine_x3_fwd on pre-split operands: `xplanes` / `wplanes` must describe `x` / `W` as they are now (x and W
    themselves feed the exact re-rank)."""
    B, L = x.shape
    K = W.shape[0]
    _f32(x, "x"); _f32(W, "W"); _f32(inv_nx, "inv_nx"); _f32(inv_nw, "inv_nw")
    assert W.is_contiguous() and W.shape[1] == L and bmu.dtype == torch.int64 and (dist is None or dist.is_contiguous())
    assert reranked is None or (reranked.dtype == torch.int32 and reranked.is_cuda)
    nbytes = lib.vsom_bmu_cosine_x3_planes_workspace_bytes(B, K, L)
    if nbytes == 0:
        raise ValueError(f"bmu_cosine_x3_planes_fwd: shape B={B} K={K} L={L} is not covered by the planes form")
    ws = scratch(nbytes, x.device)
    rec = _timers.get("bmu_cosine_dots")
    if rec is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(_lib_mod.launch_torch_stream())
    check(lib.vsom_bmu_cosine_x3_planes_dots(ptr(xplanes), ptr(wplanes), B, K, L, ptr(ws), ws.numel(), stream()),
          "vsom_bmu_cosine_x3_planes_dots")
    if rec is not None:
        e1.record(_lib_mod.launch_torch_stream())
        rec.append((e0, e1))
    check(lib.vsom_bmu_cosine_x3_planes_finalize(ptr(x), _rows(x), ptr(W), ptr(xplanes), ptr(wplanes), ptr(ws), ws.numel(),
                                                 ptr(dist), ptr(bmu), ptr(inv_nx), ptr(inv_nw), ptr(reranked), B, K, L, stream()),
          "vsom_bmu_cosine_x3_planes_finalize")
    return dist, bmu


def bmu_cosine_fwd(x, W, inv_nx, inv_nw, dist: Optional[torch.Tensor], bmu):
    B, L = x.shape
    K = W.shape[0]
    _f32(x, "x"); _f32(W, "W")
    assert W.is_contiguous() and W.shape[1] == L and bmu.dtype == torch.int64 and (dist is None or dist.is_contiguous())
    nbytes = lib.vsom_bmu_cosine_workspace_bytes(B, K, L)
    ws = scratch(nbytes, x.device)
    rec = _timers.get("bmu_cosine_dots")
    if rec is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(_lib_mod.launch_torch_stream())
    check(lib.vsom_bmu_cosine_dots(ptr(x), _rows(x), ptr(W), B, K, L, ptr(ws), ws.numel(), stream()), "vsom_bmu_cosine_dots")
    if rec is not None:
        e1.record(_lib_mod.launch_torch_stream())
        rec.append((e0, e1))
    check(lib.vsom_bmu_cosine_finalize(ptr(ws), ws.numel(), ptr(inv_nx), ptr(inv_nw), ptr(dist), ptr(bmu), B, K, L, stream()),
          "vsom_bmu_cosine_finalize")
    return dist, bmu


def som_neigh_loss(dist, bmu, grid, T, loss_sum, h=None, inv_nx=None, inv_nw=None, grad_scale=0.0, coef=None,
                   row_dot=None, col_dot=None, distance=DIST_COSINE):
    B, K = dist.shape
    assert dist.is_contiguous() and grid.is_contiguous() and bmu.dtype == torch.int64
    nbytes = lib.vsom_som_neigh_workspace_bytes(B, K)
    ws = scratch(nbytes, dist.device)
    check(lib.vsom_som_neigh_loss(ptr(dist), ptr(bmu), ptr(grid), float(T), ptr(inv_nx), ptr(inv_nw), float(grad_scale),
                                  ptr(h), ptr(loss_sum), ptr(coef), ptr(row_dot), ptr(col_dot), B, K, int(distance), ptr(ws), ws.numel(),
                                  stream()), "vsom_som_neigh_loss")
    return loss_sum


def som_bwd(x, W, coef, row_dot, col_dot, gW, gX, accumulate_gx=True):
    B, L = x.shape
    K = W.shape[0]
    assert W.is_contiguous() and coef.is_contiguous() and gW.is_contiguous()
    check(lib.vsom_som_bwd(ptr(x), _rows(x), ptr(W), ptr(coef), ptr(row_dot), ptr(col_dot), ptr(gW), ptr(gX), _rows(gX),
                           int(accumulate_gx), B, K, L, stream()), "vsom_som_bwd")


# ---------------------------------------------------------------- losses
def l1_unpatchify(pred, img, loss_sum, recon=None, dpred=None, grad_scale=0.0, p=1):
    B, Cc, S, _ = img.shape
    assert pred.is_contiguous() and img.is_contiguous()
    nbytes = lib.vsom_l1_unpatchify_workspace_bytes(B, Cc, S, p)
    ws = scratch(nbytes, img.device)
    check(lib.vsom_l1_unpatchify(ptr(pred), ptr(img), ptr(recon), ptr(loss_sum), ptr(dpred), float(grad_scale), B, Cc, S, p,
                                 ptr(ws), ws.numel(), stream()), "vsom_l1_unpatchify")
    return loss_sum


def cross_entropy_ls(logits, y, smoothing, loss_sum, dlogits=None, grad_scale=0.0):
    B, Cn = logits.shape
    assert logits.is_contiguous() and y.dtype == torch.int64
    nbytes = lib.vsom_cross_entropy_ls_workspace_bytes(B)
    ws = scratch(nbytes, logits.device)
    check(lib.vsom_cross_entropy_ls(ptr(logits), ptr(y), float(smoothing), ptr(loss_sum), ptr(dlogits), float(grad_scale), B,
                                    Cn, ptr(ws), ws.numel(), stream()), "vsom_cross_entropy_ls")
    return loss_sum


# ---------------------------------------------------------------- optimiser / utilities
def adamw_step(p, g, m, v, wd_chunk, lr, beta1, beta2, eps, step, grad_scale=1.0, adamw=True, planes=None):
    """One AdamW / Adam step over the flat arena.  `planes` = (element offset, R, L, plane buffer) of a [R, L] parameter
    whose BMU plane image is rewritten from the updated values in the same pass."""
    n = p.numel()
    if planes is not None:
        off, R, L, buf = planes
        check(lib.vsom_adamw_step_planes(ptr(p), ptr(g), ptr(m), ptr(v), ptr(wd_chunk), n, float(lr), float(beta1), float(beta2),
                                         float(eps), int(step), float(grad_scale), int(adamw), int(off), int(R), int(L), ptr(buf),
                                         buf.numel(), stream()), "vsom_adamw_step_planes")
        return
    check(lib.vsom_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(wd_chunk), n, float(lr), float(beta1), float(beta2),
                              float(eps), int(step), float(grad_scale), int(adamw), stream()), "vsom_adamw_step")


def fill(t, value):
    check(lib.vsom_fill(ptr(t), t.numel(), float(value), stream()), "vsom_fill")
    return t


def set_attention_fused(fused):
    """Test / measurement hook: False (0) runs the short-sequence attention backward as two launches, True (1) is the
    default (one launch, scores shared between its phases where the shape allows), 2 = one launch with recomputed scores."""
    global _attention_fused
    check(lib.vsom_set_attention_fused(int(fused)), "vsom_set_attention_fused")
    _attention_fused = int(fused)


_attention_fused = 1


def get_attention_fused() -> int:
    """The value last set through set_attention_fused (the library's default otherwise)."""
    return _attention_fused


def scaled_mul(out, a, b=None, scale_dev=None, factor=1.0):
    """out = factor * scale_dev[0] * a * b  (b / scale_dev optional)."""
    _f32(out, "out"); _f32(a, "a")
    assert out.is_contiguous() and a.is_contiguous() and out.numel() == a.numel() and (b is None or (b.is_contiguous() and b.numel() == a.numel()))
    check(lib.vsom_scaled_mul(ptr(out), ptr(a), ptr(b), a.numel(), ptr(scale_dev), float(factor), stream()), "vsom_scaled_mul")
    return out


def som_weighted_loss(dist, weights, loss_sum, inv_nx=None, inv_nw=None, grad_scale=0.0, coef=None, row_dot=None, col_dot=None,
                      distance=0):
    """loss_sum <- sum(weights * dist) (+ backward coefficients of grad_scale * that sum when coef etc. are given)."""
    B, K = dist.shape
    _f32(dist, "dist"); _f32(weights, "weights")
    assert dist.is_contiguous() and weights.is_contiguous() and weights.shape == dist.shape
    nbytes = lib.vsom_som_neigh_workspace_bytes(B, K)
    ws = scratch(nbytes, dist.device)
    check(lib.vsom_som_weighted_loss(ptr(dist), ptr(weights), ptr(inv_nx), ptr(inv_nw), float(grad_scale), ptr(loss_sum), ptr(coef),
                                     ptr(row_dot), ptr(col_dot), B, K, int(distance), ptr(ws), ws.numel(), stream()),
          "vsom_som_weighted_loss")
    return loss_sum


def lincomb2(out, a, ca, b, cb, counter=None):
    """out[0] = ca * a[0] + cb * b[0] (device scalars); `counter` (a 0-dim int64 device tensor) += 1 in the same launch."""
    if counter is not None:
        assert counter.dtype == torch.int64 and counter.is_cuda and counter.numel() == 1
    check(lib.vsom_lincomb2(ptr(out), ptr(a), float(ca), ptr(b), float(cb), None if counter is None else counter.data_ptr(), stream()),
          "vsom_lincomb2")
    return out


def loss_parts(parts, main_sum, main_scale, som_sum, som_coef, som_scale, counter=None):
    """parts[0:3] = (total, main term, SOM term) of the step; `counter` += 1 in the same launch (see lincomb2)."""
    if counter is not None:
        assert counter.dtype == torch.int64 and counter.is_cuda and counter.numel() == 1
    _f32(parts, "parts")
    assert parts.numel() >= 3 and parts.is_contiguous()
    check(lib.vsom_loss_parts(ptr(parts), ptr(main_sum), float(main_scale), ptr(som_sum), float(som_coef), float(som_scale),
                              None if counter is None else counter.data_ptr(), stream()), "vsom_loss_parts")
    return parts


def scale_by(t, scale_dev):
    """t *= scale_dev[0] (device scalar, no host sync)."""
    _f32(t, "t"); _f32(scale_dev, "scale")
    assert t.is_contiguous() and scale_dev.numel() == 1
    check(lib.vsom_scale_by(ptr(t), t.numel(), ptr(scale_dev), stream()), "vsom_scale_by")
    return t


def reduce_slabs(slabs, out):
    nslabs, n = slabs.shape
    check(lib.vsom_reduce_slabs(ptr(slabs), slabs.stride(0), nslabs, ptr(out), n, stream()), "vsom_reduce_slabs")
    return out


# ---------------------------------------------------------------- evaluation
def contingency(a, b, table, bad):
    """table[a[i], b[i]] += 1 (int64 device tensors; table is [na, nb] int64, accumulated)."""
    assert a.dtype == torch.int64 and b.dtype == torch.int64 and table.dtype == torch.int64 and bad.dtype == torch.int32
    assert a.is_cuda and a.is_contiguous() and b.is_contiguous() and table.is_contiguous() and a.numel() == b.numel()
    na, nb = table.shape
    check(lib.vsom_contingency(ptr(a), ptr(b), a.numel(), na, nb, ptr(table), ptr(bad), stream()), "vsom_contingency")
    return table


def argmax_rows(x, out):
    rows, cols = x.shape
    _f32(x, "x")
    assert out.dtype == torch.int64
    check(lib.vsom_argmax_rows(ptr(x), _rows(x), rows, cols, ptr(out), stream()), "vsom_argmax_rows")
    return out


# ---------------------------------------------------------------- data-parallel exchange (RCCL)
COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """Rank 0: a fresh RCCL unique id (host bytes) to hand to every rank's comm_init."""
    import ctypes
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    check(lib.vsom_comm_unique_id(buf), "vsom_comm_unique_id")
    return buf.raw


def comm_init(unique_id: bytes, world_size: int, rank: int):
    """Collective: build this process's communicator (the current HIP device is this rank's GPU)."""
    import ctypes
    assert len(unique_id) == COMM_ID_BYTES
    buf = ctypes.create_string_buffer(unique_id, COMM_ID_BYTES)
    check(lib.vsom_comm_init(buf, int(world_size), int(rank)), "vsom_comm_init")


def comm_info():
    import ctypes
    w, r = ctypes.c_int(), ctypes.c_int()
    check(lib.vsom_comm_info(ctypes.byref(w), ctypes.byref(r)), "vsom_comm_info")
    return w.value, r.value


def comm_allreduce_sum(t):
    """In-place sum of `t` over the ranks, enqueued on the launch stream (no host sync)."""
    _f32(t, "t")
    assert t.is_contiguous()
    check(lib.vsom_comm_allreduce_sum(ptr(t), t.numel(), stream()), "vsom_comm_allreduce_sum")
    return t


def comm_destroy():
    check(lib.vsom_comm_destroy(), "vsom_comm_destroy")


# ---------------------------------------------------------------- launch tape
def tape_begin() -> int:
    tid = int(lib.vsom_tape_begin())
    if tid <= 0:
        check(tid, "vsom_tape_begin")
    return tid


def tape_cut() -> int:
    return int(lib.vsom_tape_cut())


def tape_end() -> int:
    return int(lib.vsom_tape_end())


def tape_recording() -> int:
    """0 = not recording, 1 = recording, 2 = recording but paused."""
    return int(lib.vsom_tape_recording())


class tape_hole:
    """A call whose arguments change from step to step: executed but kept OFF the tape, which is cut around it (the host
    re-issues it between two replayed segments).  A no-op when no tape is being recorded."""

    def __enter__(self):
        self.active = tape_recording() == 1
        if self.active:
            lib.vsom_tape_cut()
            check(lib.vsom_tape_pause(1), "vsom_tape_pause")

    def __exit__(self, *exc):
        if self.active:
            check(lib.vsom_tape_pause(0), "vsom_tape_pause")


def tape_replay(tape: int, segment: int):
    check(lib.vsom_tape_replay(int(tape), int(segment)), "vsom_tape_replay")


def tape_segment_ops(tape: int, segment: int) -> int:
    return int(lib.vsom_tape_segment_ops(int(tape), int(segment)))


def tape_destroy(tape: int):
    check(lib.vsom_tape_destroy(int(tape)), "vsom_tape_destroy")
