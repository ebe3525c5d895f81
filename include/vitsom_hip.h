/* libvitsom_hip.so -- C-ABI of the MI355X-native ViT-SOM training-step kernels (gfx950 only).
 *
 * The reference (aluo7/ViT-SOM) is pure Python and has NO FFI/plugin interface;
 * each entry point below replaces the torch/ATen op sequence at the cited reference lines
 * (paths relative to the reference repo root).  A maintainer binds them with ctypes
 * (see INTEGRATION.md); the build's own host mirror lives in vit_som_amd/.
 *
 * Conventions (every entry):
 *   - returns 0 (VSOM_OK) on success, a negative VSOM_E* for a rejected call (bad shape,
 *     alignment, unsupported configuration, short workspace) or a positive hipError_t;
 *     vsom_last_error_string() describes the last failure on the calling thread;
 *   - all pointers are DEVICE pointers owned by the caller, fp32 row-major unless stated;
 *     BMU indices and labels are int64; "ld*" are row strides in ELEMENTS;
 *   - no allocation, no host synchronisation, no exceptions; work is enqueued on `stream`;
 *   - scratch is caller-supplied, sized by the matching *_workspace_bytes() query.
 */
#ifndef VITSOM_HIP_H
#define VITSOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vsom_stream_t; /* == hipStream_t */

#define VSOM_OK 0
#define VSOM_EINVAL (-1)       /* null pointer / non-positive or inconsistent shape */
#define VSOM_EALIGN (-2)       /* pointer or leading dimension not 16-byte aligned where required */
#define VSOM_EUNSUPPORTED (-3) /* configuration outside what the kernels implement */
#define VSOM_EWORKSPACE (-4)   /* workspace pointer null or too small */

#define VSOM_VERSION 100

#define VSOM_DIST_COSINE 0
#define VSOM_DIST_EUCLIDEAN 1
#define VSOM_DIST_MANHATTAN 2   /* torch.cdist(p=1) (models/som_layer.py:115-116; the DESOM configs' distance) */

int vsom_version(void);
const char* vsom_last_error_string(void);

/* ------------------------------------------------------------------ Linear layers
 * (fp32-accurate on the matrix cores: split-bf16 engine by default, exact-f32 MFMA on request -- vsom_set_gemm_mode) */
/* Y[M,N] = X[M,K] * W[N,K]^T + bias      -- nn.Linear forward: qkv models/vit.py:30,
 * decoder_pred vit.py:234, cls_head models/vit_som.py:77.  bias may be NULL. */
int vsom_linear_fwd(const float* X, long ldx, const float* W, const float* bias, float* Y, long ldy,
                    int M, int N, int K, vsom_stream_t stream);

/* pre = X*W^T + bias ; Yact = gelu_erf(pre) ; Ygrad = gelu_erf'(pre)   -- mlp.0 + nn.GELU(),
 * vit.py:53-54.  The derivative is saved instead of the pre-activation (it is all the backward
 * needs, and both come from one exponential).  Both outputs dense [M,N]. */
int vsom_linear_gelu_fwd(const float* X, long ldx, const float* W, const float* bias, float* Ygrad,
                         float* Yact, int M, int N, int K, vsom_stream_t stream);

/* pre = X*W^T + bias ; Yact = max(pre, 0) ; Ygrad = (pre > 0) as 1.0 / 0.0   -- nn.Linear + nn.ReLU of the
 * DESOM autoencoder (models/ae.py:44-59).  Same contract as vsom_linear_gelu_fwd: the backward is
 * vsom_linear_bwd_input*(…, gelu_grad = Ygrad). */
int vsom_linear_relu_fwd(const float* X, long ldx, const float* W, const float* bias, float* Ygrad,
                         float* Yact, int M, int N, int K, vsom_stream_t stream);

/* Y[m] = X[m]*W^T + bias + R[m % r_mod]   -- Linear + residual add (attn.proj vit.py:38,61;
 * mlp.2 vit.py:55,62: r_mod = M, R = block input) and Linear + broadcast table
 * (decoder_embed + decoder_pos_embed vit.py:225-226: r_mod = tokens per image). */
int vsom_linear_residual_fwd(const float* X, long ldx, const float* W, const float* bias,
                             const float* R, long ldr, int r_mod, float* Y, long ldy, int M, int N,
                             int K, vsom_stream_t stream);

/* dX[M,K] (+)= dY[M,N] * W[N,K]  (optionally  .* gelu_grad[M,K], the Ygrad of vsom_linear_gelu_fwd)
 * -- autograd of nn.Linear w.r.t. its input (and of nn.GELU when gelu_grad != NULL, dense [M,K]). */
int vsom_linear_bwd_input(const float* dY, long lddy, const float* W, float* dX, long lddx, int M,
                          int N, int K, int accumulate, const float* gelu_grad, vsom_stream_t stream);

/* Same product from a TRANSPOSED weight copy Wt[K,N] (row-major, ld = N): both operands are then
 * contiguous along the reduction and the GEMM runs on the same kernel family as the forward
 * Linear (vsom_transpose_many keeps the copies current). */
int vsom_linear_bwd_input_t(const float* dY, long lddy, const float* Wt, float* dX, long lddx, int M,
                            int N, int K, int accumulate, const float* gelu_grad, vsom_stream_t stream);

/* Batched 2-D transpose: for i < count, dst_base[dst_off_i + c*rows_i + r] = src_base[src_off_i + r*cols_i + c].
 * `table` is a DEVICE array of count x 4 int64 {src_off, dst_off, rows, cols} (offsets in floats);
 * max_rows / max_cols bound every entry (they size the grid). */
int vsom_transpose_many(const float* src_base, float* dst_base, const long long* table, int count,
                        int max_rows, int max_cols, vsom_stream_t stream);

/* Arithmetic of the nn.Linear-shaped GEMMs (vsom_linear_*_fwd, vsom_linear_bwd_input_t, vsom_linear_bwd_weight,
 * vsom_patch_embed_*).  fp32 operands, fp32 accumulation and fp32 results in every mode:
 *   VSOM_GEMM_F32              v_mfma_f32_32x32x2_f32, bitwise an fmaf chain;
 *   VSOM_GEMM_SPLIT_BF16       every operand split EXACTLY into three bf16 pieces, the six leading cross products on
 *                              v_mfma_f32_32x32x16_bf16 (csrc/gemm_x6.h): dropped terms < 2^-22 relative -- fp32-accurate,
 *                              2.7x fewer matrix-core cycles than the f32 MFMA;
 *   VSOM_GEMM_SPLIT_BF16_GRAD3 (default) forward GEMMs exactly as VSOM_GEMM_SPLIT_BF16 -- outputs, logits, distances and
 *                              BMU indices are bit-identical to that mode -- while the Linear layers' weight-gradient and
 *                              input-gradient GEMMs (vsom_linear_bwd_weight, vsom_linear_bwd_input_t) use the two-piece
 *                              round-to-nearest split, three products: |error| <= 3 * 2^-16 per product in the worst case,
 *                              4e-6 relative on a weight gradient, 1e-5 on the gradients of the whole 12 + 2-layer step
 *                              (tools/accuracy_modes.py; the bar is 1e-4), half the matrix-core work of the backward GEMMs.
 * The cosine BMU pass has its own engines: vsom_bmu_cosine_x3_* (default: two-piece split, three products, exact
 * re-rank) and vsom_bmu_cosine_* (exact-f32 MFMA; what the host mirror uses in VSOM_GEMM_F32 mode and for the
 * euclidean distance).  Process-wide; returns VSOM_EINVAL on an unknown mode. */
#define VSOM_GEMM_F32 0
#define VSOM_GEMM_SPLIT_BF16 1
#define VSOM_GEMM_SPLIT_BF16_GRAD3 2
int vsom_set_gemm_mode(int mode);
int vsom_get_gemm_mode(void);

/* dW[N,K] = dY[M,N]^T * X[M,K] ;  db[N] = column sums of dY (db may be NULL)
 * -- autograd of nn.Linear w.r.t. weight/bias.  The reduction over the M token rows is split
 * across workgroups into fp32 slabs in `ws` and summed in a fixed order (deterministic). */
size_t vsom_linear_bwd_weight_workspace_bytes(int M, int N, int K);
int vsom_linear_bwd_weight(const float* dY, long lddy, const float* X, long ldx, float* dW, float* db,
                           int M, int N, int K, void* ws, size_t ws_bytes, vsom_stream_t stream);

/* ------------------------------------------------------------------ patch embedding */
/* tokens[B, n+1, E]: row 0 = cls_token + pos[0]; row 1+i = Conv2d(k=s=p)(img)[patch i] + bias + pos[1+i]
 * -- timm PatchEmbed + vit.py:207-212.  img [B,C,S,S]; Wpe [E, C*p*p] (= conv weight [E,C,p,p]);
 * pos [n+1, E]; xp_ws [B*n, C*p*p] scratch that the backward re-uses (gathered patches). */
int vsom_patch_embed_fwd(const float* img, const float* Wpe, const float* bpe, const float* pos,
                         const float* cls_token, float* tokens, float* xp_ws, int B, int C, int S,
                         int p, int E, vsom_stream_t stream);
size_t vsom_patch_embed_bwd_workspace_bytes(int B, int C, int S, int p, int E);
/* dWpe, dbpe, dcls_token[E] from dtokens[B, n+1, E] (autograd of the above; pos is frozen). */
int vsom_patch_embed_bwd(const float* dtokens, const float* xp_ws, float* dWpe, float* dbpe,
                         float* dcls_token, int B, int C, int S, int p, int E, void* ws,
                         size_t ws_bytes, vsom_stream_t stream);

/* ------------------------------------------------------------------ LayerNorm */
/* Y = (X - mean)/sqrt(var + eps) * gamma + beta, rows x cols; saves mean/rstd per row.
 * -- nn.LayerNorm(eps=1e-6): vit.py:48,50,85,95 (eps from vit_som.py:50). cols <= 1024. */
int vsom_layernorm_fwd(const float* X, const float* gamma, const float* beta, float* Y, float* mean,
                       float* rstd, int rows, int cols, float eps, vsom_stream_t stream);
size_t vsom_layernorm_bwd_workspace_bytes(int rows, int cols);
/* dX = (resid ? resid : 0) + LN'(dY);  dgamma, dbeta = column reductions (deterministic). */
int vsom_layernorm_bwd(const float* dY, const float* X, const float* mean, const float* rstd,
                       const float* gamma, const float* resid, float* dX, float* dgamma, float* dbeta,
                       int rows, int cols, void* ws, size_t ws_bytes, vsom_stream_t stream);
/* The same in two halves (autograd of nn.LayerNorm, vit.py:48,50,85,95): _partial computes dX and leaves the column
 * partials in `part` (vsom_layernorm_bwd_workspace_bytes, kept by the caller); one vsom_layernorm_bwd_finish_many launch
 * then produces dgamma / dbeta for `count` such calls, bit for bit what vsom_layernorm_bwd would have written (the step's
 * 30 tiny reductions leave its critical chain).  jobs_dev: device array of VSOM_LN_JOB_WORDS 64-bit words per job --
 * part, dgamma, dbeta (pointers), (nblk << 32) | cols with nblk = workspace_bytes / (8 cols); jobs first .. first+count-1
 * are reduced, max_cols = the widest of them.  Shapes: vsom_layernorm_bwd_deferrable(rows, cols) != 0. */
#define VSOM_LN_JOB_WORDS 4
int vsom_layernorm_bwd_deferrable(int rows, int cols);
int vsom_layernorm_bwd_partial(const float* dY, const float* X, const float* mean, const float* rstd,
                               const float* gamma, const float* resid, float* dX, int rows, int cols, void* part,
                               size_t part_bytes, vsom_stream_t stream);
int vsom_layernorm_bwd_finish_many(const int64_t* jobs_dev, int first, int count, int max_cols, vsom_stream_t stream);

/* ------------------------------------------------------------------ multi-head attention */
/* out[B,N,H*hd] = softmax(q k^T * hd^-0.5) v, qkv laid out [B,N,3,H,hd] (the qkv Linear's
 * output, vit.py:30-37; dropout p=0).  lse[B,H,N] = log-sum-exp of the scaled scores (saved for
 * the backward, which recomputes the probabilities). */
int vsom_attention_fwd(const float* qkv, float* out, float* lse, int B, int N, int H, int hd,
                       vsom_stream_t stream);
/* dqkv[B,N,3,H,hd] from dout[B,N,H*hd] (autograd of the above). delta_ws: [B,H,N] floats. */
int vsom_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                       float* dqkv, float* delta_ws, int B, int N, int H, int hd,
                       vsom_stream_t stream);
/* probs[B,H,N,N] = softmax(q k^T * hd^-0.5) -- the attention maps of `return_attn=True` (vit.py:33-34,41-42), formed
 * from qkv and the lse the forward saved (the fused kernels never materialise them).  Visualisation path only. */
int vsom_attention_probs(const float* qkv, const float* lse, float* probs, int B, int N, int H, int hd,
                         vsom_stream_t stream);
/* test hook: 0 = run the short-sequence backward as two launches (dQ, then dK/dV), 1 = one launch whose phases share
   the scores (default), 2 = one launch that recomputes them: bit-identical results (the GPU suite checks) -- except that
   form 1, at hd = 64 in VSOM_GEMM_SPLIT_BF16_GRAD3 mode, forms its seven products on the bf16 matrix cores from the
   two-piece split of the gradient GEMMs (gradients within 8e-6 relative of fp64); 3 = form 1 with fp32 products always */
int vsom_set_attention_fused(int fused);

/* ------------------------------------------------------------------ SOM layer */
/* inv_norm[r] = 1 / max(||X[r,:]||_2, eps)   -- F.normalize(p=2, eps=1e-12), som_layer.py:120-121 */
int vsom_row_inv_norm(const float* X, long ldx, int rows, int cols, float eps, float* inv_norm,
                      vsom_stream_t stream);
/* sqnorm[r] = ||X[r,:]||_2^2  (for the euclidean distance) */
int vsom_row_sqnorm(const float* X, long ldx, int rows, int cols, float* sqnorm, vsom_stream_t stream);
/* Best-matching-unit search, cosine: dist[B,K] = 1 - (X/|X|)(W/|W|)^T ; bmu[b] = first argmin_k
 * -- SOMLayer.compute_distances + forward, som_layer.py:83-89,119-122.  X [B,L] (row stride ldx:
 * the patch tokens of image b are contiguous), W [K,L] dense.  dist may be NULL. */
size_t vsom_bmu_cosine_workspace_bytes(int B, int K, int L);
int vsom_bmu_cosine_fwd(const float* X, long ldx, const float* W, const float* inv_nx,
                        const float* inv_nw, float* dist, int64_t* bmu, int B, int K, int L, void* ws,
                        size_t ws_bytes, vsom_stream_t stream);
/* The two halves of vsom_bmu_cosine_fwd, callable separately (bench.py times the distance pass
 * alone): _dots = the [B,L]x[L,K] contraction, split over L into fp32 partial slabs in ws;
 * _finalize = slab sum, 1 - dot * inv_nx * inv_nw, first-argmin. */
int vsom_bmu_cosine_dots(const float* X, long ldx, const float* W, int B, int K, int L, void* ws,
                         size_t ws_bytes, vsom_stream_t stream);
int vsom_bmu_cosine_finalize(const void* ws, size_t ws_bytes, const float* inv_nx, const float* inv_nw,
                             float* dist, int64_t* bmu, int B, int K, int L, vsom_stream_t stream);
/* Best-matching-unit search, euclidean: dist = torch.cdist(X, W, p=2) in its matmul form
 * sqrt(clamp_min(|x|^2 + |w|^2 - 2 x.w, 1e-30)) -- som_layer.py:117-118; same workspace as cosine. */
int vsom_bmu_euclid_fwd(const float* X, long ldx, const float* W, const float* sq_x, const float* sq_w,
                        float* dist, int64_t* bmu, int B, int K, int L, void* ws, size_t ws_bytes,
                        vsom_stream_t stream);
/* Neighbourhood weights + SOM loss + backward coefficients in one pass over [B,K]:
 *   h_ik = exp(-||g_k - g_bmu(i)||^2 / (2 T^2))              compute_weights, som_layer.py:144-152
 *   loss_sum[0] = sum_ik h_ik d_ik   (caller divides by B*K)  som_loss, som_layer.py:137-142
 *   (all loss sums are two-stage fixed-order reductions: bitwise reproducible)
 *   coef[i,k] = -c * h_ik * inv_nx[i] * inv_nw[k],  c = grad_scale
 *   row_dot[i] = c * inv_nx[i]^2 * sum_k h_ik (1 - d_ik);  col_dot[k] likewise over i
 * (distance = VSOM_DIST_EUCLIDEAN: coef = -c h/d, row_dot = c sum_k h/d, col_dot = c sum_i h/d, the
 *  coefficients of d|x-w|; inv_nx / inv_nw are then unused and vsom_som_bwd applies unchanged)
 * h may be NULL; coef/row_dot/col_dot may all be NULL (forward only).  grid [K,2] float. */
size_t vsom_som_neigh_workspace_bytes(int B, int K);
int vsom_som_neigh_loss(const float* dist, const int64_t* bmu, const float* grid, float T,
                        const float* inv_nx, const float* inv_nw, float grad_scale, float* h,
                        float* loss_sum, float* coef, float* row_dot, float* col_dot, int B, int K,
                        int distance, void* ws, size_t ws_bytes, vsom_stream_t stream);
/* Prototype gradient ("per-BMU neighbourhood accumulator") and input gradient of
 * gamma_t * mean(h * d) through both F.normalize calls:
 *   gW[K,L]  = coef^T X + col_dot[k] * W[k,:]
 *   gX[B,L] += coef  W + row_dot[i] * X[i,:]        (accumulated into the token gradient) */
int vsom_som_bwd(const float* X, long ldx, const float* W, const float* coef, const float* row_dot,
                 const float* col_dot, float* gW, float* gX, long ldgx, int accumulate_gx, int B, int K,
                 int L, vsom_stream_t stream);

/* manhattan BMU search: dist[B,K] = sum_l |X[i,l] - W[k,l]| (torch.cdist p=1, som_layer.py:115-116),
 * bmu = first argmin.  Tiled VALU kernels (|x - w| has no dot-product form); the reduction over l is
 * split into fp32 slabs in `ws`, summed in a fixed order.  dist may be NULL. */
size_t vsom_bmu_manhattan_workspace_bytes(int B, int K, int L);
int vsom_bmu_manhattan_fwd(const float* X, long ldx, const float* W, float* dist, int64_t* bmu, int B, int K,
                           int L, void* ws, size_t ws_bytes, vsom_stream_t stream);

/* Backward of som_loss through the manhattan distances, from coef = dLoss/d dist written by
 * vsom_som_neigh_loss(distance = VSOM_DIST_MANHATTAN):
 *   gW[k,l]   = -sum_i coef[i,k] sign(X[i,l] - W[k,l])
 *   gX[i,l] (+)= sum_k coef[i,k] sign(X[i,l] - W[k,l])          (sign(0) = 0, as torch) */
int vsom_som_bwd_manhattan(const float* X, long ldx, const float* W, const float* coef, float* gW, float* gX,
                           long ldgx, int accumulate_gx, int B, int K, int L, vsom_stream_t stream);

/* ------------------------------------------------------------------ losses */
/* loss_sum[0] = sum_i |pred[i] - target[i]| over n elements (nn.L1Loss numerator, models/desom.py:44,146);
 * dpred[i] = grad_scale * sign(pred[i] - target[i]) when dpred != NULL.  Fixed-order reduction through `ws`. */
size_t vsom_l1_loss_workspace_bytes(long n);
int vsom_l1_loss(const float* pred, const float* target, float* loss_sum, float* dpred, float grad_scale, long n,
                 void* ws, size_t ws_bytes, vsom_stream_t stream);

/* L1Loss(unpatchify(pred[:,1:,:]), img) -- vit.py:141-153,234-236 + vit_som.py:100.
 * pred [B, n+1, p*p*C] (row 0 of each image = CLS prediction, ignored); img [B,C,S,S].
 * recon [B,C,S,S] (may be NULL); loss_sum[0] = sum |recon-img| (caller divides by B*C*S*S);
 * dpred (may be NULL) = grad_scale * sign(recon-img) in pred layout, CLS rows zero. */
size_t vsom_l1_unpatchify_workspace_bytes(int B, int C, int S, int p);
int vsom_l1_unpatchify(const float* pred, const float* img, float* recon, float* loss_sum,
                       float* dpred, float grad_scale, int B, int C, int S, int p, void* ws,
                       size_t ws_bytes, vsom_stream_t stream);
/* CrossEntropyLoss(label_smoothing=s)(logits[B,C], y[B]) -- vit_som.py:62,96.
 * loss_sum[0] = sum_b loss_b (caller divides by B); dlogits (may be NULL) = grad_scale * dloss_b/dlogits. */
size_t vsom_cross_entropy_ls_workspace_bytes(int B);
int vsom_cross_entropy_ls(const float* logits, const int64_t* y, float smoothing, float* loss_sum,
                          float* dlogits, float grad_scale, int B, int C, void* ws, size_t ws_bytes,
                          vsom_stream_t stream);

/* ------------------------------------------------------------------ optimiser */
/* One decoupled-weight-decay Adam step over a flat fp32 arena (torch.optim.AdamW semantics,
 * vit_som.py:146-151).  wd_per_chunk[i] is the weight decay of elements [256 i, 256 i + 256);
 * adamw=0 selects torch.optim.Adam (L2 in the gradient).  g is multiplied by grad_scale first
 * (1/world_size after a sum all-reduce). */
int vsom_adamw_step(float* p, const float* g, float* m, float* v, const float* wd_per_chunk, long n,
                    float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                    int adamw, vsom_stream_t stream);

/* ------------------------------------------------------------------ data-parallel exchange (RCCL over xGMI) */
/* One process per GPU, one communicator per process.  Replaces the implicit DDP gradient all-reduce of the reference
 * (experiments/benchmarking/train_vit_som.py:44-45,86-91: DEVICES > 1 -> Lightning DDPStrategy -> NCCL).  RCCL is bound
 * at run time (dlopen; a copy already in the process, e.g. torch.distributed's, is shared; VSOM_RCCL_PATH overrides).
 *   vsom_comm_unique_id   rank 0 fills `id_out` (HOST memory, VSOM_COMM_ID_BYTES); the host distributes it to all ranks
 *   vsom_comm_init        collective: every rank calls it with the same id (the current HIP device is the rank's GPU)
 *   vsom_comm_allreduce_sum  in-place sum of buf[0..n) over the ranks, enqueued on `stream` (no host sync): the host
 *                         mirror calls it per arena slice as the backward finishes it (ViT gradients and the [K,L]
 *                         prototype accumulators alike); AdamW then applies grad_scale = 1/world_size
 *   vsom_comm_info        world_size / rank of the live communicator (0 / -1 when none)
 *   vsom_comm_destroy     releases it (idempotent) */
#define VSOM_COMM_ID_BYTES 128
int vsom_comm_unique_id(void* id_out);
int vsom_comm_init(const void* unique_id, int world_size, int rank);
int vsom_comm_allreduce_sum(float* buf, long n, vsom_stream_t stream);
int vsom_comm_info(int* world_size, int* rank);
int vsom_comm_destroy(void);

/* ------------------------------------------------------------------ launch tape (the step as ONE host call per segment) */
/* The reference drives its step from Python through ATen, one host round trip per op (vit_som.py:80-105 under Lightning's fit
 * loop); so does the host mirror, ~420 launches and ~4 ms of host time per step -- more than the GPU needs below ~256 images
 * per GPU (BASELINE c4 at 128 per GPU, c3 read as 512 global).  A tape records the launches of one step WHILE THEY RUN and
 * re-issues them from C:
 *   vsom_tape_begin      start recording on the calling thread; returns the tape id (> 0).  Every launch any vsom_* entry
 *                        makes on this thread from now on is executed AND kept (kernel, grid, LDS, stream, by-value arguments),
 *                        as are vsom_event_record / vsom_stream_wait_event and vsom_comm_allreduce_sum
 *   vsom_tape_cut        close the current segment, open the next; returns the index of the closed one
 *   vsom_tape_pause      1: execute but do not keep what follows (calls whose arguments change per step), 0: resume
 *   vsom_tape_end        stop recording; returns the number of segments
 *   vsom_tape_replay     re-issue one segment's operations in order (no per-launch host work beyond hipLaunchKernel)
 *   vsom_tape_segment_ops / vsom_tape_recording (0 no, 1 recording, 2 paused) / vsom_tape_destroy
 * Pointers are frozen into the tape: the caller replays only while every buffer the recorded step touched is alive and in
 * place (the host mirror stages each batch into fixed input buffers and ties the tape to its activation buffers).
 *   vsom_event_record / vsom_stream_wait_event: library-owned events (ids 0..511, created on first use) for the edges
 *   between the step's HIP streams, so that they are part of a tape. */
int vsom_tape_begin(void);
int vsom_tape_cut(void);
int vsom_tape_pause(int paused);
int vsom_tape_end(void);
int vsom_tape_recording(void);
int vsom_tape_segment_ops(int tape, int segment);
int vsom_tape_replay(int tape, int segment);
int vsom_tape_destroy(int tape);
int vsom_event_record(int ev, vsom_stream_t stream);
int vsom_stream_wait_event(vsom_stream_t stream, int ev);

/* ------------------------------------------------------------------ evaluation (tools/evaluation.py) */
/* table[a[i] * nb + b[i]] += 1 for i < n  (uint64 counts, accumulates: zero it before the first
 * batch) -- the contingency matrix of calculate_purity (evaluation.py:142-145) and of the
 * classification metrics.  Entries outside [0,na) x [0,nb) are counted in out_of_range[0]. */
int vsom_contingency(const int64_t* a, const int64_t* b, long n, int na, int nb,
                     unsigned long long* table, int* out_of_range, vsom_stream_t stream);
/* out[r] = first argmax_c X[r,c] -- torch.argmax(cls_logits, dim=1), evaluation.py:119 */
int vsom_argmax_rows(const float* X, long ldx, int rows, int cols, int64_t* out, vsom_stream_t stream);

/* SOMLayer.som_loss(weights, distances) = mean(weights * distances) for ARBITRARY weights (som_layer.py:137-142):
   loss_sum <- sum_ik weights[i,k] dist[i,k]; with coef/row_dot/col_dot given, also the backward coefficients of
   grad_scale * that sum w.r.t. the distances' inputs (what vsom_som_bwd consumes) -- with weights = an upstream
   gradient dL/d dist this is the autograd of SOMLayer.forward itself.  Workspace: vsom_som_neigh_workspace_bytes. */
int vsom_som_weighted_loss(const float* dist, const float* weights, const float* inv_nx, const float* inv_nw,
                           float grad_scale, float* loss_sum, float* coef, float* row_dot, float* col_dot, int B, int K,
                           int distance, void* ws, size_t ws_bytes, vsom_stream_t stream);

/* Cosine BMU pass as a reduced-precision contraction + exact re-rank; replaces
   F.normalize x 2 + matmul + argmin of som_layer.py:119-122, 83-89 in one pass over X and W:
   stage 1 = X W^T on the bf16 matrix cores from a two-piece round-to-nearest split (three products; |error of
   the normalised dot| <= 3 * 2^-16) + the squared row norms of X and W; stage 2 = inv_nx / inv_nw
   (1 / max(|row|, 1e-12)), dist, and bmu = first argmin after every prototype within 2e-4 of the approximate
   minimum has been re-ranked with an exact (fp64-accumulated) dot product, whose distance also replaces the
   approximate one in dist: bmu == argmin(dist) exactly.  K <= 2048; rows 16-byte aligned. */
size_t vsom_bmu_cosine_x3_workspace_bytes(int B, int K, int L);
int vsom_bmu_cosine_x3_dots(const float* X, long ldx, const float* W, int B, int K, int L, void* ws, size_t ws_bytes,
                            vsom_stream_t stream);
int vsom_bmu_cosine_x3_finalize(const float* X, long ldx, const float* W, const void* ws, size_t ws_bytes, float* dist,
                                int64_t* bmu, float* inv_nx, float* inv_nw, int* reranked, int B, int K, int L,
                                vsom_stream_t stream);

/* The same pass on PRE-SPLIT operands ("plane images"): the two-piece bf16 split is taken out of the contraction and done
   once per operand -- for the prototypes by the optimizer step that rewrites them anyway (vsom_adamw_step_planes), for the
   samples by vsom_bmu_planes_from -- so that the contraction is LDS-DMA -> ds_read -> MFMA with no VALU in its loop.
   A plane buffer (vsom_bmu_planes_bytes(R, L) bytes, 16-byte aligned) holds, for an operand [R, L]: the fragment image
   (per 16-deep k step and 32-row block the two planes as 1 KB MFMA fragments, rows >= R and k >= L zero) followed by the
   rows' squared-norm partials.  The caller owns validity: a plane buffer describes the operand as it was when written.
   Covered shapes: vsom_bmu_cosine_x3_planes_supported (B >= 192, L % 8 == 0, K <= 2048, images < 2 GB); slabs, and
   therefore dist / bmu, are those of vsom_bmu_cosine_x3_dots bit for bit (norms: last-bit differences).
   Same reference lines as above (som_layer.py:119-122, 83-89); the AdamW variant replaces vit_som.py:146-151. */
size_t vsom_bmu_planes_bytes(int R, int L);
int vsom_bmu_planes_from(const float* src, long ld, int R, int L, void* planes, size_t planes_bytes, vsom_stream_t stream);
/* vsom_adamw_step over the arena of n elements (same arguments, bitwise the same update) that also writes the plane
   buffer of the [R, L] parameter at element offset slice_off (a multiple of 256) from its UPDATED values. */
int vsom_adamw_step_planes(float* p, const float* g, float* m, float* v, const float* wd_per_chunk, long n, float lr,
                           float beta1, float beta2, float eps, int step, float grad_scale, int adamw, long slice_off,
                           int R, int L, void* planes, size_t planes_bytes, vsom_stream_t stream);
int vsom_bmu_cosine_x3_planes_supported(int B, int K, int L);
size_t vsom_bmu_cosine_x3_planes_workspace_bytes(int B, int K, int L);
int vsom_bmu_cosine_x3_planes_dots(const void* xplanes, const void* wplanes, int B, int K, int L, void* ws, size_t ws_bytes,
                                   vsom_stream_t stream);
int vsom_bmu_cosine_x3_planes_finalize(const float* X, long ldx, const float* W, const void* xplanes, const void* wplanes,
                                       const void* ws, size_t ws_bytes, float* dist, int64_t* bmu, float* inv_nx, float* inv_nw,
                                       int* reranked, int B, int K, int L, vsom_stream_t stream);

/* ------------------------------------------------------------------ small utilities */
int vsom_fill(float* p, long n, float value, vsom_stream_t stream);
/* out[0] = ca * a[0] + cb * b[0]: the step's total loss from its two device-side sums (vit_som.py:93,98); `counter`
   (nullable, one int64 on the device) is incremented by 1 in the same launch: `self.iteration += 1` (vit_som.py:104) */
int vsom_lincomb2(float* out, const float* a, float ca, const float* b, float cb, int64_t* counter, vsom_stream_t stream);
/* The step's loss terms in one launch (vit_som.py:93-102): parts[0] = total = main_scale * main_sum[0] + som_coef * som_sum[0]
   (som_coef = gamma_t / (B K): bitwise vsom_lincomb2's result), parts[1] = main_scale * main_sum[0] (reconstruction or
   cross-entropy term), parts[2] = som_scale * som_sum[0] (the SOM term before gamma); `counter` as in vsom_lincomb2. */
int vsom_loss_parts(float* parts, const float* main_sum, float main_scale, const float* som_sum, float som_coef,
                    float som_scale, int64_t* counter, vsom_stream_t stream);
/* out[i] = factor * (*scale_dev) * a[i] * b[i]  (b, scale_dev nullable -> 1): the elementwise products autograd needs
   for mean(weights * distances) (som_layer.py:137-142) with the upstream gradient as a device scalar */
int vsom_scaled_mul(float* out, const float* a, const float* b, long n, const float* scale_dev, float factor,
                    vsom_stream_t stream);
/* p[i] *= *scale_dev (a device scalar: no host sync) -- the incoming gradient of loss.backward(), applied to the
   loss-side gradient seeds before the backward kernels run (torch autograd's role at vit_som.py:80-105) */
int vsom_scale_by(float* p, long n, const float* scale_dev, vsom_stream_t stream);
/* out[j] = sum_s slabs[s*stride + j], j in [0,n) -- fixed summation order */
int vsom_reduce_slabs(const float* slabs, long stride, int nslabs, float* out, long n,
                      vsom_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VITSOM_HIP_H */
