// Patch embedding, losses, fused AdamW, small utilities.
#include "adamw.h"

namespace vsom {

// ------------------------------------------------------------------ patch gather
// xp[(b*n + pi), c*p*p + py*p + px] = img[b, c, (pi / g)*p + py, (pi % g)*p + px]
// (column order = flattened Conv2d weight [E, C, p, p])
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ img,
                                                           float* __restrict__ xp, long total, int C, int S,
                                                           int p, int g) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    // idx enumerates image pixels (coalesced reads); writes are p-float runs
    const int x = idx % S;
    const int y = (idx / S) % S;
    const int c = (idx / ((long)S * S)) % C;
    const long b = idx / ((long)S * S * C);
    const int pi = (y / p) * g + (x / p);
    const int col = c * p * p + (y % p) * p + (x % p);
    xp[(b * g * g + pi) * (long)(C * p * p) + col] = img[idx];
}

// tokens[b, 0, :] = cls_token + pos[0]
__global__ __launch_bounds__(256) void cls_rows_kernel(const float* __restrict__ cls_token,
                                                       const float* __restrict__ pos, float* __restrict__ tokens,
                                                       int B, int Ntok, int E) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * E) return;
    const int e = idx % E;
    const long b = idx / E;
    tokens[b * Ntok * E + e] = cls_token[e] + pos[e];
}

// dcls[e] = sum_b dtokens[b, 0, e]   (fixed order).  A workgroup owns 32 columns; its 8 row groups
// each sum every 8th image with 4 independent accumulators and are combined through LDS in order.
__global__ __launch_bounds__(256) void cls_grad_kernel(const float* __restrict__ dtokens, float* __restrict__ dcls,
                                                       int B, int Ntok, int E) {
    __shared__ float part[8][32];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < E) {
        const long stride = (long)Ntok * E;
        int b = rg;
        for (; b + 24 < B; b += 32) {
            s0 += dtokens[(long)b * stride + e];
            s1 += dtokens[(long)(b + 8) * stride + e];
            s2 += dtokens[(long)(b + 16) * stride + e];
            s3 += dtokens[(long)(b + 24) * stride + e];
        }
        for (; b < B; b += 8) s0 += dtokens[(long)b * stride + e];
    }
    part[rg][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && e < E) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += part[i][c];
        dcls[e] = s;
    }
}

// ------------------------------------------------------------------ L1 + unpatchify
// pred[b, 1+pi, (py*p + px)*C + c]  <->  pixel (b, c, (pi/g)*p + py, (pi%g)*p + px)
__global__ __launch_bounds__(256) void l1_unpatchify_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ img,
                                                            float* __restrict__ recon, float* __restrict__ part,
                                                            float* __restrict__ dpred, float gscale, long total,
                                                            int C, int S, int p, int g) {
    const int pd = p * p * C, Ntok = g * g + 1;
    float acc = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        // idx enumerates pred elements [B, Ntok, pd]
        const int j = idx % pd;
        const int tok = (idx / pd) % Ntok;
        const long b = idx / ((long)pd * Ntok);
        if (tok == 0) {
            if (dpred) dpred[idx] = 0.f;
            continue;
        }
        const int pi = tok - 1;
        const int c = j % C, px = (j / C) % p, py = j / (C * p);
        const long pix = ((b * C + c) * S + (pi / g) * p + py) * S + (pi % g) * p + px;
        const float v = pred[idx];
        const float d = v - img[pix];
        if (recon) recon[pix] = v;
        acc += fabsf(d);
        if (dpred) dpred[idx] = (d > 0.f) ? gscale : ((d < 0.f) ? -gscale : 0.f);
    }
    acc = wave_sum(acc);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// plain L1 over n elements (+ sign gradient)
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                      float* __restrict__ part, float* __restrict__ dpred, float gscale,
                                                      long n) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        acc += fabsf(d);
        if (dpred) dpred[i] = (d > 0.f) ? gscale : ((d < 0.f) ? -gscale : 0.f);
    }
    acc = wave_sum(acc);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

static int l1_blocks(long total) {
    long b = (total + 1023) / 1024;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------ cross entropy with label smoothing
// wave per sample row
__global__ __launch_bounds__(256) void ce_ls_kernel(const float* __restrict__ logits, const int64_t* __restrict__ y,
                                                    float smoothing, float* __restrict__ part,
                                                    float* __restrict__ dlogits, float gscale, int B, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* z = logits + (long)row * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, z[c]);
    mx = wave_max(mx);
    float se = 0.f, sz = 0.f;
    for (int c = lane; c < C; c += 64) { se += expf(z[c] - mx); sz += z[c]; }
    se = wave_sum(se);
    sz = wave_sum(sz);
    const float lse = mx + logf(se);
    const int64_t t = y[row];
    // loss = (1-s) * (lse - z[t]) + s * (lse - mean(z))
    if (lane == 0) part[row] = (1.f - smoothing) * (lse - z[t]) + smoothing * (lse - sz / (float)C);
    if (dlogits) {
        float* dz = dlogits + (long)row * C;
        const float u = smoothing / (float)C;
        for (int c = lane; c < C; c += 64) {
            const float sm = expf(z[c] - lse);
            dz[c] = gscale * (sm - u - ((c == t) ? (1.f - smoothing) : 0.f));
        }
    }
}

// ------------------------------------------------------------------ AdamW over a flat arena
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const float* __restrict__ wd_chunk, long n4, const AdamwC c) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float wd = wd_chunk[i >> 6];              // 64 float4 = 256 elements per chunk
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
        adamw_update(pp, gg, mm, vv, wd, c);
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, long n, float value) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = value;
}

__global__ __launch_bounds__(256) void scale_by_kernel(float* __restrict__ p, long n, const float* __restrict__ s) {
    const float v = *s;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] *= v;
}

// out[i] = factor * (*scale_dev) * a[i] * b[i]      (b nullable -> 1; scale_dev nullable -> 1)
__global__ __launch_bounds__(256) void scaled_mul_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                         const float* __restrict__ b, long n, const float* __restrict__ s,
                                                         float factor) {
    const float v = (s ? *s : 1.0f) * factor;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = v * a[i] * (b ? b[i] : 1.0f);
}

// out[0] = ca * a[0] + cb * b[0]   (total loss from the two loss sums; one lane) and, with it, the step counter
__global__ void lincomb2_kernel(float* __restrict__ out, const float* __restrict__ a, float ca, const float* __restrict__ b, float cb,
                                long long* __restrict__ counter) {
    if (threadIdx.x == 0) {
        out[0] = fmaf(cb, b[0], ca * a[0]);
        if (counter) counter[0] += 1;
    }
}

// parts[0] = total = main_scale * main_sum + som_coef * som_sum ; parts[1] = main ; parts[2] = som_scale * som_sum
__global__ void loss_parts_kernel(float* __restrict__ parts, const float* __restrict__ main_sum, float main_scale,
                                  const float* __restrict__ som_sum, float som_coef, float som_scale, long long* __restrict__ counter) {
    if (threadIdx.x == 0) {
        const float m = main_scale * main_sum[0];
        parts[0] = fmaf(som_coef, som_sum[0], m);
        parts[1] = m;
        parts[2] = som_scale * som_sum[0];
        if (counter) counter[0] += 1;
    }
}

// batched 32x32-tile transpose through LDS; blockIdx.y = tensor, blockIdx.x = tile (surplus tiles exit)
__global__ __launch_bounds__(256) void transpose_many_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                             const long long* __restrict__ table) {
    __shared__ float tile[32][33];
    const long long* e = table + 4 * (long)blockIdx.y;
    const long so = e[0], dof = e[1];
    const int rows = (int)e[2], cols = (int)e[3];
    const int tc = (cols + 31) >> 5, tr = (rows + 31) >> 5;
    if ((int)blockIdx.x >= tc * tr) return;
    const int r0 = ((int)blockIdx.x / tc) << 5, c0 = ((int)blockIdx.x % tc) << 5;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = src_base + so;
    float* dst = dst_base + dof;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                    // all four loads in flight before the first LDS store
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        v[i] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < rows && c < cols) dst[(long)c * rows + r] = tile[tx][ty + 8 * i];
    }
}

static int grid_for(long n, int per_block, int cap) {
    long b = (n + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace vsom

using namespace vsom;

extern "C" {

int vsom_fill(float* p, long n, float value, vsom_stream_t stream) {
    VSOM_REQUIRE(p && n >= 0, VSOM_EINVAL, "fill: bad arguments");
    if (n == 0) return VSOM_OK;
    VSOM_LAUNCH(fill_kernel, dim3(grid_for(n, 1024, 4096)), dim3(256), 0, stream, p, n, value);
    VSOM_LAUNCH_CHECK("fill_kernel");
}

int vsom_scale_by(float* p, long n, const float* scale_dev, vsom_stream_t stream) {
    VSOM_REQUIRE(p && scale_dev && n >= 0, VSOM_EINVAL, "scale_by: bad arguments");
    if (n == 0) return VSOM_OK;
    VSOM_LAUNCH(scale_by_kernel, dim3(grid_for(n, 1024, 4096)), dim3(256), 0, stream, p, n, scale_dev);
    VSOM_LAUNCH_CHECK("scale_by_kernel");
}

int vsom_scaled_mul(float* out, const float* a, const float* b, long n, const float* scale_dev, float factor,
                    vsom_stream_t stream) {
    VSOM_REQUIRE(out && a && n >= 0, VSOM_EINVAL, "scaled_mul: bad arguments");
    if (n == 0) return VSOM_OK;
    VSOM_LAUNCH(scaled_mul_kernel, dim3(grid_for(n, 1024, 4096)), dim3(256), 0, stream, out, a, b, n, scale_dev, factor);
    VSOM_LAUNCH_CHECK("scaled_mul_kernel");
}

int vsom_lincomb2(float* out, const float* a, float ca, const float* b, float cb, int64_t* counter, vsom_stream_t stream) {
    VSOM_REQUIRE(out && a && b, VSOM_EINVAL, "lincomb2: null pointer");
    VSOM_LAUNCH(lincomb2_kernel, dim3(1), dim3(64), 0, stream, out, a, ca, b, cb, reinterpret_cast<long long*>(counter));
    VSOM_LAUNCH_CHECK("lincomb2_kernel");
}

int vsom_loss_parts(float* parts, const float* main_sum, float main_scale, const float* som_sum, float som_coef, float som_scale,
                    int64_t* counter, vsom_stream_t stream) {
    VSOM_REQUIRE(parts && main_sum && som_sum, VSOM_EINVAL, "loss_parts: null pointer");
    VSOM_LAUNCH(loss_parts_kernel, dim3(1), dim3(64), 0, stream, parts, main_sum, main_scale, som_sum, som_coef, som_scale,
                       reinterpret_cast<long long*>(counter));
    VSOM_LAUNCH_CHECK("loss_parts_kernel");
}

int vsom_transpose_many(const float* src_base, float* dst_base, const long long* table, int count, int max_rows,
                        int max_cols, vsom_stream_t stream) {
    VSOM_REQUIRE(src_base && dst_base && table && count >= 0 && max_rows > 0 && max_cols > 0, VSOM_EINVAL, "transpose_many: bad arguments");
    if (count == 0) return VSOM_OK;
    VSOM_REQUIRE(count <= 65535, VSOM_EINVAL, "transpose_many: more than 65535 tensors");
    dim3 grid(cdiv(max_rows, 32) * cdiv(max_cols, 32), count);
    VSOM_LAUNCH(transpose_many_kernel, grid, dim3(256), 0, stream, src_base, dst_base, table);
    VSOM_LAUNCH_CHECK("transpose_many_kernel");
}

int vsom_patch_embed_fwd(const float* img, const float* Wpe, const float* bpe, const float* pos,
                         const float* cls_token, float* tokens, float* xp_ws, int B, int C, int S, int p, int E,
                         vsom_stream_t stream) {
    VSOM_REQUIRE(img && Wpe && bpe && pos && cls_token && tokens && xp_ws, VSOM_EINVAL, "patch_embed_fwd: null pointer");
    VSOM_REQUIRE(B > 0 && C > 0 && S > 0 && p > 0 && E > 0 && S % p == 0, VSOM_EINVAL, "patch_embed_fwd: bad shape");
    const int g = S / p, n = g * g, Ntok = n + 1, pd = C * p * p;
    const long total = (long)B * C * S * S;
    VSOM_LAUNCH(patch_gather_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, img, xp_ws, total, C, S, p, g);
    int rc = hip_status(hipGetLastError(), "patch_gather_kernel");
    if (rc) return rc;
    // tokens[b, 1+pi, :] = xp[b*n+pi, :] Wpe^T + bpe + pos[1+pi, :]
    GemmP q = {};
    q.A = xp_ws; q.lda = pd; q.B = Wpe; q.ldb = pd; q.C = tokens; q.ldc = E;
    q.M = B * n; q.N = E; q.K = pd; q.bias = bpe;
    q.R = pos; q.ldr = E; q.r_mod = n; q.r_off = 1;
    q.c_seg = n; q.c_stride = Ntok; q.c_off = 1;
    rc = launch_gemm(true, true, EPI_BIAS_RES, q, 1, stream);
    if (rc) return rc;
    VSOM_LAUNCH(cls_rows_kernel, dim3(cdiv((long)B * E, 256)), dim3(256), 0, stream, cls_token, pos, tokens, B, Ntok, E);
    VSOM_LAUNCH_CHECK("cls_rows_kernel");
}

size_t vsom_patch_embed_bwd_workspace_bytes(int B, int C, int S, int p, int E) {
    if (B <= 0 || C <= 0 || S <= 0 || p <= 0 || E <= 0) return 0;
    const int n = (S / p) * (S / p);
    return vsom_linear_bwd_weight_workspace_bytes(B * n, E, C * p * p);
}

int vsom_patch_embed_bwd(const float* dtokens, const float* xp_ws, float* dWpe, float* dbpe, float* dcls_token,
                         int B, int C, int S, int p, int E, void* ws, size_t ws_bytes, vsom_stream_t stream) {
    VSOM_REQUIRE(dtokens && xp_ws && dWpe && dbpe && dcls_token, VSOM_EINVAL, "patch_embed_bwd: null pointer");
    VSOM_REQUIRE(B > 0 && C > 0 && S > 0 && p > 0 && E > 0 && S % p == 0, VSOM_EINVAL, "patch_embed_bwd: bad shape");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_patch_embed_bwd_workspace_bytes(B, C, S, p, E), VSOM_EWORKSPACE, "patch_embed_bwd: workspace too small");
    const int g = S / p, n = g * g, Ntok = n + 1, pd = C * p * p;
    // dWpe[E, pd] = sum over patch rows of dtokens[row]^T xp[row]; the CLS rows are skipped by the
    // reduction-row map of the k-strided A operand.
    const int M = B * n;
    int rc = linear_bwd_weight_impl(dtokens, E, xp_ws, pd, dWpe, dbpe, M, E, pd, n, Ntok, 1, ws, ws_bytes, stream);
    if (rc) return rc;
    VSOM_LAUNCH(cls_grad_kernel, dim3(cdiv(E, 32)), dim3(256), 0, stream, dtokens, dcls_token, B, Ntok, E);
    VSOM_LAUNCH_CHECK("cls_grad_kernel");
}

size_t vsom_l1_unpatchify_workspace_bytes(int B, int C, int S, int p) {
    if (B <= 0 || C <= 0 || S <= 0 || p <= 0) return 0;
    const int g = S / p;
    return (size_t)l1_blocks((long)B * (g * g + 1) * p * p * C) * sizeof(float);
}

size_t vsom_l1_loss_workspace_bytes(long n) { return n > 0 ? (size_t)l1_blocks(n) * sizeof(float) : 0; }

int vsom_l1_loss(const float* pred, const float* target, float* loss_sum, float* dpred, float grad_scale, long n,
                 void* ws, size_t ws_bytes, vsom_stream_t stream) {
    VSOM_REQUIRE(pred && target && loss_sum && n > 0, VSOM_EINVAL, "l1_loss: bad arguments");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_l1_loss_workspace_bytes(n), VSOM_EWORKSPACE, "l1_loss: workspace too small");
    const int nblk = l1_blocks(n);
    float* part = static_cast<float*>(ws);
    VSOM_LAUNCH(l1_loss_kernel, dim3(nblk), dim3(256), 0, stream, pred, target, part, dpred, grad_scale, n);
    int rc = hip_status(hipGetLastError(), "l1_loss_kernel");
    if (rc) return rc;
    return sum_partials(part, nblk, loss_sum, stream);
}

int vsom_l1_unpatchify(const float* pred, const float* img, float* recon, float* loss_sum, float* dpred,
                       float grad_scale, int B, int C, int S, int p, void* ws, size_t ws_bytes,
                       vsom_stream_t stream) {
    VSOM_REQUIRE(pred && img && loss_sum, VSOM_EINVAL, "l1_unpatchify: null pointer");
    VSOM_REQUIRE(B > 0 && C > 0 && S > 0 && p > 0 && S % p == 0, VSOM_EINVAL, "l1_unpatchify: bad shape");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_l1_unpatchify_workspace_bytes(B, C, S, p), VSOM_EWORKSPACE, "l1_unpatchify: workspace too small");
    const int g = S / p;
    const long total = (long)B * (g * g + 1) * p * p * C;
    const int nblk = l1_blocks(total);
    float* part = static_cast<float*>(ws);
    VSOM_LAUNCH(l1_unpatchify_kernel, dim3(nblk), dim3(256), 0, stream, pred, img, recon, part, dpred, grad_scale,
                       total, C, S, p, g);
    int rc = hip_status(hipGetLastError(), "l1_unpatchify_kernel");
    if (rc) return rc;
    return sum_partials(part, nblk, loss_sum, stream);
}

size_t vsom_cross_entropy_ls_workspace_bytes(int B) { return B > 0 ? (size_t)B * sizeof(float) : 0; }

int vsom_cross_entropy_ls(const float* logits, const int64_t* y, float smoothing, float* loss_sum, float* dlogits,
                          float grad_scale, int B, int C, void* ws, size_t ws_bytes, vsom_stream_t stream) {
    VSOM_REQUIRE(logits && y && loss_sum, VSOM_EINVAL, "cross_entropy_ls: null pointer");
    VSOM_REQUIRE(B > 0 && C > 0, VSOM_EINVAL, "cross_entropy_ls: bad shape");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_cross_entropy_ls_workspace_bytes(B), VSOM_EWORKSPACE, "cross_entropy_ls: workspace too small");
    float* part = static_cast<float*>(ws);
    VSOM_LAUNCH(ce_ls_kernel, dim3(cdiv(B, 4)), dim3(256), 0, stream, logits, y, smoothing, part, dlogits,
                       grad_scale, B, C);
    int rc = hip_status(hipGetLastError(), "ce_ls_kernel");
    if (rc) return rc;
    return sum_partials(part, B, loss_sum, stream);
}

int vsom_adamw_step(float* p, const float* g, float* m, float* v, const float* wd_per_chunk, long n, float lr,
                    float beta1, float beta2, float eps, int step, float grad_scale, int adamw,
                    vsom_stream_t stream) {
    VSOM_REQUIRE(p && g && m && v && wd_per_chunk, VSOM_EINVAL, "adamw_step: null pointer");
    VSOM_REQUIRE(n > 0 && n % 256 == 0, VSOM_EINVAL, "adamw_step: n=%ld must be a positive multiple of 256", n);
    VSOM_REQUIRE(step >= 1, VSOM_EINVAL, "adamw_step: step must be >= 1");
    VSOM_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), VSOM_EALIGN, "adamw_step: 16-byte alignment required");
    const long n4 = n / 4;
    VSOM_LAUNCH(adamw_kernel, dim3(grid_for(n4, 256, 8192)), dim3(256), 0, stream, p, g, m, v, wd_per_chunk, n4,
                adamw_constants(lr, beta1, beta2, eps, step, grad_scale, adamw));
    VSOM_LAUNCH_CHECK("adamw_kernel");
}

}  // extern "C"

// ------------------------------------------------------------------ evaluation helpers (tools/evaluation.py)
namespace vsom {
// table[a[i] * nb + b[i]] += 1   (integer atomics: order-independent, bitwise reproducible)
__global__ __launch_bounds__(256) void contingency_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b,
                                                          long n, int na, int nb, unsigned long long* __restrict__ table,
                                                          int* __restrict__ bad) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int64_t x = a[i], y = b[i];
        if (x < 0 || x >= na || y < 0 || y >= nb) { atomicAdd(bad, 1); continue; }
        atomicAdd(table + x * nb + y, 1ULL);
    }
}
// out[r] = first argmax_c X[r, c]   (torch.argmax semantics), one wave per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ X, long ldx, int rows, int cols,
                                                          int64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < cols; c += 64) {
        const float v = X[(long)row * ldx + c];
        if (v > best || (v == best && c < bi)) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}
}  // namespace vsom

extern "C" {

int vsom_contingency(const int64_t* a, const int64_t* b, long n, int na, int nb, unsigned long long* table,
                     int* out_of_range, vsom_stream_t stream) {
    VSOM_REQUIRE(a && b && table && out_of_range, VSOM_EINVAL, "contingency: null pointer");
    VSOM_REQUIRE(n >= 0 && na > 0 && nb > 0, VSOM_EINVAL, "contingency: bad sizes");
    if (n == 0) return VSOM_OK;
    VSOM_LAUNCH(vsom::contingency_kernel, dim3(vsom::grid_for(n, 256, 2048)), dim3(256), 0, stream, a, b, n, na, nb,
                       table, out_of_range);
    VSOM_LAUNCH_CHECK("contingency_kernel");
}

int vsom_argmax_rows(const float* X, long ldx, int rows, int cols, int64_t* out, vsom_stream_t stream) {
    VSOM_REQUIRE(X && out && rows > 0 && cols > 0 && ldx >= cols, VSOM_EINVAL, "argmax_rows: bad arguments");
    VSOM_LAUNCH(vsom::argmax_rows_kernel, dim3(vsom::cdiv(rows, 4)), dim3(256), 0, stream, X, ldx, rows, cols, out);
    VSOM_LAUNCH_CHECK("argmax_rows_kernel");
}

}  // extern "C"
