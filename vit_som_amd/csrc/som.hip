// SOM layer kernels: row norms, BMU distance pass (+ first-argmin), neighbourhood weights /
// loss / backward coefficients, prototype ("neighbourhood accumulator") and input gradients.
#include "gemm_f32.h"

namespace vsom {

// ------------------------------------------------------------------ 1 / max(||row||, eps)
// one wave per row, 16-byte loads; rows are 12-49k floats at the BASELINE configs
__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ X, long ldx, int rows,
                                                           int cols, float eps, float* __restrict__ out,
                                                           int vec, int squared) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = X + (long)row * ldx;
    float s = 0.f;
    if (vec) {
        const int n4 = cols >> 2;
        for (int i = lane; i < n4; i += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
            s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
        }
        for (int i = (n4 << 2) + lane; i < cols; i += 64) s = fmaf(x[i], x[i], s);
    } else {
        for (int i = lane; i < cols; i += 64) s = fmaf(x[i], x[i], s);
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = squared ? s : 1.0f / fmaxf(sqrtf(s), eps);
}

// long rows (the SOM's L = 12288): one WORKGROUP per row, every thread's 16-byte loads issued up
// front (a wave per row leaves a 512-row input on 128 workgroups with one load in flight per lane)
__global__ __launch_bounds__(256) void row_inv_norm_wide_kernel(const float* __restrict__ X, long ldx, int cols, float eps,
                                                                float* __restrict__ out, int squared) {
    const f32x4* x = reinterpret_cast<const f32x4*>(X + (long)blockIdx.x * ldx);
    const int n4 = cols >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = threadIdx.x;
    for (; i + 768 < n4; i += 1024) {
        const f32x4 a = x[i], b = x[i + 256], c = x[i + 512], d = x[i + 768];
        s0 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
        s1 += (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
        s2 += (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
        s3 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    for (; i < n4; i += 256) { const f32x4 a = x[i]; s0 += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]); }
    float s = wave_sum((s0 + s1) + (s2 + s3));
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        out[blockIdx.x] = squared ? s : 1.0f / fmaxf(sqrtf(s), eps);
    }
}

// ------------------------------------------------------------------ BMU finalize
// dist[i,k] = 1 - (sum_s slab[s][i,k]) * inv_nx[i] * inv_nw[k];  bmu[i] = first argmin_k.
// One workgroup per sample row; (value, index) reduction with lowest-index tie-break, so
// bmu is EXACTLY torch.argmin of the distances this kernel writes.
__global__ __launch_bounds__(256) void bmu_finalize_kernel(const float* __restrict__ slab, long slab_stride,
                                                           int nslabs, const float* __restrict__ inv_nx,
                                                           const float* __restrict__ inv_nw,
                                                           float* __restrict__ dist, int64_t* __restrict__ bmu,
                                                           int K, int euclid) {
    const int i = blockIdx.x;
    const float rx = inv_nx ? inv_nx[i] : 0.f;   // cosine: 1/|x_i| ; euclidean: |x_i|^2 ; manhattan (euclid == 2): unused
    float best = INFINITY;
    int bidx = 0x7fffffff;
    // four prototypes per thread and pass: their slab loads are independent and in flight together
    // (one at a time, the nslabs x K/256 dependent loads of a thread are all serialised); the sum over
    // the slabs of ONE (i, k) keeps its order s = 0, 1, ...
    for (int k0 = threadIdx.x; k0 < K; k0 += 1024) {
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
        const float* p = slab + (long)i * K + k0;
        for (int s = 0; s < nslabs; ++s) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + 256 * u < K) dot[u] += p[256 * u];
            p += slab_stride;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 256 * u;
            if (k >= K) break;
            // euclidean: torch.cdist's matmul form  sqrt(clamp_min(|x|^2 + |w|^2 - 2 x.w, 1e-30))
            const float d = euclid == 2 ? dot[u]      // manhattan: the slabs already hold partial distances
                          : euclid ? sqrtf(fmaxf(fmaf(-2.0f, dot[u], rx + inv_nw[k]), 1e-30f)) : 1.0f - dot[u] * rx * inv_nw[k];
            if (dist) dist[(long)i * K + k] = d;
            if (d < best || (d == best && k < bidx)) { best = d; bidx = k; }
        }
    }
    // NaN distances never win (d < best is false) unless every entry is NaN -> index 0x7fffffff
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    __shared__ float sb[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sb[w] < best || (sb[w] == best && si[w] < bidx)) { best = sb[w]; bidx = si[w]; }
        bmu[i] = (bidx == 0x7fffffff) ? 0 : (int64_t)bidx;
    }
}

static int bmu_splits(int B, int K, int L) {
    return choose_splits(cdiv(B, 128) * cdiv(K, 64), cdiv(L, 32), 32, true);
}

// ------------------------------------------------------------------ neighbourhood / loss / coefficients
// block per sample row i
__global__ __launch_bounds__(256) void som_neigh_row_kernel(const float* __restrict__ dist,
                                                            const int64_t* __restrict__ bmu,
                                                            const float* __restrict__ grid, float inv_2T2,
                                                            const float* __restrict__ inv_nx,
                                                            const float* __restrict__ inv_nw, float c,
                                                            float* __restrict__ h_out, float* __restrict__ coef,
                                                            float* __restrict__ row_dot,
                                                            float* __restrict__ loss_part, int K, int euclid,
                                                            const float* __restrict__ h_in) {
    // h_in != null: explicit weights h[i,k] = h_in[i,k] (any tensor; som_loss(weights, distances) and the autograd
    // of the distances themselves) instead of the Gaussian neighbourhood of the BMU
    const int i = blockIdx.x;
    const int64_t b = h_in ? 0 : bmu[i];
    const float by = h_in ? 0.f : grid[2 * b], bx = h_in ? 0.f : grid[2 * b + 1];
    const float rx = (inv_nx && !euclid) ? inv_nx[i] : 0.f;
    float lsum = 0.f, dsum = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        float h;
        if (h_in) {
            h = h_in[(long)i * K + k];
        } else {
            const float dy = grid[2 * k] - by, dx = grid[2 * k + 1] - bx;
            h = expf(-(dy * dy + dx * dx) * inv_2T2);
        }
        const float d = dist[(long)i * K + k];
        if (h_out) h_out[(long)i * K + k] = h;
        if (euclid == 2) {                                      // manhattan: dLoss/d dist = c h
            if (coef) coef[(long)i * K + k] = c * h;
        } else if (euclid) {
            const float hd = (d > 0.f) ? h / d : 0.f;           // d|x-w|/dx = (x-w)/d ; torch gives 0 at d == 0
            if (coef) coef[(long)i * K + k] = -c * hd;
            dsum += hd;
        } else {
            if (coef) coef[(long)i * K + k] = -c * h * rx * inv_nw[k];
            dsum = fmaf(h, 1.0f - d, dsum);
        }
        lsum = fmaf(h, d, lsum);
    }
    lsum = wave_sum(lsum);
    dsum = wave_sum(dsum);
    __shared__ float s1[4], s2[4];
    if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = lsum; s2[threadIdx.x >> 6] = dsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        loss_part[i] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
        if (row_dot) row_dot[i] = c * (euclid ? 1.0f : rx * rx) * ((s2[0] + s2[1]) + (s2[2] + s2[3]));
    }
}
// col_dot[k] = c rw^2 sum_i term(i,k).  A workgroup owns 32 prototype columns; its 32 row groups (16 waves) each take
// every 32nd sample row with four independent accumulators (four rows of loads in flight: with 8 groups and two
// accumulators the kernel was a chain of 32 dependent global-load latencies, 42 us for 3 MB) and are combined through
// LDS in a fixed order -> bitwise reproducible.
constexpr int NCOL_RG = 32;
__global__ __launch_bounds__(NCOL_RG * 32) void som_neigh_col_kernel(const float* __restrict__ dist,
                                                                     const int64_t* __restrict__ bmu,
                                                                     const float* __restrict__ grid, float inv_2T2,
                                                                     const float* __restrict__ inv_nw, float c,
                                                                     float* __restrict__ col_dot, int B, int K, int euclid,
                                                                     const float* __restrict__ h_in) {
    __shared__ float part[NCOL_RG][32];
    const int cidx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + cidx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (k < K) {
        const float gy = h_in ? 0.f : grid[2 * k], gx = h_in ? 0.f : grid[2 * k + 1];
        auto term = [&](int i) {
            float h;
            if (h_in) {
                h = h_in[(long)i * K + k];
            } else {
                const int64_t b = bmu[i];
                const float dy = gy - grid[2 * b], dx = gx - grid[2 * b + 1];
                h = expf(-(dy * dy + dx * dx) * inv_2T2);
            }
            const float d = dist[(long)i * K + k];
            return euclid ? ((d > 0.f) ? h / d : 0.f) : h * (1.0f - d);
        };
        int i = rg;
        for (; i + 3 * NCOL_RG < B; i += 4 * NCOL_RG) {
            s0 += term(i); s1 += term(i + NCOL_RG); s2 += term(i + 2 * NCOL_RG); s3 += term(i + 3 * NCOL_RG);
        }
        for (; i < B; i += NCOL_RG) s0 += term(i);
    }
    part[rg][cidx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && k < K) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCOL_RG; ++j) s += part[j][cidx];
        const float rw = euclid ? 1.0f : inv_nw[k];
        col_dot[k] = c * rw * rw * s;
    }
}
// deterministic sum of n partials into out[0] (single workgroup)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int n,
                                                           float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = wave_sum(s);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

int bmu_finalize_plain(const float* slab, long slab_stride, int nslabs, float* dist, int64_t* bmu, int B, int K,
                       hipStream_t stream) {
    VSOM_LAUNCH(bmu_finalize_kernel, dim3(B), dim3(256), 0, stream, slab, slab_stride, nslabs, (const float*)nullptr,
                       (const float*)nullptr, dist, bmu, K, 2);
    VSOM_LAUNCH_CHECK("bmu_finalize_kernel");
}

int sum_partials(const float* part, int n, float* out, hipStream_t stream) {
    VSOM_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, stream, part, n, out);
    VSOM_LAUNCH_CHECK("sum_partials_kernel");
}

}  // namespace vsom

using namespace vsom;

extern "C" {

int vsom_row_inv_norm(const float* X, long ldx, int rows, int cols, float eps, float* inv_norm,
                      vsom_stream_t stream) {
    VSOM_REQUIRE(X && inv_norm && rows > 0 && cols > 0 && ldx >= cols, VSOM_EINVAL, "row_inv_norm: bad arguments");
    const int vec = aligned16(X) && (ldx % 4 == 0);
    if (vec && cols % 4 == 0 && cols >= 4096) {
        VSOM_LAUNCH(row_inv_norm_wide_kernel, dim3(rows), dim3(256), 0, stream, X, ldx, cols, eps, inv_norm, 0);
        VSOM_LAUNCH_CHECK("row_inv_norm_wide_kernel");
    }
    VSOM_LAUNCH(row_inv_norm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, X, ldx, rows, cols, eps,
                       inv_norm, vec, 0);
    VSOM_LAUNCH_CHECK("row_inv_norm_kernel");
}

int vsom_row_sqnorm(const float* X, long ldx, int rows, int cols, float* sqnorm, vsom_stream_t stream) {
    VSOM_REQUIRE(X && sqnorm && rows > 0 && cols > 0 && ldx >= cols, VSOM_EINVAL, "row_sqnorm: bad arguments");
    const int vec = aligned16(X) && (ldx % 4 == 0);
    if (vec && cols % 4 == 0 && cols >= 4096) {
        VSOM_LAUNCH(row_inv_norm_wide_kernel, dim3(rows), dim3(256), 0, stream, X, ldx, cols, 0.f, sqnorm, 1);
        VSOM_LAUNCH_CHECK("row_inv_norm_wide_kernel");
    }
    VSOM_LAUNCH(row_inv_norm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, X, ldx, rows, cols, 0.f, sqnorm,
                       vec, 1);
    VSOM_LAUNCH_CHECK("row_inv_norm_kernel");
}

size_t vsom_bmu_cosine_workspace_bytes(int B, int K, int L) {
    if (B <= 0 || K <= 0 || L <= 0) return 0;
    return (size_t)bmu_splits(B, K, L) * (size_t)B * K * sizeof(float);
}

int vsom_bmu_cosine_dots(const float* X, long ldx, const float* W, int B, int K, int L, void* ws, size_t ws_bytes,
                         vsom_stream_t stream) {
    VSOM_REQUIRE(X && W, VSOM_EINVAL, "bmu_cosine_dots: null pointer");
    VSOM_REQUIRE(B > 0 && K > 0 && L > 0 && ldx >= L, VSOM_EINVAL, "bmu_cosine_dots: bad shape B=%d K=%d L=%d ldx=%ld", B, K, L, ldx);
    VSOM_REQUIRE(ws && ws_bytes >= vsom_bmu_cosine_workspace_bytes(B, K, L), VSOM_EWORKSPACE,
                 "bmu_cosine_dots: workspace too small");
    GemmP g = {};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = L;
    g.M = B; g.N = K; g.K = L;
    g.slab = static_cast<float*>(ws); g.slab_stride = (long)B * K;
    return launch_gemm(true, true, EPI_SLAB, g, bmu_splits(B, K, L), stream);
}

int vsom_bmu_cosine_finalize(const void* ws, size_t ws_bytes, const float* inv_nx, const float* inv_nw, float* dist,
                             int64_t* bmu, int B, int K, int L, vsom_stream_t stream) {
    VSOM_REQUIRE(inv_nx && inv_nw && bmu, VSOM_EINVAL, "bmu_cosine_finalize: null pointer");
    VSOM_REQUIRE(B > 0 && K > 0 && L > 0, VSOM_EINVAL, "bmu_cosine_finalize: bad shape");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_bmu_cosine_workspace_bytes(B, K, L), VSOM_EWORKSPACE,
                 "bmu_cosine_finalize: workspace too small");
    VSOM_LAUNCH(bmu_finalize_kernel, dim3(B), dim3(256), 0, stream, (const float*)ws, (long)B * K,
                       bmu_splits(B, K, L), inv_nx, inv_nw, dist, bmu, K, 0);
    VSOM_LAUNCH_CHECK("bmu_finalize_kernel");
}

int vsom_bmu_euclid_fwd(const float* X, long ldx, const float* W, const float* sq_x, const float* sq_w, float* dist,
                        int64_t* bmu, int B, int K, int L, void* ws, size_t ws_bytes, vsom_stream_t stream) {
    VSOM_REQUIRE(sq_x && sq_w && bmu, VSOM_EINVAL, "bmu_euclid_fwd: null pointer");
    int rc = vsom_bmu_cosine_dots(X, ldx, W, B, K, L, ws, ws_bytes, stream);      // the same X.W^T contraction
    if (rc) return rc;
    VSOM_LAUNCH(bmu_finalize_kernel, dim3(B), dim3(256), 0, stream, (const float*)ws, (long)B * K,
                       bmu_splits(B, K, L), sq_x, sq_w, dist, bmu, K, 1);
    VSOM_LAUNCH_CHECK("bmu_finalize_kernel");
}

int vsom_bmu_cosine_fwd(const float* X, long ldx, const float* W, const float* inv_nx, const float* inv_nw,
                        float* dist, int64_t* bmu, int B, int K, int L, void* ws, size_t ws_bytes,
                        vsom_stream_t stream) {
    VSOM_REQUIRE(inv_nx && inv_nw && bmu, VSOM_EINVAL, "bmu_cosine_fwd: null pointer");
    int rc = vsom_bmu_cosine_dots(X, ldx, W, B, K, L, ws, ws_bytes, stream);
    if (rc) return rc;
    return vsom_bmu_cosine_finalize(ws, ws_bytes, inv_nx, inv_nw, dist, bmu, B, K, L, stream);
}

size_t vsom_som_neigh_workspace_bytes(int B, int K) { (void)K; return B > 0 ? (size_t)B * sizeof(float) : 0; }

int vsom_som_neigh_loss(const float* dist, const int64_t* bmu, const float* grid, float T, const float* inv_nx,
                        const float* inv_nw, float grad_scale, float* h, float* loss_sum, float* coef,
                        float* row_dot, float* col_dot, int B, int K, int distance, void* ws, size_t ws_bytes,
                        vsom_stream_t stream) {
    VSOM_REQUIRE(distance == VSOM_DIST_COSINE || distance == VSOM_DIST_EUCLIDEAN || distance == VSOM_DIST_MANHATTAN,
                 VSOM_EUNSUPPORTED, "som_neigh_loss: distance %d not supported", distance);
    const int euclid = distance == VSOM_DIST_EUCLIDEAN ? 1 : (distance == VSOM_DIST_MANHATTAN ? 2 : 0);
    VSOM_REQUIRE(dist && bmu && grid && loss_sum, VSOM_EINVAL, "som_neigh_loss: null pointer");
    VSOM_REQUIRE(B > 0 && K > 0 && T > 0.f, VSOM_EINVAL, "som_neigh_loss: bad shape/temperature");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_som_neigh_workspace_bytes(B, K), VSOM_EWORKSPACE, "som_neigh_loss: workspace too small");
    const bool bwd = coef || row_dot || col_dot;
    VSOM_REQUIRE(!bwd || (euclid == 2 && coef) || (coef && row_dot && col_dot && (euclid || (inv_nx && inv_nw))), VSOM_EINVAL,
                 "som_neigh_loss: backward outputs need coef, row_dot, col_dot (and inv_nx, inv_nw for cosine) together");
    const float inv_2T2 = (float)(1.0 / (2.0 * (double)T * (double)T));
    float* part = static_cast<float*>(ws);
    VSOM_LAUNCH(som_neigh_row_kernel, dim3(B), dim3(256), 0, stream, dist, bmu, grid, inv_2T2, inv_nx, inv_nw,
                       grad_scale, h, coef, row_dot, part, K, euclid, (const float*)nullptr);
    int rc = hip_status(hipGetLastError(), "som_neigh_row_kernel");
    if (rc) return rc;
    rc = sum_partials(part, B, loss_sum, stream);
    if (rc) return rc;
    if (bwd && euclid != 2) {
        VSOM_LAUNCH(som_neigh_col_kernel, dim3(cdiv(K, 32)), dim3(NCOL_RG * 32), 0, stream, dist, bmu, grid, inv_2T2,
                           inv_nw, grad_scale, col_dot, B, K, euclid, (const float*)nullptr);
        rc = hip_status(hipGetLastError(), "som_neigh_col_kernel");
    }
    return rc;
}

int vsom_som_weighted_loss(const float* dist, const float* weights, const float* inv_nx, const float* inv_nw,
                           float grad_scale, float* loss_sum, float* coef, float* row_dot, float* col_dot, int B, int K,
                           int distance, void* ws, size_t ws_bytes, vsom_stream_t stream) {
    VSOM_REQUIRE(distance == VSOM_DIST_COSINE || distance == VSOM_DIST_EUCLIDEAN || distance == VSOM_DIST_MANHATTAN,
                 VSOM_EUNSUPPORTED, "som_weighted_loss: distance %d not supported", distance);
    const int euclid = distance == VSOM_DIST_EUCLIDEAN ? 1 : (distance == VSOM_DIST_MANHATTAN ? 2 : 0);
    VSOM_REQUIRE(dist && weights && loss_sum, VSOM_EINVAL, "som_weighted_loss: null pointer");
    VSOM_REQUIRE(B > 0 && K > 0, VSOM_EINVAL, "som_weighted_loss: bad shape");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_som_neigh_workspace_bytes(B, K), VSOM_EWORKSPACE, "som_weighted_loss: workspace too small");
    const bool bwd = coef || row_dot || col_dot;
    VSOM_REQUIRE(!bwd || (euclid == 2 && coef) || (coef && row_dot && col_dot && (euclid || (inv_nx && inv_nw))), VSOM_EINVAL,
                 "som_weighted_loss: backward outputs need coef, row_dot, col_dot (and inv_nx, inv_nw for cosine) together");
    float* part = static_cast<float*>(ws);
    VSOM_LAUNCH(som_neigh_row_kernel, dim3(B), dim3(256), 0, stream, dist, (const int64_t*)nullptr, (const float*)nullptr,
                       0.f, inv_nx, inv_nw, grad_scale, (float*)nullptr, coef, row_dot, part, K, euclid, weights);
    int rc = hip_status(hipGetLastError(), "som_neigh_row_kernel");
    if (rc) return rc;
    rc = sum_partials(part, B, loss_sum, stream);
    if (rc) return rc;
    if (bwd && euclid != 2) {
        VSOM_LAUNCH(som_neigh_col_kernel, dim3(cdiv(K, 32)), dim3(NCOL_RG * 32), 0, stream, dist, (const int64_t*)nullptr,
                           (const float*)nullptr, 0.f, inv_nw, grad_scale, col_dot, B, K, euclid, weights);
        rc = hip_status(hipGetLastError(), "som_neigh_col_kernel");
    }
    return rc;
}

int vsom_som_bwd(const float* X, long ldx, const float* W, const float* coef, const float* row_dot,
                 const float* col_dot, float* gW, float* gX, long ldgx, int accumulate_gx, int B, int K, int L,
                 vsom_stream_t stream) {
    VSOM_REQUIRE(X && W && coef && row_dot && col_dot && gW && gX, VSOM_EINVAL, "som_bwd: null pointer");
    VSOM_REQUIRE(B > 0 && K > 0 && L > 0 && ldx >= L && ldgx >= L, VSOM_EINVAL, "som_bwd: bad shape");
    // gW[K,L] = coef^T[K,B] X[B,L] + col_dot[k] W[k,:]      (reduction over the batch)
    GemmP g = {};
    g.A = coef; g.lda = K; g.B = X; g.ldb = ldx; g.C = gW; g.ldc = L;
    g.M = K; g.N = L; g.K = B;
    g.rowscale = col_dot; g.R = W; g.ldr = L; g.accumulate = 0;
    g.products = gemm_grad_products();          // gradient GEMMs: three products in the default mode (gemm_f32.hip)
    int rc = launch_gemm(false, false, EPI_ROWAXPY, g, 1, stream);
    if (rc) return rc;
    // gX[B,L] (+)= coef[B,K] W[K,L] + row_dot[i] X[i,:]     (reduction over the prototypes)
    GemmP q = {};
    q.A = coef; q.lda = K; q.B = W; q.ldb = L; q.C = gX; q.ldc = ldgx;
    q.M = B; q.N = L; q.K = K;
    q.rowscale = row_dot; q.R = X; q.ldr = ldx; q.accumulate = accumulate_gx;
    q.products = gemm_grad_products();
    return launch_gemm(true, false, EPI_ROWAXPY, q, 1, stream);
}

}  // extern "C"
