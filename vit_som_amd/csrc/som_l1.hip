// Manhattan (L1) variant of the SOM layer: torch.cdist(x, prototypes, p=1) (models/som_layer.py:115-116,
// the DESOM configs' distance) and its autograd.  |x - w| does not factor through a dot product, so
// these are tiled VALU kernels (no MFMA): a 64 x 64 output tile per workgroup, 4 x 4 outputs per
// thread, both operand tiles staged through LDS in 32-deep chunks of the reduction index.
//   dist[i,k]  = sum_l |x[i,l] - w[k,l]|                       (reduction over l, split into slabs)
//   gX[i,l]   += sum_k coef[i,k] sign(x[i,l] - w[k,l])         (reduction over k)
//   gW[k,l]    = -sum_i coef[i,k] sign(x[i,l] - w[k,l])        (reduction over i)
// with coef = dLoss/d dist (vsom_som_neigh_loss, distance = VSOM_DIST_MANHATTAN); sign(0) = 0 as in torch.
#include "common.h"

namespace vsom {

constexpr int L1_LD = 68;        // LDS row stride (floats) of a 64-wide tile row

// Stage a [64 rows x 32 reduction] block whose rows are contiguous along the reduction index,
// TRANSPOSED into lds[32][L1_LD] (lds[c][r] = src[(row0 + r) * ld + c0 + c]); zero outside.
__device__ __forceinline__ void l1_stage_t(const float* __restrict__ src, long ld, int row0, int nrows, int c0, int ncols,
                                           float* lds, int t, bool vec) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = p * 32 + (t >> 3), c = (t & 7) << 2;
        const int row = row0 + r, col = c0 + c;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < nrows && col < ncols) {
            const float* s = src + (long)row * ld + col;
            if (vec && col + 3 < ncols) {
                v = *reinterpret_cast<const f32x4*>(s);
            } else {
                v[0] = s[0];
                if (col + 1 < ncols) v[1] = s[1];
                if (col + 2 < ncols) v[2] = s[2];
                if (col + 3 < ncols) v[3] = s[3];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[(c + e) * L1_LD + r] = v[e];
    }
}
// Stage a [32 reduction x 64 cols] block whose rows are reduction indices (natural order):
// lds[r][c] = src[(r0 + r) * ld + col0 + c]; zero outside.
__device__ __forceinline__ void l1_stage_n(const float* __restrict__ src, long ld, int r0, int nrows, int col0, int ncols,
                                           float* lds, int t, bool vec) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = p * 16 + (t >> 4), c = (t & 15) << 2;
        const int row = r0 + r, col = col0 + c;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < nrows && col < ncols) {
            const float* s = src + (long)row * ld + col;
            if (vec && col + 3 < ncols) {
                v = *reinterpret_cast<const f32x4*>(s);
            } else {
                v[0] = s[0];
                if (col + 1 < ncols) v[1] = s[1];
                if (col + 2 < ncols) v[2] = s[2];
                if (col + 3 < ncols) v[3] = s[3];
            }
        }
        *reinterpret_cast<f32x4*>(lds + r * L1_LD + c) = v;
    }
}

// slab[z][i,k] = sum over this split's l of |x[i,l] - w[k,l]|
__global__ __launch_bounds__(256) void l1_dist_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W,
                                                      float* __restrict__ slab, long slab_stride, int B, int K, int L,
                                                      int chunks_per_split, int vx, int vw) {
    __shared__ __attribute__((aligned(16))) float Xs[32 * L1_LD], Ws[32 * L1_LD];
    const int t = threadIdx.x, ti = t >> 4, tk = t & 15;
    const int tiles_k = (K + 63) >> 6;
    const int i0 = ((int)blockIdx.x / tiles_k) << 6, k0 = ((int)blockIdx.x % tiles_k) << 6, z = blockIdx.y;
    const int nchunks = (L + 31) >> 5;
    const int c_begin = z * chunks_per_split;
    int c_end = c_begin + chunks_per_split;
    if (c_end > nchunks) c_end = nchunks;
    float acc[4][4] = {};
    for (int ch = c_begin; ch < c_end; ++ch) {
        l1_stage_t(X, ldx, i0, B, ch << 5, L, Xs, t, vx);
        l1_stage_t(W, L, k0, K, ch << 5, L, Ws, t, vw);
        __syncthreads();
#pragma unroll 8
        for (int l = 0; l < 32; ++l) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(Xs + l * L1_LD + 4 * ti);
            const f32x4 b = *reinterpret_cast<const f32x4*>(Ws + l * L1_LD + 4 * tk);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] += fabsf(a[r] - b[c]);
        }
        __syncthreads();
    }
    float* out = slab + (long)z * slab_stride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * ti + r;
        if (i >= B) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + 4 * tk + c;
            if (k < K) out[(long)i * K + k] = acc[r][c];
        }
    }
}

__device__ __forceinline__ float l1_signed(float d, float c) { return d > 0.f ? c : (d < 0.f ? -c : 0.f); }

// gX[i,l] (+)= sum_k coef[i,k] sign(x[i,l] - w[k,l])
__global__ __launch_bounds__(256) void l1_bwd_x_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W,
                                                       const float* __restrict__ coef, float* __restrict__ gX, long ldgx,
                                                       int accumulate, int B, int K, int L, int vw, int vc) {
    __shared__ __attribute__((aligned(16))) float Cs[32 * L1_LD], Ws[32 * L1_LD];
    const int t = threadIdx.x, ti = t >> 4, tl = t & 15;
    const int tiles_l = (L + 63) >> 6;
    const int i0 = ((int)blockIdx.x / tiles_l) << 6, l0 = ((int)blockIdx.x % tiles_l) << 6;
    float x[4][4], acc[4][4] = {};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = i0 + 4 * ti + r, l = l0 + 4 * tl + c;
            x[r][c] = (i < B && l < L) ? X[(long)i * ldx + l] : 0.f;
        }
    for (int kc = 0; kc < K; kc += 32) {
        l1_stage_t(coef, K, i0, B, kc, K, Cs, t, vc);          // Cs[k][i]
        l1_stage_n(W, L, kc, K, l0, L, Ws, t, vw);             // Ws[k][l]
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < 32; ++kk) {
            const f32x4 cv = *reinterpret_cast<const f32x4*>(Cs + kk * L1_LD + 4 * ti);
            const f32x4 w = *reinterpret_cast<const f32x4*>(Ws + kk * L1_LD + 4 * tl);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] += l1_signed(x[r][c] - w[c], cv[r]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * ti + r;
        if (i >= B) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int l = l0 + 4 * tl + c;
            if (l >= L) continue;
            float* d = gX + (long)i * ldgx + l;
            *d = accumulate ? *d + acc[r][c] : acc[r][c];
        }
    }
}

// gW[k,l] = -sum_i coef[i,k] sign(x[i,l] - w[k,l])
__global__ __launch_bounds__(256) void l1_bwd_w_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W,
                                                       const float* __restrict__ coef, float* __restrict__ gW, int B, int K,
                                                       int L, int vx, int vc) {
    __shared__ __attribute__((aligned(16))) float Cs[32 * L1_LD], Xs[32 * L1_LD];
    const int t = threadIdx.x, tk = t >> 4, tl = t & 15;
    const int tiles_l = (L + 63) >> 6;
    const int k0 = ((int)blockIdx.x / tiles_l) << 6, l0 = ((int)blockIdx.x % tiles_l) << 6;
    float w[4][4], acc[4][4] = {};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + 4 * tk + r, l = l0 + 4 * tl + c;
            w[r][c] = (k < K && l < L) ? W[(long)k * L + l] : 0.f;
        }
    for (int ic = 0; ic < B; ic += 32) {
        l1_stage_n(coef, K, ic, B, k0, K, Cs, t, vc);          // Cs[i][k]
        l1_stage_n(X, ldx, ic, B, l0, L, Xs, t, vx);           // Xs[i][l]
        __syncthreads();
#pragma unroll 4
        for (int ii = 0; ii < 32; ++ii) {
            const f32x4 cv = *reinterpret_cast<const f32x4*>(Cs + ii * L1_LD + 4 * tk);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(Xs + ii * L1_LD + 4 * tl);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] -= l1_signed(xv[c] - w[r][c], cv[r]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + 4 * tk + r;
        if (k >= K) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int l = l0 + 4 * tl + c;
            if (l < L) gW[(long)k * L + l] = acc[r][c];
        }
    }
}

// split count of the distance pass: enough workgroups to fill the chip (256 CUs x several)
static int l1_splits(int B, int K, int L) {
    const long tiles = (long)cdiv(B, 64) * cdiv(K, 64);
    const int nchunks = cdiv(L, 32);
    long s = (1024 + tiles - 1) / tiles;
    if (s > 16) s = 16;
    if (s > nchunks) s = nchunks;
    if (s < 1) s = 1;
    const int per = cdiv(nchunks, (int)s);
    return cdiv(nchunks, per);
}

int bmu_finalize_plain(const float* slab, long slab_stride, int nslabs, float* dist, int64_t* bmu, int B, int K,
                       hipStream_t stream);      // som.hip

}  // namespace vsom

using namespace vsom;

extern "C" {

size_t vsom_bmu_manhattan_workspace_bytes(int B, int K, int L) {
    if (B <= 0 || K <= 0 || L <= 0) return 0;
    return (size_t)l1_splits(B, K, L) * (size_t)B * K * sizeof(float);
}

int vsom_bmu_manhattan_fwd(const float* X, long ldx, const float* W, float* dist, int64_t* bmu, int B, int K, int L,
                           void* ws, size_t ws_bytes, vsom_stream_t stream) {
    VSOM_REQUIRE(X && W && bmu, VSOM_EINVAL, "bmu_manhattan_fwd: null pointer");
    VSOM_REQUIRE(B > 0 && K > 0 && L > 0 && ldx >= L, VSOM_EINVAL, "bmu_manhattan_fwd: bad shape B=%d K=%d L=%d ldx=%ld", B, K, L, ldx);
    VSOM_REQUIRE(ws && ws_bytes >= vsom_bmu_manhattan_workspace_bytes(B, K, L), VSOM_EWORKSPACE,
                 "bmu_manhattan_fwd: workspace too small");
    const int s = l1_splits(B, K, L);
    const int per = cdiv(cdiv(L, 32), s);
    dim3 grid(cdiv(B, 64) * cdiv(K, 64), s);
    VSOM_LAUNCH(l1_dist_kernel, grid, dim3(256), 0, stream, X, ldx, W, static_cast<float*>(ws), (long)B * K, B, K, L,
                       per, (int)(aligned16(X) && ldx % 4 == 0), (int)(aligned16(W) && L % 4 == 0));
    int rc = hip_status(hipGetLastError(), "l1_dist_kernel");
    if (rc) return rc;
    return bmu_finalize_plain(static_cast<const float*>(ws), (long)B * K, s, dist, bmu, B, K, stream);
}

int vsom_som_bwd_manhattan(const float* X, long ldx, const float* W, const float* coef, float* gW, float* gX,
                           long ldgx, int accumulate_gx, int B, int K, int L, vsom_stream_t stream) {
    VSOM_REQUIRE(X && W && coef && gW && gX, VSOM_EINVAL, "som_bwd_manhattan: null pointer");
    VSOM_REQUIRE(B > 0 && K > 0 && L > 0 && ldx >= L && ldgx >= L, VSOM_EINVAL, "som_bwd_manhattan: bad shape");
    const int vx = aligned16(X) && ldx % 4 == 0, vw = aligned16(W) && L % 4 == 0, vc = aligned16(coef) && K % 4 == 0;
    VSOM_LAUNCH(l1_bwd_w_kernel, dim3(cdiv(K, 64) * cdiv(L, 64)), dim3(256), 0, stream, X, ldx, W, coef, gW, B, K, L, vx, vc);
    int rc = hip_status(hipGetLastError(), "l1_bwd_w_kernel");
    if (rc) return rc;
    VSOM_LAUNCH(l1_bwd_x_kernel, dim3(cdiv(B, 64) * cdiv(L, 64)), dim3(256), 0, stream, X, ldx, W, coef, gX, ldgx,
                       accumulate_gx, B, K, L, vw, vc);
    VSOM_LAUNCH_CHECK("l1_bwd_x_kernel");
}

}  // extern "C"
