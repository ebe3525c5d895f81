// LayerNorm forward / backward: one wave per token row, row held in registers, 16 B-free
// scalar-coalesced accesses (rows are 4..1024 floats; E = 192 -> 3 values per lane).
#include "gemm_f32.h"

namespace vsom {

template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ X,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ Y, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int rows, int cols,
                                                            float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = X + (long)row * cols;
    float v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        v[j] = (c < cols) ? x[c] : 0.f;
        s += v[j];
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        const float d = (c < cols) ? v[j] - mu : 0.f;
        q = fmaf(d, d, q);
    }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
    float* y = Y + (long)row * cols;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c < cols) y[c] = (v[j] - mu) * rs * gamma[c] + beta[c];
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dX = resid + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dY * gamma
// per-workgroup partial sums of dgamma = sum dY*xhat and dbeta = sum dY go to ws[blk][2][cols]
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dY,
                                                            const float* __restrict__ X,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ resid,
                                                            float* __restrict__ dX, float* __restrict__ part,
                                                            int rows, int cols) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // [4 waves][2][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gam[MAXV], dg[MAXV], db[MAXV];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        gam[j] = (c < cols) ? gamma[c] : 0.f;
        dg[j] = 0.f; db[j] = 0.f;
    }
    const float inv_n = 1.0f / (float)cols;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        const float* x = X + (long)row * cols;
        const float* dy = dY + (long)row * cols;
        float xh[MAXV], g[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = lane + 64 * j;
            const bool ok = c < cols;
            const float d = ok ? dy[c] : 0.f;
            xh[j] = ok ? (x[c] - mu) * rs : 0.f;
            g[j] = d * gam[j];
            s1 += g[j];
            s2 = fmaf(g[j], xh[j], s2);
            dg[j] = fmaf(d, xh[j], dg[j]);
            db[j] += d;
        }
        s1 = wave_sum(s1) * inv_n;
        s2 = wave_sum(s2) * inv_n;
        float* dx = dX + (long)row * cols;
        const float* rr = resid ? resid + (long)row * cols : nullptr;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int c = lane + 64 * j;
            if (c < cols) {
                float val = rs * (g[j] - s1 - xh[j] * s2);
                if (rr) val += rr[c];
                dx[c] = val;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int c = lane + 64 * j;
        if (c < cols) { sh[(wave * 2 + 0) * cols + c] = dg[j]; sh[(wave * 2 + 1) * cols + c] = db[j]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * cols; c += 256) {
        const float s = (sh[c] + sh[2 * cols + c]) + (sh[4 * cols + c] + sh[6 * cols + c]);
        part[(long)blockIdx.x * 2 * cols + c] = s;
    }
}

// ---- 16-byte variants for rows of cols <= 64 * NCH floats, cols % 4 == 0 (E = 192 -> NCH = 3, the
// decoder's 96 -> NCH = 2 with half of the second chunk masked): 16 lanes per row, 4 rows per wave,
// each lane holds NCH float4 chunks (chunk j of lane l = columns 4 (l + 16 j) ..) -- four times the
// bytes in flight per wave of the scalar kernels above.
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    return v;
}

template <int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_v4_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ Y,
                                                               float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                               int cols, float eps) {
    const int sub = threadIdx.x & 15;
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = row < rows;
    const f32x4* x = reinterpret_cast<const f32x4*>(X + (long)(ok ? row : 0) * cols);
    const float inv_n = 1.0f / (float)cols;
    f32x4 v[NCH];
    bool cv[NCH];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        cv[j] = 4 * (sub + 16 * j) < cols;
        v[j] = cv[j] ? x[sub + 16 * j] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mu = group16_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = cv[j] ? v[j][e] - mu : 0.f; q = fmaf(d, d, q); }
    const float rs = rsqrtf(group16_sum(q) * inv_n + eps);
    if (!ok) return;
    f32x4* y = reinterpret_cast<f32x4*>(Y + (long)row * cols);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (!cv[j]) continue;
        const f32x4 g4 = reinterpret_cast<const f32x4*>(gamma)[sub + 16 * j];
        const f32x4 b4 = reinterpret_cast<const f32x4*>(beta)[sub + 16 * j];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mu) * rs * g4[e] + b4[e];
        y[sub + 16 * j] = o;
    }
    if (sub == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_v4_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ resid,
                                                               float* __restrict__ dX, float* __restrict__ part, int rows,
                                                               int cols) {
    __shared__ __attribute__((aligned(16))) float sh[16 * 2 * 64 * NCH];      // [row group][dgamma | dbeta][cols]
    const int sub = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const float inv_n = 1.0f / (float)cols;
    f32x4 gam[NCH], dg[NCH], db[NCH];
    bool cv[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        cv[j] = 4 * (sub + 16 * j) < cols;
        gam[j] = cv[j] ? reinterpret_cast<const f32x4*>(gamma)[sub + 16 * j] : f32x4{0.f, 0.f, 0.f, 0.f};
        dg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row0 = blockIdx.x * 16; row0 < rows; row0 += gridDim.x * 16) {
        const int row = row0 + rg;
        const bool ok = row < rows;
        const long base = (long)(ok ? row : 0) * cols;
        const float mu = mean[ok ? row : 0], rs = rstd[ok ? row : 0];
        f32x4 xh[NCH], g[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f}, xv = f32x4{mu, mu, mu, mu};
            if (cv[j]) {
                xv = reinterpret_cast<const f32x4*>(X + base)[sub + 16 * j];
                if (ok) d = reinterpret_cast<const f32x4*>(dY + base)[sub + 16 * j];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[j][e] = (xv[e] - mu) * rs;
                g[j][e] = d[e] * gam[j][e];
                s1 += g[j][e];
                s2 = fmaf(g[j][e], xh[j][e], s2);
                dg[j][e] = fmaf(d[e], xh[j][e], dg[j][e]);
                db[j][e] += d[e];
            }
        }
        s1 = group16_sum(s1) * inv_n;
        s2 = group16_sum(s2) * inv_n;
        if (ok) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                if (!cv[j]) continue;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[j][e] - s1 - xh[j][e] * s2);
                if (resid) {
                    const f32x4 rr = reinterpret_cast<const f32x4*>(resid + base)[sub + 16 * j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += rr[e];
                }
                reinterpret_cast<f32x4*>(dX + base)[sub + 16 * j] = o;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (!cv[j]) continue;
        reinterpret_cast<f32x4*>(sh + (rg * 2 + 0) * cols)[sub + 16 * j] = dg[j];
        reinterpret_cast<f32x4*>(sh + (rg * 2 + 1) * cols)[sub + 16 * j] = db[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * cols; c += 256) {          // fixed order over the 16 row groups
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += sh[q * 2 * cols + c];
        part[(long)blockIdx.x * 2 * cols + c] = s;
    }
}

// One launch for the column reductions of many LayerNorm backwards.  A job = four 64-bit words: the partials
// [nblk][2 cols], dgamma, dbeta, (nblk << 32) | cols.  blockIdx.y = job, blockIdx.x as in reduce_slabs_kernel<16, 1>.
__global__ __launch_bounds__(1024) void ln_finish_many_kernel(const int64_t* __restrict__ jobs) {
    __shared__ __attribute__((aligned(16))) f32x4 sh[16][64];
    const int64_t* j = jobs + (long)blockIdx.y * VSOM_LN_JOB_WORDS;
    const float* part = reinterpret_cast<const float*>(j[0]);
    float* dgamma = reinterpret_cast<float*>(j[1]);
    float* dbeta = reinterpret_cast<float*>(j[2]);
    const int nblk = (int)(j[3] >> 32), cols = (int)(j[3] & 0xffffffff);
    const int nb1 = (cols + 63) / 64;
    if ((int)blockIdx.x >= 2 * nb1) return;                    // a job narrower than the widest of the launch
    const int vec = ((reinterpret_cast<uintptr_t>(part) & 15u) == 0) && ((2L * cols) % 4 == 0) && (cols % 4 == 0);
    reduce_slabs_body<16, 1>(sh, (int)blockIdx.x, part, 2L * cols, nblk, dgamma, cols, dbeta, cols, cols, nb1, vec);
}

static int ln_bwd_blocks(int rows) {
    // one partial [2][cols] per workgroup goes through the slab reducer afterwards (a handful of
    // workgroups walking all partials): 512 workgroups of 4 x 16 rows keep ~9 KB of loads in flight per
    // wave and leave the reducer a quarter of the partials 2048 single-pass workgroups would
    int b = cdiv(rows, 16);
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return b;
}

}  // namespace vsom

using namespace vsom;

extern "C" {

int vsom_layernorm_fwd(const float* X, const float* gamma, const float* beta, float* Y, float* mean, float* rstd,
                       int rows, int cols, float eps, vsom_stream_t stream) {
    VSOM_REQUIRE(X && gamma && beta && Y && mean && rstd, VSOM_EINVAL, "layernorm_fwd: null pointer");
    VSOM_REQUIRE(rows > 0 && cols > 0, VSOM_EINVAL, "layernorm_fwd: bad shape");
    VSOM_REQUIRE(cols <= 1024, VSOM_EUNSUPPORTED, "layernorm_fwd: cols=%d > 1024", cols);
    dim3 grid(cdiv(rows, 4)), block(256);
    if (cols % 4 == 0 && cols <= 256 && aligned16(X) && aligned16(Y) && aligned16(gamma) && aligned16(beta)) {
        dim3 g16(cdiv(rows, 16));
        switch ((cols + 63) / 64) {
            case 1: VSOM_LAUNCH(layernorm_fwd_v4_kernel<1>, g16, block, 0, stream, X, gamma, beta, Y, mean, rstd, rows, cols, eps); break;
            case 2: VSOM_LAUNCH(layernorm_fwd_v4_kernel<2>, g16, block, 0, stream, X, gamma, beta, Y, mean, rstd, rows, cols, eps); break;
            case 3: VSOM_LAUNCH(layernorm_fwd_v4_kernel<3>, g16, block, 0, stream, X, gamma, beta, Y, mean, rstd, rows, cols, eps); break;
            default: VSOM_LAUNCH(layernorm_fwd_v4_kernel<4>, g16, block, 0, stream, X, gamma, beta, Y, mean, rstd, rows, cols, eps); break;
        }
        VSOM_LAUNCH_CHECK("layernorm_fwd_v4_kernel");
    }
    if (cols <= 256)
        VSOM_LAUNCH(layernorm_fwd_kernel<4>, grid, block, 0, stream, X, gamma, beta, Y, mean, rstd, rows, cols, eps);
    else
        VSOM_LAUNCH(layernorm_fwd_kernel<16>, grid, block, 0, stream, X, gamma, beta, Y, mean, rstd, rows, cols, eps);
    VSOM_LAUNCH_CHECK("layernorm_fwd_kernel");
}

size_t vsom_layernorm_bwd_workspace_bytes(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)ln_bwd_blocks(rows) * 2 * (size_t)cols * sizeof(float);
}

static int ln_bwd_launch(const float* dY, const float* X, const float* mean, const float* rstd, const float* gamma,
                         const float* resid, float* dX, int rows, int cols, float* part, hipStream_t stream) {
    const int nblk = ln_bwd_blocks(rows);
    const size_t shmem = (size_t)8 * cols * sizeof(float);
    const bool v4 = cols % 4 == 0 && cols <= 256 && aligned16(dY) && aligned16(X) && aligned16(gamma) && aligned16(dX) &&
                    (!resid || aligned16(resid));
    if (v4) {
        switch ((cols + 63) / 64) {
            case 1: VSOM_LAUNCH(layernorm_bwd_v4_kernel<1>, dim3(nblk), dim3(256), 0, stream, dY, X, mean, rstd, gamma, resid, dX, part, rows, cols); break;
            case 2: VSOM_LAUNCH(layernorm_bwd_v4_kernel<2>, dim3(nblk), dim3(256), 0, stream, dY, X, mean, rstd, gamma, resid, dX, part, rows, cols); break;
            case 3: VSOM_LAUNCH(layernorm_bwd_v4_kernel<3>, dim3(nblk), dim3(256), 0, stream, dY, X, mean, rstd, gamma, resid, dX, part, rows, cols); break;
            default: VSOM_LAUNCH(layernorm_bwd_v4_kernel<4>, dim3(nblk), dim3(256), 0, stream, dY, X, mean, rstd, gamma, resid, dX, part, rows, cols); break;
        }
    } else if (cols <= 256)
        VSOM_LAUNCH(layernorm_bwd_kernel<4>, dim3(nblk), dim3(256), shmem, stream, dY, X, mean, rstd, gamma, resid, dX, part, rows, cols);
    else
        VSOM_LAUNCH(layernorm_bwd_kernel<16>, dim3(nblk), dim3(256), shmem, stream, dY, X, mean, rstd, gamma, resid, dX, part, rows, cols);
    return hip_status(hipGetLastError(), "layernorm_bwd_kernel");
}

int vsom_layernorm_bwd(const float* dY, const float* X, const float* mean, const float* rstd, const float* gamma,
                       const float* resid, float* dX, float* dgamma, float* dbeta, int rows, int cols, void* ws,
                       size_t ws_bytes, vsom_stream_t stream) {
    VSOM_REQUIRE(dY && X && mean && rstd && gamma && dX && dgamma && dbeta, VSOM_EINVAL, "layernorm_bwd: null pointer");
    VSOM_REQUIRE(rows > 0 && cols > 0, VSOM_EINVAL, "layernorm_bwd: bad shape");
    VSOM_REQUIRE(cols <= 1024, VSOM_EUNSUPPORTED, "layernorm_bwd: cols=%d > 1024", cols);
    VSOM_REQUIRE(ws && ws_bytes >= vsom_layernorm_bwd_workspace_bytes(rows, cols), VSOM_EWORKSPACE, "layernorm_bwd: workspace too small");
    VSOM_REQUIRE(aligned16(ws), VSOM_EALIGN, "layernorm_bwd: workspace must be 16-byte aligned");
    float* part = static_cast<float*>(ws);
    const int rc = ln_bwd_launch(dY, X, mean, rstd, gamma, resid, dX, rows, cols, part, stream);
    if (rc) return rc;
    return reduce_slabs2_internal(part, 2L * cols, ln_bwd_blocks(rows), dgamma, cols, dbeta, cols, cols, stream);
}

/* The same in two halves, for a caller that runs many LayerNorm backwards before anybody needs their dgamma / dbeta:
   _partial does everything but the column reduction (its partials stay in `part`, which the caller keeps), and ONE
   vsom_layernorm_bwd_finish_many launch reduces the partials of `count` such calls -- same kernel body, same order,
   same bits as vsom_layernorm_bwd (30 six-workgroup launches on the step's critical chain become one). */
int vsom_layernorm_bwd_deferrable(int rows, int cols) {
    // the shapes whose reduction vsom_layernorm_bwd runs in the one-column-per-lane form (reduce_slabs2_internal)
    return rows > 0 && cols > 0 && cols <= 1024 && 2L * cols <= 4096 && ln_bwd_blocks(rows) >= 32;
}

int vsom_layernorm_bwd_partial(const float* dY, const float* X, const float* mean, const float* rstd, const float* gamma,
                               const float* resid, float* dX, int rows, int cols, void* part, size_t part_bytes,
                               vsom_stream_t stream) {
    VSOM_REQUIRE(dY && X && mean && rstd && gamma && dX, VSOM_EINVAL, "layernorm_bwd_partial: null pointer");
    VSOM_REQUIRE(vsom_layernorm_bwd_deferrable(rows, cols), VSOM_EUNSUPPORTED, "layernorm_bwd_partial: shape rows=%d cols=%d", rows, cols);
    VSOM_REQUIRE(part && aligned16(part) && part_bytes >= vsom_layernorm_bwd_workspace_bytes(rows, cols), VSOM_EWORKSPACE,
                 "layernorm_bwd_partial: partial buffer too small or misaligned");
    return ln_bwd_launch(dY, X, mean, rstd, gamma, resid, dX, rows, cols, static_cast<float*>(part), stream);
}

int vsom_layernorm_bwd_finish_many(const int64_t* jobs_dev, int first, int count, int max_cols, vsom_stream_t stream) {
    VSOM_REQUIRE(jobs_dev && first >= 0 && count >= 0 && max_cols > 0 && 2L * max_cols <= 4096, VSOM_EINVAL, "layernorm_bwd_finish_many: bad arguments");
    if (count == 0) return VSOM_OK;
    VSOM_LAUNCH(ln_finish_many_kernel, dim3(2 * cdiv(max_cols, 64), count), dim3(1024), 0, stream, jobs_dev + (long)first * VSOM_LN_JOB_WORDS);
    VSOM_LAUNCH_CHECK("ln_finish_many_kernel");
}

}  // extern "C"
