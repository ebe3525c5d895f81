// BMU distance pass for the cosine SOM (models/som_layer.py:119-122, 83-89):
//     dist[i,k] = 1 - <x_i, w_k> / (max(|x_i|, eps) max(|w_k|, eps)),   bmu[i] = first argmin_k dist[i,k]
// as a reduced-precision contraction + exact re-rank (the only way off the f32-MFMA roofline).
//
//  1. bmu_x3_kernel: X W^T on the bf16 matrix cores from a TWO-piece round-to-nearest split of each fp32
//     operand, a = a1 + a2 + r2 with |a2| <= 2^-9 |a|, |r2| <= 2^-17 |a|, three products a2 b1 + a1 b2 + a1 b1
//     (fp32 accumulate).  Dropped: a1 s2 + r2 b1 + r1 s1, each <= 2^-16 |a||b| (r1 = a - a1, s* likewise), so
//     |error of the normalised dot| <= 3 * 2^-16 = 4.6e-5 in the worst case (Cauchy-Schwarz; observed ~1e-7:
//     the terms carry random signs) -- half the matrix-core work of the six-product engine of gemm_x6.h and a
//     sixteenth of the f32 MFMA's.  The squared row norms of X and W ride along (the tiles pass through the
//     registers anyway), so the two row-norm passes over X and W disappear.  Reduction over L split across
//     workgroups, partial dots / norms in fp32 slabs summed in fixed order (bitwise reproducible).
//  2. bmu_norms_kernel: inv_nx, inv_nw from the norm partials.
//  3. bmu_x3_finalize_kernel (one workgroup per sample): distances, approximate minimum, then every prototype
//     within BMU_WINDOW of it is RE-RANKED with an exact dot product (fp32 products accumulated in fp64 over the
//     whole row); the exact distances replace the approximate ones in dist and the BMU is their first minimum.
//     Since BMU_WINDOW > 2 x the contraction's error bound, the true minimum is always among the candidates and
//     every prototype outside keeps a value above the winner's: bmu == argmin(dist) holds exactly, and bmu is
//     the argmin of distances that are exact to fp64 rounding wherever it matters.  The candidates get their slots
//     from a block-wide prefix sum (deterministic, column order) and are re-ranked 256 at a time, however many there
//     are (all K of them for a dead input or collapsed prototypes: exact ties then resolve to the lowest index).
#include "gemm_x6.h"
#include "adamw.h"

#include <type_traits>

namespace vsom {

constexpr float BMU_WINDOW = 2.0e-4f;      // 2 x (3 * 2^-16 = 4.6e-5, the split's worst case) = 9.2e-5, + as much again for the fp32
                                           // accumulation over L <= 49152 terms and the slab sums

struct BmuP {
    const float* X; long ldx; const float* W;
    int B, K, L;
    int ktiles_per_split;
    float* slab; long slab_stride;      // [splits][B*K]
    float* xsq; float* wsq;             // [splits][B], [splits][K] partial squared norms
    unsigned x_bytes, w_bytes;
};

// k-contiguous fp32 tile ROWS x 32 staged by NT threads: thread t loads float4 (row = p * (NT / 8) + t / 8, k = (t % 8) * 4)
template <int ROWS, int NT> struct X3Stage { f32x4 v[ROWS / (NT / 8)]; };
template <int ROWS, int NT> struct X3Off { unsigned off[ROWS / (NT / 8)]; };
template <int ROWS, int NT>
__device__ __forceinline__ void x3_init(X3Off<ROWS, NT>& o, long ld, int row0, int nrows, int t) {
#pragma unroll
    for (int p = 0; p < ROWS / (NT / 8); ++p) {
        const int row = row0 + p * (NT / 8) + (t >> 3);
        o.off[p] = (row < nrows) ? (unsigned)(((long)row * ld + ((t & 7) << 2)) << 2) : OOB;
    }
}
template <int ROWS, int NT>
__device__ __forceinline__ void x3_load(X3Stage<ROWS, NT>& s, __amdgpu_buffer_rsrc_t rsrc, const X3Off<ROWS, NT>& o, int k0, int K, int t) {
    const bool kok = k0 + ((t & 7) << 2) < K;
    const unsigned kbytes = (unsigned)k0 << 2;
#pragma unroll
    for (int p = 0; p < ROWS / (NT / 8); ++p) s.v[p] = bload4(rsrc, (kok && o.off[p] != OOB) ? o.off[p] + kbytes : OOB);
}

// tile (WAVES_M WM 32) x (WAVES_N WN 32) x 32; LDS: 2 planes x (BM + BN) rows x 64 B (swizzled image of gemm_x6.h).  The split costs ~18 VALU per
// float4 against 3 (not 6) MFMAs per 16-deep step, so the tile has to be LARGE to keep the loop off the VALU issue
// limit: 7.5 VALU per MFMA at 128 x 128 (measured 89 us, issue-bound), 3.5 at 256 x 192.
template <int WM, int WN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void bmu_x3_kernel(const BmuP g) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, NT = WAVES_M * WAVES_N * 64, RPP = NT / 8;
    constexpr int PA = BM * X6_RS, PB = BN * X6_RS;
    __shared__ __attribute__((aligned(16))) char lds[2 * (PA + PB)];
    char* As = lds; char* Bs = lds + 2 * PA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WAVES_N) * (WM * 32), wn0 = (wave % WAVES_N) * (WN * 32);
    const int tiles_n = (g.K + BN - 1) / BN, tiles_m = (g.B + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);          // split-major: an XCD owns a slice of L
    const int z = lid / ntiles, rem = lid - z * ntiles;
    const int tm = rem % tiles_m, tn = rem / tiles_m;          // neighbours share the (larger) W panel
    const int bm0 = tm * BM, bn0 = tn * BN;
    const int ktiles = (g.L + 31) >> 5;
    const int kt_begin = z * g.ktiles_per_split;
    int kt_end = kt_begin + g.ktiles_per_split;
    if (kt_end > ktiles) kt_end = ktiles;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X), 0, (int)g.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W), 0, (int)g.w_bytes, 0x00020000);
    // (a second staging register set -- two k-tiles of loads in flight -- measured slower: 95 vs 87 us at 128 x 128)
    X3Stage<BM, NT> sa0; X3Stage<BN, NT> sb0;
    X3Off<BM, NT> oa; X3Off<BN, NT> ob;
    x3_init<BM, NT>(oa, g.ldx, bm0, g.B, t);
    x3_init<BN, NT>(ob, g.L, bn0, g.K, t);
    const bool want_x = tn == 0, want_w = tm == 0;            // squared-norm partials: one column / row of tiles
    float ssa[BM / RPP], ssb[BN / RPP];
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) ssa[p] = 0.f;
#pragma unroll
    for (int p = 0; p < BN / RPP; ++p) ssb[p] = 0.f;

    auto gload = [&](X3Stage<BM, NT>& sa, X3Stage<BN, NT>& sb, int kt) {       // kt beyond the range: k >= L -> zeros (never stored)
        x3_load<BM, NT>(sa, rsA, oa, kt << 5, g.L, t);
        x3_load<BN, NT>(sb, rsB, ob, kt << 5, g.L, t);
    };
    auto lstore = [&](const X3Stage<BM, NT>& sa, const X3Stage<BN, NT>& sb) {
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) {
            const f32x4 v = sa.v[p];
            if (want_x) ssa[p] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            uint2 p1, p2;
            x3_split(v, p1, p2);
            char* dst = As + x6_piece_off(p * RPP + (t >> 3), t & 7);
            *reinterpret_cast<uint2*>(dst) = p1;
            *reinterpret_cast<uint2*>(dst + PA) = p2;
        }
#pragma unroll
        for (int p = 0; p < BN / RPP; ++p) {
            const f32x4 v = sb.v[p];
            if (want_w) ssb[p] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            uint2 p1, p2;
            x3_split(v, p1, p2);
            char* dst = Bs + x6_piece_off(p * RPP + (t >> 3), t & 7);
            *reinterpret_cast<uint2*>(dst) = p1;
            *reinterpret_cast<uint2*>(dst + PB) = p2;
        }
    };
    auto mfma_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[WM][2], b[WN][2];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    a[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PA + x6_chunk_off(wm0 + i * 32 + r, 2 * ks + h));
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    b[j][pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * PB + x6_chunk_off(wn0 + j * 32 + r, 2 * ks + h));
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);   // 2^-9 terms
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);   // leading term
                    acc[i][j] = c;
                }
        }
    };
    if (kt_begin < kt_end) {
        gload(sa0, sb0, kt_begin);
        lstore(sa0, sb0);
    }
    __syncthreads();
    for (int kt = kt_begin; kt + 1 < kt_end; ++kt) {       // branch-free body, last k-tile peeled (gemm_x6.h)
        gload(sa0, sb0, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        lstore(sa0, sb0);
        __syncthreads();
    }
    if (kt_begin < kt_end) mfma_tile();

    // squared-norm partials: the 8 threads of a row are 8 consecutive lanes
    if (want_x) {
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) {
            float v = ssa[p];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
            const int m = bm0 + p * RPP + (t >> 3);
            if ((t & 7) == 0 && m < g.B) g.xsq[(long)z * g.B + m] = v;
        }
    }
    if (want_w) {
#pragma unroll
        for (int p = 0; p < BN / RPP; ++p) {
            float v = ssb[p];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
            const int n = bn0 + p * RPP + (t >> 3);
            if ((t & 7) == 0 && n < g.K) g.wsq[(long)z * g.K + n] = v;
        }
    }
    // slab[z][m * K + n]; accumulator register v: row (v & 3) + 8 (v >> 2) + 4 h, column r
    float* sl = g.slab + (long)z * g.slab_stride;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = bn0 + wn0 + j * 32 + r;
            if (n >= g.K) continue;
            const int mb = bm0 + wm0 + i * 32 + 4 * h;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = mb + (v & 3) + 8 * (v >> 2);
                if (m < g.B) sl[(long)m * g.K + n] = acc[i][j][v];
            }
        }
}

// inv[i] = 1 / max(sqrt(sum_z part[z][i]), eps)   (F.normalize's eps = 1e-12, som_layer.py:120-121)
__global__ __launch_bounds__(256) void bmu_norms_kernel(const float* __restrict__ xsq, const float* __restrict__ wsq, int nz,
                                                        int B, int K, float* __restrict__ inv_nx, float* __restrict__ inv_nw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B + K) return;
    const bool isx = i < B;
    const float* p = isx ? xsq + i : wsq + (i - B);
    const int n = isx ? B : K;
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += p[(long)z * n];
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    if (isx) inv_nx[i] = inv; else inv_nw[i - B] = inv;
}

// The same for MANY partials per row (the plane images carry one per 64 elements of a row: 192 at L = 12288): 64 rows per
// workgroup, wave w sums the partials z = w, w + 8, ... (all loads independent), the eight sums are added in wave order.
__global__ __launch_bounds__(512) void bmu_norms_wide_kernel(const float* __restrict__ xsq, const float* __restrict__ wsq, int nz,
                                                             int B, int K, float* __restrict__ inv_nx, float* __restrict__ inv_nw) {
    __shared__ float part[8][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const bool live = i < B + K, isx = i < B;
    const float* p = isx ? xsq + i : wsq + (i - B);
    const int n = isx ? B : K;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (live) {
        int z = w;
        for (; z + 24 < nz; z += 32) {
            s0 += p[(long)z * n]; s1 += p[(long)(z + 8) * n]; s2 += p[(long)(z + 16) * n]; s3 += p[(long)(z + 24) * n];
        }
        for (; z < nz; z += 8) s0 += p[(long)z * n];
    }
    part[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && live) {
        float s = part[0][lane];
#pragma unroll
        for (int q = 1; q < 8; ++q) s += part[q][lane];
        const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
        if (isx) inv_nx[i] = inv; else inv_nw[i - B] = inv;
    }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One workgroup per sample row: distances, approximate first minimum, exact re-rank of the candidates.
constexpr int BMU_KPT = 8;           // prototypes per thread held in registers (K <= 2048)
// V4: K % 4 == 0 and 16-byte aligned slabs / dist -- a thread owns two quads of consecutive columns (16-byte accesses,
// two slabs of loads in flight) instead of eight columns 256 apart (4-byte accesses behind a bounds branch each)
template <bool V4>
__global__ __launch_bounds__(256) void bmu_x3_finalize_kernel(const float* __restrict__ slab, long slab_stride, int nslabs,
                                                              const float* __restrict__ X, long ldx, const float* __restrict__ W,
                                                              const float* __restrict__ inv_nx, const float* __restrict__ inv_nw,
                                                              float* __restrict__ dist, int64_t* __restrict__ bmu, int K, int L,
                                                              int* __restrict__ rerank_count) {
    __shared__ float sb[4];
    __shared__ int si[4];
    __shared__ int cand[256];
    __shared__ int wtot[4];
    __shared__ double sd[4];
    const int i = blockIdx.x, t = threadIdx.x;
    const float rx = inv_nx[i];
    static_assert(BMU_KPT == 8, "two quads per thread");
    auto kcol = [&](int u) { return V4 ? 4 * t + (u & 3) + 1024 * (u >> 2) : t + 256 * u; };
    float d[BMU_KPT];
    float best = INFINITY;
    int bidx = 0x7fffffff;
    {
        float dot[BMU_KPT];
#pragma unroll
        for (int u = 0; u < BMU_KPT; ++u) dot[u] = 0.f;
        if constexpr (V4) {
            const float* p = slab + (long)i * K + 4 * t;
            const bool q0 = 4 * t < K, q1 = 4 * t + 1024 < K;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            int s = 0;
            for (; s + 1 < nslabs; s += 2) {                   // fixed order s = 0, 1, ... per (i, k); two slabs in flight
                const f32x4 a0 = q0 ? *reinterpret_cast<const f32x4*>(p) : z, a1 = q1 ? *reinterpret_cast<const f32x4*>(p + 1024) : z;
                const f32x4 b0 = q0 ? *reinterpret_cast<const f32x4*>(p + slab_stride) : z;
                const f32x4 b1 = q1 ? *reinterpret_cast<const f32x4*>(p + slab_stride + 1024) : z;
#pragma unroll
                for (int e = 0; e < 4; ++e) { dot[e] += a0[e]; dot[4 + e] += a1[e]; }
#pragma unroll
                for (int e = 0; e < 4; ++e) { dot[e] += b0[e]; dot[4 + e] += b1[e]; }
                p += 2 * slab_stride;
            }
            if (s < nslabs) {
                const f32x4 a0 = q0 ? *reinterpret_cast<const f32x4*>(p) : z, a1 = q1 ? *reinterpret_cast<const f32x4*>(p + 1024) : z;
#pragma unroll
                for (int e = 0; e < 4; ++e) { dot[e] += a0[e]; dot[4 + e] += a1[e]; }
            }
        } else {
            const float* p = slab + (long)i * K + t;
            for (int s = 0; s < nslabs; ++s) {                 // fixed order s = 0, 1, ... per (i, k)
#pragma unroll
                for (int u = 0; u < BMU_KPT; ++u)
                    if (t + 256 * u < K) dot[u] += p[256 * u];
                p += slab_stride;
            }
        }
#pragma unroll
        for (int u = 0; u < BMU_KPT; ++u) {
            const int k = kcol(u);
            d[u] = INFINITY;
            if (k >= K) continue;
            d[u] = 1.0f - dot[u] * rx * inv_nw[k];
            if (d[u] < best || (d[u] == best && k < bidx)) { best = d[u]; bidx = k; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if ((t & 63) == 0) { sb[t >> 6] = best; si[t >> 6] = bidx; }
    __syncthreads();
    best = sb[0]; bidx = si[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (sb[w] < best || (sb[w] == best && si[w] < bidx)) { best = sb[w]; bidx = si[w]; }
    // candidates: everything within the window of the approximate minimum (NaN never qualifies).  Slot of a candidate =
    // exclusive prefix sum of the per-thread counts (wave scan + the waves before it): deterministic.
    const float lim = best + BMU_WINDOW;
    int mine = 0;
#pragma unroll
    for (int u = 0; u < BMU_KPT; ++u) mine += (kcol(u) < K && d[u] <= lim) ? 1 : 0;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if ((t & 63) >= o) incl += v;
    }
    if ((t & 63) == 63) wtot[t >> 6] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < (t >> 6); ++w) base += wtot[w];
    const int nc = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (nc > 1) {
        // exact dots: fp32 products accumulated in fp64; candidates 256 at a time
        const f32x4* xr = reinterpret_cast<const f32x4*>(X + (long)i * ldx);
        const int n4 = L >> 2;
        float ebest = INFINITY;
        int eidx = 0x7fffffff;
        for (int c0 = 0; c0 < nc; c0 += 256) {
            __syncthreads();                       // cand free again
            int slot = base;
#pragma unroll
            for (int u = 0; u < BMU_KPT; ++u) {
                const int k = kcol(u);
                if (k < K && d[u] <= lim) {
                    if (slot >= c0 && slot < c0 + 256) cand[slot - c0] = k;
                    ++slot;
                }
            }
            __syncthreads();
            const int ncc = nc - c0 < 256 ? nc - c0 : 256;
            for (int c = 0; c < ncc; ++c) {
                const int k = cand[c];
                const f32x4* wr = reinterpret_cast<const f32x4*>(W + (long)k * L);
                double s0 = 0.0, s1 = 0.0;
                int j = t;
                for (; j + 256 < n4; j += 512) {
                    const f32x4 xa = xr[j], wa = wr[j], xb = xr[j + 256], wb = wr[j + 256];
                    s0 += (double)xa[0] * wa[0] + (double)xa[1] * wa[1] + (double)xa[2] * wa[2] + (double)xa[3] * wa[3];
                    s1 += (double)xb[0] * wb[0] + (double)xb[1] * wb[1] + (double)xb[2] * wb[2] + (double)xb[3] * wb[3];
                }
                for (; j < n4; j += 256) {
                    const f32x4 xa = xr[j], wa = wr[j];
                    s0 += (double)xa[0] * wa[0] + (double)xa[1] * wa[1] + (double)xa[2] * wa[2] + (double)xa[3] * wa[3];
                }
                for (int e = (n4 << 2) + t; e < L; e += 256) s0 += (double)X[(long)i * ldx + e] * W[(long)k * L + e];
                const double ws = wave_sum_f64(s0 + s1);
                __syncthreads();                       // sd free again
                if ((t & 63) == 0) sd[t >> 6] = ws;
                __syncthreads();
                const double dotx = (sd[0] + sd[1]) + (sd[2] + sd[3]);
                const float de = (float)(1.0 - dotx * (double)rx * (double)inv_nw[k]);
                if (t == 0 && dist) dist[(long)i * K + k] = de;
                if (de < ebest || (de == ebest && k < eidx)) { ebest = de; eidx = k; }
            }
        }
        bidx = eidx;
        if (t == 0 && rerank_count) atomicAdd(rerank_count, 1);
    }
    // the approximate distances of everything that was not re-ranked
    if (dist) {
#pragma unroll
        for (int u = 0; u < BMU_KPT; ++u) {
            const int k = kcol(u);
            if (k < K && !(nc > 1 && d[u] <= lim)) dist[(long)i * K + k] = d[u];
        }
    }
    if (t == 0) bmu[i] = (bidx == 0x7fffffff) ? 0 : (int64_t)bidx;
}

// ================================================================================ pre-split plane images
// The contraction above spends as many issue slots on the split (global -> registers -> two bf16 planes -> LDS) as on the
// matrix cores, every workgroup re-splitting the same operand tiles.  The "planes" form takes the split OUT of the loop: an
// operand [R, L] is split once into a FRAGMENT IMAGE -- for every 16-deep k step s and every block rb of 32 rows, the two
// planes as 1 KB MFMA fragments in lane order (lane l = (r = l & 31, h = l >> 5) holds row 32 rb + r, k = 16 s + 8 h .. + 8):
//     image[((s * nrb + rb) * 2 + plane) * 1 KB + 16 l],   rows >= R and k >= L zero,
// followed by the squared-norm partials of the rows, [ceil(L / 64)][R] floats (fixed summation order).  The prototypes'
// image is written by the optimizer step that updates them (vsom_adamw_step_planes: +4 B per element on a pass that moves
// 28), the samples' by planes_kernel<false> (or any producer).  The contraction is then
//     buffer_load_dwordx4 ... lds (1 KB pieces, no VGPRs, no VALU) -> ds_read_b128 (lane-linear, conflict-free) -> MFMA
// over a ring of four 16-deep stages (28 KB each), DMA three stages ahead with counted vmcnt, ONE barrier per stage, the
// fragments of stage s + 1 read under the MFMAs of stage s (register double buffer), everything that is not an MFMA issued
// between the three MFMA groups of a stage.  Same products, same order, same reduction split as bmu_x3_kernel<2,3,4,2>:
// the slabs are bit-identical (lab/bmu_planes_lab.hip checks that), so distances and BMUs are too, up to the last bit of
// the row norms (their partial sums are cut at other places).
constexpr int PL_CHUNK = 64;                       // k per workgroup of the image writer (4 stages)

struct PlanesP {
    const float* src; long ld;                     // fp32 operand (writer) ...
    float* p; const float* g; float* m; float* v;  // ... or the AdamW state of the slice (p is then the operand, ld = L)
    const float* wd_chunk; long off;               // weight decay per 256 elements of the arena, slice offset in it
    int R, L, nrb, nst;
    uint4* img; float* sq;                         // image, squared-norm partials [chunk][R]
    AdamwC c;
    int nblk_planes; long flat_lo4, flat_gap4, flat_n4;   // AdamW form: blocks past nblk_planes update the rest of the arena --
                                                           // its float4s [0, flat_lo4) and [flat_lo4 + flat_gap4, ...), flat_n4 in all
};

// One workgroup = 32 rows x 64 k: thread t = (row r = t >> 3, k = 8 (t & 7) .. + 8) reads 32 contiguous bytes (a row of the
// block = 256 contiguous bytes), optionally applies the AdamW update, splits, and the block leaves through LDS as eight
// whole 1 KB fragments.
template <bool ADAMW>
__global__ __launch_bounds__(256) void planes_kernel(const PlanesP q) {
    __shared__ uint4 frag[8][64];
    if constexpr (ADAMW) {
        if ((int)blockIdx.x >= q.nblk_planes) {            // the arena outside the slice: adamw_kernel's loop (misc.hip)
            const long nb = gridDim.x - q.nblk_planes;
            for (long j = ((long)blockIdx.x - q.nblk_planes) * 256 + threadIdx.x; j < q.flat_n4; j += nb * 256) {
                const long i = j < q.flat_lo4 ? j : j + q.flat_gap4;
                const float wd = q.wd_chunk[i >> 6];
                f32x4 pp = reinterpret_cast<f32x4*>(q.p)[i];
                const f32x4 gg = reinterpret_cast<const f32x4*>(q.g)[i];
                f32x4 mm = reinterpret_cast<f32x4*>(q.m)[i];
                f32x4 vv = reinterpret_cast<f32x4*>(q.v)[i];
                adamw_update(pp, gg, mm, vv, wd, q.c);
                reinterpret_cast<f32x4*>(q.p)[i] = pp;
                reinterpret_cast<f32x4*>(q.m)[i] = mm;
                reinterpret_cast<f32x4*>(q.v)[i] = vv;
            }
            return;
        }
    }
    const int t = threadIdx.x, r = t >> 3, kq = t & 7;
    const int rb = blockIdx.x % q.nrb, chunk = blockIdx.x / q.nrb;
    const int row = rb * 32 + r, k = chunk * PL_CHUNK + kq * 8;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (row < q.R && k < q.L) {                    // L % 8 == 0: the 8 elements are inside the row together
        if constexpr (ADAMW) {
            const long e = q.off + (long)row * q.L + k;
            const float wd = q.wd_chunk[e >> 8];
            f32x4* pp = reinterpret_cast<f32x4*>(q.p + e);
            const f32x4* gp = reinterpret_cast<const f32x4*>(q.g + e);
            f32x4* mp = reinterpret_cast<f32x4*>(q.m + e);
            f32x4* vp = reinterpret_cast<f32x4*>(q.v + e);
            f32x4 m0 = mp[0], m1 = mp[1], w0 = vp[0], w1 = vp[1];
            v0 = pp[0]; v1 = pp[1];
            adamw_update(v0, gp[0], m0, w0, wd, q.c);
            adamw_update(v1, gp[1], m1, w1, wd, q.c);
            pp[0] = v0; pp[1] = v1; mp[0] = m0; mp[1] = m1; vp[0] = w0; vp[1] = w1;
        } else {
            const f32x4* sp = reinterpret_cast<const f32x4*>(q.src + (long)row * q.ld + k);
            v0 = sp[0]; v1 = sp[1];
        }
    }
    float ss = ((v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3])) + ((v1[0] * v1[0] + v1[1] * v1[1]) + (v1[2] * v1[2] + v1[3] * v1[3]));
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
    if (kq == 0 && row < q.R) q.sq[(long)chunk * q.R + row] = ss;
    uint2 a1, a2, b1, b2;
    x3_split(v0, a1, a2);
    x3_split(v1, b1, b2);
    const int sl = kq >> 1, lane = ((kq & 1) << 5) | r;
    frag[sl * 2 + 0][lane] = uint4{a1.x, a1.y, b1.x, b1.y};
    frag[sl * 2 + 1][lane] = uint4{a2.x, a2.y, b2.x, b2.y};
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int f = pass * 4 + (t >> 6), stage = chunk * 4 + (f >> 1);
        if (stage < q.nst) q.img[(((long)stage * q.nrb + rb) * 2 + (f & 1)) * 64 + (t & 63)] = frag[f][t & 63];
    }
}

struct BmuPlP {
    const void* ximg; const void* wimg;
    unsigned ximg_bytes, wimg_bytes;
    int nrb_x, nrb_w;                   // 32-row blocks of X and W
    int B, K, nst;                      // nst = 16-deep stages in total
    int stages_per_split;
    float* slab; long slab_stride;
};

typedef int pl_i32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor (base, stride 0, num_records = bytes, DATA_FORMAT = 32)
__device__ __forceinline__ pl_i32x4 pl_srd(const void* base, unsigned bytes) {
    const unsigned long a = (unsigned long)base;
    return pl_i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// One 1 KB piece global -> LDS (lane l: 16 B from voff to lds_dst + 16 l); voff beyond the buffer moves nothing.  In assembly
// so that hipcc does not know it writes LDS: with the builtin it orders every later ds_read behind the transfer
// (s_waitcnt vmcnt(0) right after the issue).  The waits are placed by hand.
__device__ __forceinline__ void pl_dma16(pl_i32x4 rs, unsigned lds_dst, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}
__device__ __forceinline__ unsigned pl_lds_addr(const char* p) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)p;
}
// fragment reads in assembly too (left to hipcc they sink to just before their MFMA)
template <int OFF>
__device__ __forceinline__ void pl_frag_read(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
struct PlFrags { bf16x8 a[2][2], b[3][2]; };         // wave tile 64 x 96: 2 + 3 row blocks, two planes each
__device__ __forceinline__ void pl_frags_issue(PlFrags& f, unsigned aaddr, unsigned baddr) {
    pl_frag_read<0>(f.a[0][0], aaddr); pl_frag_read<1024>(f.a[0][1], aaddr);
    pl_frag_read<2048>(f.a[1][0], aaddr); pl_frag_read<3072>(f.a[1][1], aaddr);
    pl_frag_read<0>(f.b[0][0], baddr); pl_frag_read<1024>(f.b[0][1], baddr);
    pl_frag_read<2048>(f.b[1][0], baddr); pl_frag_read<3072>(f.b[1][1], baddr);
    pl_frag_read<4096>(f.b[2][0], baddr); pl_frag_read<5120>(f.b[2][1], baddr);
}
// all fragment reads of this wave have returned; the registers pass through the statement so that no MFMA moves above it
__device__ __forceinline__ void pl_frags_wait(PlFrags& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]),
                 "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.b[2][0]), "+v"(f.b[2][1]));
}

constexpr int PL_NST = 4, PL_AHEAD = PL_NST - 1;                // ring slots, stages in flight towards LDS
constexpr int PL_RA = 8, PL_RB = 6;                             // 32-row blocks per tile: 256 x 192
constexpr int PL_STAGE = (PL_RA + PL_RB) * 2048, PL_PIECES = (PL_RA + PL_RB) * 2, PL_PPW = (PL_PIECES + 7) / 8;
constexpr int PL_LDS = PL_NST * PL_STAGE + 1024;                // + 1 KB that absorbs the filler pieces

__global__ __launch_bounds__(512) void bmu_x3_planes_kernel(const BmuPlP g) {
    constexpr int WM = 2, WN = 3, WAVES_N = 2, NW = 8, BM = PL_RA * 32, BN = PL_RB * 32;
    constexpr unsigned NOWHERE = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) char pl_lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wmi = wave / WAVES_N, wni = wave % WAVES_N;
    const int tiles_n = (g.K + BN - 1) / BN, tiles_m = (g.B + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);          // as bmu_x3_kernel: split-major, neighbours share the W panel
    const int z = lid / ntiles, rem = lid - z * ntiles;
    const int tm = rem % tiles_m, tn = rem / tiles_m;
    const int s_begin = z * g.stages_per_split;
    int s_end = s_begin + g.stages_per_split;
    if (s_end > g.nst) s_end = g.nst;
    const int S = s_end - s_begin;                              // even: a split is a whole number of 32-deep k-tiles

    const pl_i32x4 rsX = pl_srd(g.ximg, g.ximg_bytes), rsW = pl_srd(g.wimg, g.wimg_bytes);
    const unsigned lds0 = pl_lds_addr(pl_lds);
    // this wave's pieces of a stage: q = 8 i + wave; q < 16: X piece q, else W piece q - 16; q >= 28: filler (moves nothing)
    unsigned pv[PL_PPW], pd[PL_PPW], pstep[PL_PPW];
    bool px[PL_PPW];
#pragma unroll
    for (int i = 0; i < PL_PPW; ++i) {
        const int q = i * NW + wave;
        px[i] = q < 2 * PL_RA;
        const int qq = px[i] ? q : q - 2 * PL_RA;
        const int rb0 = px[i] ? tm * PL_RA : tn * PL_RB, nrb = px[i] ? g.nrb_x : g.nrb_w;
        const bool ok = q < PL_PIECES && rb0 + (qq >> 1) < nrb;             // row blocks past the operand's end are not fetched
        pv[i] = ok ? (unsigned)(((long)rb0 * 2 + qq) * 1024 + lane * 16) : NOWHERE;
        pstep[i] = ok ? (unsigned)nrb * 2048u : 0u;
        pd[i] = q < PL_PIECES ? (unsigned)(q * 1024) : (unsigned)(PL_NST * PL_STAGE);
    }
    auto dma_stage = [&](int s_rel) {               // stage s_begin + s_rel -> ring slot s_rel % NST; past the split's end: nothing
        const bool live = s_rel < S;
        const unsigned slot = lds0 + (unsigned)(s_rel % PL_NST) * PL_STAGE;
#pragma unroll
        for (int i = 0; i < PL_PPW; ++i) {
            const unsigned off = (live && pv[i] != NOWHERE) ? pv[i] + (unsigned)(s_begin + s_rel) * pstep[i] : NOWHERE;
            pl_dma16(px[i] ? rsX : rsW, (pd[i] == (unsigned)(PL_NST * PL_STAGE) ? lds0 : slot) + pd[i], off);
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    const unsigned a_in_stage = (unsigned)(wmi * WM * 2048 + lane * 16), b_in_stage = (unsigned)(PL_RA * 2048 + wni * WN * 2048 + lane * 16);
    PlFrags f0, f1;
    // product P of the three (a2 b1, a1 b2, a1 b1 -- smallest first, the order of bmu_x3_kernel) for the six tiles of the wave
    auto mfma_group = [&](const PlFrags& f, auto P) {
        constexpr int PA_ = P.value == 0 ? 1 : 0, PB_ = P.value == 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][PA_], f.b[j][PB_], acc[i][j], 0, 0, 0);
    };
    // stage s: its fragments are on their way into `cur`; stages s + 1 .. s + AHEAD - 1 are in flight towards LDS
    auto step = [&](int s, PlFrags& cur, PlFrags& nxt) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PL_AHEAD - 2) * PL_PPW) : "memory");   // this wave's pieces of stage s + 1 have landed
        __builtin_amdgcn_s_barrier();                                          // everybody's have; nobody reads slot (s - 1) % NST any more
        pl_frags_wait(cur);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(cur, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        const unsigned slot = lds0 + (unsigned)((s + 1) % PL_NST) * PL_STAGE;
        pl_frags_issue(nxt, slot + a_in_stage, slot + b_in_stage);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(cur, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        dma_stage(s + PL_AHEAD);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(cur, std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int a = 0; a < PL_AHEAD; ++a) dma_stage(a);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((PL_AHEAD - 1) * PL_PPW) : "memory");
    __builtin_amdgcn_s_barrier();
    pl_frags_issue(f0, lds0 + a_in_stage, lds0 + b_in_stage);
    for (int s = 0; s < S; s += 2) {
        step(s, f0, f1);
        step(s + 1, f1, f0);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // the ring is quiet before the LDS is given back

    // slab[z][m * K + n]; accumulator register v: row (v & 3) + 8 (v >> 2) + 4 h, column r
    float* sl = g.slab + (long)z * g.slab_stride;
    const int bm0 = tm * BM, bn0 = tn * BN, wm0 = wmi * WM * 32, wn0 = wni * WN * 32;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = bn0 + wn0 + j * 32 + r;
            if (n >= g.K) continue;
            const int mb = bm0 + wm0 + i * 32 + 4 * h;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = mb + (v & 3) + 8 * (v >> 2);
                if (m < g.B) sl[(long)m * g.K + n] = acc[i][j][v];
            }
        }
}

// tile configuration: 256 x 192 with 8 waves (one workgroup per CU) for batches of >= 192 rows, else 128 x 128 / 4 waves
static bool bmu_x3_big(int B) { return B >= 192; }
static int bmu_x3_tiles(int B, int K) { return bmu_x3_big(B) ? cdiv(B, 256) * cdiv(K, 192) : cdiv(B, 128) * cdiv(K, 128); }
static int bmu_x3_splits(int B, int K, int L) {
    const int tiles = bmu_x3_tiles(B, K), ktiles = cdiv(L, 32);
    int s = (bmu_x3_big(B) ? 256 : 512) / tiles;     // one full round of resident workgroups
    if (s > ktiles) s = ktiles;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    const int per = cdiv(ktiles, s);
    return cdiv(ktiles, per);
}

}  // namespace vsom

using namespace vsom;

extern "C" {

size_t vsom_bmu_cosine_x3_workspace_bytes(int B, int K, int L) {
    if (B <= 0 || K <= 0 || L <= 0) return 0;
    const size_t s = (size_t)bmu_x3_splits(B, K, L);
    return (s * ((size_t)B * K + B + K) + 4) * sizeof(float);
}

static int x3_layout(int B, int K, int L, void* ws, BmuP& g, int& splits, int** counter) {
    splits = bmu_x3_splits(B, K, L);
    float* f = static_cast<float*>(ws);
    g.slab = f; g.slab_stride = (long)B * K;
    g.xsq = f + (size_t)splits * B * K; g.wsq = g.xsq + (size_t)splits * B;
    *counter = reinterpret_cast<int*>(g.wsq + (size_t)splits * K);
    return VSOM_OK;
}

/* stage 1: partial dots (three-product bf16 contraction) + partial squared norms into the workspace */
int vsom_bmu_cosine_x3_dots(const float* X, long ldx, const float* W, int B, int K, int L, void* ws, size_t ws_bytes,
                            vsom_stream_t stream) {
    VSOM_REQUIRE(X && W, VSOM_EINVAL, "bmu_cosine_x3_dots: null pointer");
    VSOM_REQUIRE(B > 0 && K > 0 && L > 0 && ldx >= L, VSOM_EINVAL, "bmu_cosine_x3_dots: bad shape B=%d K=%d L=%d ldx=%ld", B, K, L, ldx);
    VSOM_REQUIRE(K <= 256 * BMU_KPT, VSOM_EUNSUPPORTED, "bmu_cosine_x3: more than %d prototypes", 256 * BMU_KPT);
    VSOM_REQUIRE(L % 4 == 0 && ldx % 4 == 0 && aligned16(X) && aligned16(W), VSOM_EALIGN,
                 "bmu_cosine_x3: rows must be 16-byte aligned (L, ldx multiples of 4)");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_bmu_cosine_x3_workspace_bytes(B, K, L) && aligned16(ws), VSOM_EWORKSPACE,
                 "bmu_cosine_x3: workspace too small or misaligned");
    const long xb = ((long)(B - 1) * ldx + L) * 4, wb = (long)K * L * 4;
    VSOM_REQUIRE(xb < 0xFFFF0000L && wb < 0xFFFF0000L, VSOM_EUNSUPPORTED, "bmu_cosine_x3: operand larger than 4 GB");
    BmuP g = {};
    int splits; int* counter;
    x3_layout(B, K, L, ws, g, splits, &counter);
    g.X = X; g.ldx = ldx; g.W = W; g.B = B; g.K = K; g.L = L;
    g.ktiles_per_split = cdiv(cdiv(L, 32), splits);
    g.x_bytes = (unsigned)xb; g.w_bytes = (unsigned)wb;
    if (bmu_x3_big(B)) VSOM_LAUNCH((bmu_x3_kernel<2, 3, 4, 2>), dim3(bmu_x3_tiles(B, K) * splits), dim3(512), 0, stream, g);
    else VSOM_LAUNCH((bmu_x3_kernel<2, 2, 2, 2>), dim3(bmu_x3_tiles(B, K) * splits), dim3(256), 0, stream, g);
    VSOM_LAUNCH_CHECK("bmu_x3_kernel");
}

/* stage 2: norms, distances, first minimum, exact re-rank of the near-minimum candidates.  reranked (nullable):
   device int, incremented once per sample row that had more than one candidate */
int vsom_bmu_cosine_x3_finalize(const float* X, long ldx, const float* W, const void* ws, size_t ws_bytes, float* dist,
                                int64_t* bmu, float* inv_nx, float* inv_nw, int* reranked, int B, int K, int L,
                                vsom_stream_t stream) {
    VSOM_REQUIRE(X && W && bmu && inv_nx && inv_nw, VSOM_EINVAL, "bmu_cosine_x3_finalize: null pointer");
    VSOM_REQUIRE(B > 0 && K > 0 && K <= 256 * BMU_KPT && L > 0 && L % 4 == 0 && ldx % 4 == 0, VSOM_EINVAL, "bmu_cosine_x3_finalize: bad shape");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_bmu_cosine_x3_workspace_bytes(B, K, L), VSOM_EWORKSPACE, "bmu_cosine_x3_finalize: workspace too small");
    BmuP g = {};
    int splits; int* counter;
    x3_layout(B, K, L, const_cast<void*>(ws), g, splits, &counter);
    VSOM_LAUNCH(bmu_norms_kernel, dim3(cdiv(B + K, 256)), dim3(256), 0, stream, g.xsq, g.wsq, splits, B, K, inv_nx, inv_nw);
    int rc = hip_status(hipGetLastError(), "bmu_norms_kernel");
    if (rc) return rc;
    const bool v4 = K % 4 == 0 && aligned16(g.slab) && g.slab_stride % 4 == 0;
    if (v4)
        VSOM_LAUNCH(bmu_x3_finalize_kernel<true>, dim3(B), dim3(256), 0, stream, g.slab, g.slab_stride, splits, X, ldx, W, inv_nx,
                           inv_nw, dist, bmu, K, L, reranked);
    else
    VSOM_LAUNCH(bmu_x3_finalize_kernel<false>, dim3(B), dim3(256), 0, stream, g.slab, g.slab_stride, splits, X, ldx, W, inv_nx,
                       inv_nw, dist, bmu, K, L, reranked);
    VSOM_LAUNCH_CHECK("bmu_x3_finalize_kernel");
}

/* ---------------------------------------------------------------- pre-split plane images (see planes_kernel above) */
static size_t planes_image_bytes(int R, int L) { return (size_t)2 * cdiv(L, 32) * cdiv(R, 32) * 2048; }

size_t vsom_bmu_planes_bytes(int R, int L) {
    if (R <= 0 || L <= 0) return 0;
    return planes_image_bytes(R, L) + (((size_t)cdiv(L, PL_CHUNK) * R * sizeof(float) + 15) & ~(size_t)15);
}

static int planes_params(PlanesP& q, int R, int L, void* planes, size_t planes_bytes, const char* who) {
    VSOM_REQUIRE(R > 0 && L > 0 && L % 8 == 0, VSOM_EINVAL, "%s: bad shape R=%d L=%d (L must be a multiple of 8)", who, R, L);
    VSOM_REQUIRE(planes && aligned16(planes) && planes_bytes >= vsom_bmu_planes_bytes(R, L), VSOM_EWORKSPACE,
                 "%s: plane buffer too small or misaligned", who);
    VSOM_REQUIRE(planes_image_bytes(R, L) < 0x80000000UL, VSOM_EUNSUPPORTED, "%s: image of 2 GB or more", who);
    q.R = R; q.L = L; q.nrb = cdiv(R, 32); q.nst = 2 * cdiv(L, 32);
    q.img = static_cast<uint4*>(planes);
    q.sq = reinterpret_cast<float*>(static_cast<char*>(planes) + planes_image_bytes(R, L));
    return VSOM_OK;
}

int vsom_bmu_planes_from(const float* src, long ld, int R, int L, void* planes, size_t planes_bytes, vsom_stream_t stream) {
    VSOM_REQUIRE(src && ld >= L && ld % 4 == 0 && aligned16(src), VSOM_EINVAL, "bmu_planes_from: bad source (ld=%ld)", ld);
    PlanesP q = {};
    const int rc = planes_params(q, R, L, planes, planes_bytes, "bmu_planes_from");
    if (rc) return rc;
    q.src = src; q.ld = ld;
    VSOM_LAUNCH(planes_kernel<false>, dim3(q.nrb * cdiv(L, PL_CHUNK)), dim3(256), 0, stream, q);
    VSOM_LAUNCH_CHECK("planes_kernel");
}

int vsom_adamw_step_planes(float* p, const float* g, float* m, float* v, const float* wd_per_chunk, long n, float lr,
                           float beta1, float beta2, float eps, int step, float grad_scale, int adamw, long slice_off,
                           int R, int L, void* planes, size_t planes_bytes, vsom_stream_t stream) {
    VSOM_REQUIRE(p && g && m && v && wd_per_chunk, VSOM_EINVAL, "adamw_step_planes: null pointer");
    VSOM_REQUIRE(n > 0 && n % 256 == 0 && step >= 1, VSOM_EINVAL, "adamw_step_planes: bad n=%ld / step=%d", n, step);
    VSOM_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), VSOM_EALIGN, "adamw_step_planes: 16-byte alignment required");
    PlanesP q = {};
    int rc = planes_params(q, R, L, planes, planes_bytes, "adamw_step_planes");
    if (rc) return rc;
    const long len = (long)R * L, end = slice_off + ((len + 255) & ~255L);       // the slice's padding belongs to it
    VSOM_REQUIRE(slice_off >= 0 && slice_off % 256 == 0 && end <= n, VSOM_EINVAL,
                 "adamw_step_planes: slice [%ld, %ld) is not a 256-aligned part of the arena of %ld", slice_off, end, n);
    // one launch: the slice's blocks write parameters, moments and the image; the blocks behind them update the rest of the
    // arena exactly as adamw_kernel does (the slice's own padding is read by nobody and stays as it is)
    q.p = p; q.g = g; q.m = m; q.v = v; q.wd_chunk = wd_per_chunk; q.off = slice_off; q.ld = L;
    q.c = adamw_constants(lr, beta1, beta2, eps, step, grad_scale, adamw);
    q.nblk_planes = q.nrb * cdiv(L, PL_CHUNK);
    q.flat_lo4 = slice_off / 4; q.flat_gap4 = (end - slice_off) / 4; q.flat_n4 = (n - (end - slice_off)) / 4;
    long nflat = (q.flat_n4 + 255) / 256;
    if (nflat > 8192) nflat = 8192;
    VSOM_LAUNCH(planes_kernel<true>, dim3(q.nblk_planes + (int)nflat), dim3(256), 0, stream, q);
    VSOM_LAUNCH_CHECK("planes_kernel<adamw>");
}

int vsom_bmu_cosine_x3_planes_supported(int B, int K, int L) {
    return B > 0 && K > 0 && L > 0 && bmu_x3_big(B) && L % 8 == 0 && K <= 256 * BMU_KPT && planes_image_bytes(K, L) < 0x80000000UL &&
           planes_image_bytes(B, L) < 0x80000000UL;
}

size_t vsom_bmu_cosine_x3_planes_workspace_bytes(int B, int K, int L) {
    if (!vsom_bmu_cosine_x3_planes_supported(B, K, L)) return 0;
    return ((size_t)bmu_x3_splits(B, K, L) * B * K + 4) * sizeof(float);
}

/* stage 1 on pre-split operands: partial dots into the workspace's slabs (bit-identical to vsom_bmu_cosine_x3_dots') */
int vsom_bmu_cosine_x3_planes_dots(const void* xplanes, const void* wplanes, int B, int K, int L, void* ws, size_t ws_bytes,
                                   vsom_stream_t stream) {
    VSOM_REQUIRE(xplanes && wplanes && aligned16(xplanes) && aligned16(wplanes), VSOM_EINVAL, "bmu_cosine_x3_planes_dots: null or misaligned planes");
    VSOM_REQUIRE(vsom_bmu_cosine_x3_planes_supported(B, K, L), VSOM_EUNSUPPORTED,
                 "bmu_cosine_x3_planes_dots: shape B=%d K=%d L=%d not covered (B >= 192, L %% 8 == 0)", B, K, L);
    VSOM_REQUIRE(ws && aligned16(ws) && ws_bytes >= vsom_bmu_cosine_x3_planes_workspace_bytes(B, K, L), VSOM_EWORKSPACE,
                 "bmu_cosine_x3_planes_dots: workspace too small or misaligned");
    static const int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(bmu_x3_planes_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, PL_LDS);
    VSOM_REQUIRE(attr_rc == 0, VSOM_EUNSUPPORTED, "bmu_cosine_x3_planes_dots: cannot reserve %d bytes of LDS", PL_LDS);
    const int splits = bmu_x3_splits(B, K, L);
    BmuPlP g = {};
    g.ximg = xplanes; g.wimg = wplanes;
    g.ximg_bytes = (unsigned)planes_image_bytes(B, L); g.wimg_bytes = (unsigned)planes_image_bytes(K, L);
    g.nrb_x = cdiv(B, 32); g.nrb_w = cdiv(K, 32);
    g.B = B; g.K = K; g.nst = 2 * cdiv(L, 32);
    g.stages_per_split = 2 * cdiv(cdiv(L, 32), splits);
    g.slab = static_cast<float*>(ws); g.slab_stride = (long)B * K;
    VSOM_LAUNCH(bmu_x3_planes_kernel, dim3(bmu_x3_tiles(B, K) * splits), dim3(512), PL_LDS, stream, g);
    VSOM_LAUNCH_CHECK("bmu_x3_planes_kernel");
}

/* stage 2 for the planes form: norms from the images' partials, then the same finalize kernel */
int vsom_bmu_cosine_x3_planes_finalize(const float* X, long ldx, const float* W, const void* xplanes, const void* wplanes,
                                       const void* ws, size_t ws_bytes, float* dist, int64_t* bmu, float* inv_nx, float* inv_nw,
                                       int* reranked, int B, int K, int L, vsom_stream_t stream) {
    VSOM_REQUIRE(X && W && xplanes && wplanes && bmu && inv_nx && inv_nw, VSOM_EINVAL, "bmu_cosine_x3_planes_finalize: null pointer");
    VSOM_REQUIRE(vsom_bmu_cosine_x3_planes_supported(B, K, L) && ldx % 4 == 0 && ldx >= L, VSOM_EINVAL, "bmu_cosine_x3_planes_finalize: bad shape");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_bmu_cosine_x3_planes_workspace_bytes(B, K, L), VSOM_EWORKSPACE,
                 "bmu_cosine_x3_planes_finalize: workspace too small");
    const int splits = bmu_x3_splits(B, K, L);
    const float* xsq = reinterpret_cast<const float*>(static_cast<const char*>(xplanes) + planes_image_bytes(B, L));
    const float* wsq = reinterpret_cast<const float*>(static_cast<const char*>(wplanes) + planes_image_bytes(K, L));
    VSOM_LAUNCH(bmu_norms_wide_kernel, dim3(cdiv(B + K, 64)), dim3(512), 0, stream, xsq, wsq, cdiv(L, PL_CHUNK), B, K, inv_nx, inv_nw);
    int rc = hip_status(hipGetLastError(), "bmu_norms_wide_kernel");
    if (rc) return rc;
    const float* slab = static_cast<const float*>(ws);
    const long slab_stride = (long)B * K;
    if (K % 4 == 0 && aligned16(slab))
        VSOM_LAUNCH(bmu_x3_finalize_kernel<true>, dim3(B), dim3(256), 0, stream, slab, slab_stride, splits, X, ldx, W, inv_nx,
                    inv_nw, dist, bmu, K, L, reranked);
    else
        VSOM_LAUNCH(bmu_x3_finalize_kernel<false>, dim3(B), dim3(256), 0, stream, slab, slab_stride, splits, X, ldx, W, inv_nx,
                    inv_nw, dist, bmu, K, L, reranked);
    VSOM_LAUNCH_CHECK("bmu_x3_finalize_kernel");
}

}  // extern "C"
