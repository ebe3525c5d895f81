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

}  // extern "C"
