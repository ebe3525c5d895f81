// Split-bf16 weight-gradient GEMM:  dW[n,k] = sum_t dY[row(t), n] X[t, k]  (+ db[n] = sum_t dY[row(t), n]).
//
// Both operands are strided along the reduction index t (the token rows).  The fp32 tiles are loaded
// the way they lie in memory -- rows = t, 16-byte loads along the columns -- split into the three
// bf16 planes of gemm_x6.h and written to LDS in the SAME orientation ([t][column], 8-byte stores).
// The MFMA fragment of v_mfma_f32_32x32x16_bf16 wants 8 consecutive t for one column per lane: that
// is what ds_read_b64_tr_b16 delivers from a [t][column] image (a 16-lane group reads a block of
// 4 t-rows x 16 columns and gets it column-major), so there is no transpose in registers and the VALU
// work per element is the split alone.  A plane row is padded to == 64 (mod 128) bytes: the four
// t-rows of a transposed read then tile the 256-byte bank row (conflict-free).
//
// Tiles 192 x 64 or 64 x 192 (outputs x 32 t per step; 4 waves), one k-tile of global loads in flight
// under the MFMAs, two workgroups per CU (61 KB LDS).  The reduction over t is split over workgroups;
// partial tiles go to fp32 slabs summed by reduce_slabs in fixed order (bitwise reproducible).
// The bias gradient rides along: the dY tile passes through this thread's registers anyway, so the
// workgroups of the first column tile keep per-thread column sums and combine them through LDS in a
// fixed order at the end.
#pragma once
#include "gemm_x6.h"

namespace vsom {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct TnP {
    const float* dY; const float* X;
    long ldy, ldx;
    int T, NO, KI;                  // reduction length, output rows (columns of dY), output columns (columns of X)
    int ktiles_per_split;
    int a_seg, a_stride, a_off;     // row(t) = (t / a_seg) * a_stride + a_off + t % a_seg  (a_seg == 0: identity; else a_seg % 32 == 0)
    float* slab; long slab_stride;
    float* slab_bias; long slab_bias_stride;     // or null
    unsigned a_bytes, b_bytes;
};

constexpr int tn_stride(int C) { return ((C * 2 + 63) / 128) * 128 + 64; }

__device__ __forceinline__ s16x4 tn_trread(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// NPL = 3: the exact three-piece split, six products (fp32-accurate: |error| < 2^-22 per product).  NPL = 2: the two-piece
// round-to-nearest split, three products (|error| <= 3 * 2^-16 per product worst case, ~4e-6 relative on a weight gradient
// summed over the token rows): half the matrix-core work, two LDS planes instead of three.
template <int WM, int WN, int WAVES_M, int WAVES_N, int NPL>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void gemm_x6_tn_kernel(const TnP g) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, NT = WAVES_M * WAVES_N * 64;
    constexpr int SA = tn_stride(BM), SB = tn_stride(BN), PA = 32 * SA, PB = 32 * SB;
    constexpr int FA = (32 * BM / 4 + NT - 1) / NT, FB = (32 * BN / 4 + NT - 1) / NT;
    static_assert(NPL * (PA + PB) >= NT * FA * 16, "LDS too small for the bias partials");
    __shared__ __attribute__((aligned(16))) char lds[NPL * (PA + PB)];
    char* As = lds; char* Bs = lds + NPL * PA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm0 = (wave / WAVES_N) * WM * 32, wn0 = (wave % WAVES_N) * WN * 32;
    const int tiles_n = g.KI / BN, tiles_m = g.NO / BM, ntiles = tiles_m * tiles_n;
    // split-major, XCD-contiguous: every XCD owns a slice of the token rows and reads it once
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int z = lid / ntiles, rem = lid - z * ntiles;
    const int bm0 = (rem / tiles_n) * BM, bn0 = (rem % tiles_n) * BN;
    const int ktiles = (g.T + 31) >> 5;
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

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.dY), 0, (int)g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X), 0, (int)g.b_bytes, 0x00020000);
    // staging: float4 number idx = i*NT + t of the [32 t][C/4] tile; a thread keeps the same column quad in every k-tile
    f32x4 sa[FA], sb[FB];
    unsigned ga[FA], gb[FB];
    int la[FA], lb[FB];
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        const int idx = i * NT + t, row = idx / (BM / 4), c4 = idx % (BM / 4);
        const bool ok = idx < 32 * BM / 4;
        ga[i] = ok ? (unsigned)(((long)row * g.ldy + bm0 + c4 * 4) * 4) : OOB;
        la[i] = ok ? row * SA + c4 * 8 : -1;
    }
#pragma unroll
    for (int i = 0; i < FB; ++i) {
        const int idx = i * NT + t, row = idx / (BN / 4), c4 = idx % (BN / 4);
        const bool ok = idx < 32 * BN / 4;
        gb[i] = ok ? (unsigned)(((long)row * g.ldx + bn0 + c4 * 4) * 4) : OOB;
        lb[i] = ok ? row * SB + c4 * 8 : -1;
    }
    const bool want_colsum = g.slab_bias != nullptr && bn0 == 0;
    f32x4 cs[FA];
#pragma unroll
    for (int i = 0; i < FA; ++i) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto gload = [&](int kt) {
        // rows past the end of dY / X lie beyond the descriptor: the loads return zeros
        const long t0 = (long)kt << 5;
        const long ra = g.a_seg ? (t0 / g.a_seg) * g.a_stride + g.a_off + t0 % g.a_seg : t0;
        const unsigned sa_off = (unsigned)(ra * g.ldy * 4), sb_off = (unsigned)(t0 * g.ldx * 4);
#pragma unroll
        for (int i = 0; i < FA; ++i) sa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ga[i], sa_off, 0));
#pragma unroll
        for (int i = 0; i < FB; ++i) sb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, gb[i], sb_off, 0));
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            if (la[i] < 0) continue;
            if (want_colsum) cs[i] += sa[i];
            uint2 p1, p2, p3;
            if constexpr (NPL == 3) x6_split(sa[i], p1, p2, p3); else x3_split(sa[i], p1, p2);
            *reinterpret_cast<uint2*>(As + la[i]) = p1;
            *reinterpret_cast<uint2*>(As + PA + la[i]) = p2;
            if constexpr (NPL == 3) *reinterpret_cast<uint2*>(As + 2 * PA + la[i]) = p3;
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            if (lb[i] < 0) continue;
            uint2 p1, p2, p3;
            if constexpr (NPL == 3) x6_split(sb[i], p1, p2, p3); else x3_split(sb[i], p1, p2);
            *reinterpret_cast<uint2*>(Bs + lb[i]) = p1;
            *reinterpret_cast<uint2*>(Bs + PB + lb[i]) = p2;
            if constexpr (NPL == 3) *reinterpret_cast<uint2*>(Bs + 2 * PB + lb[i]) = p3;
        }
    };
    // transposed fragment reads: lane l -> 16-lane group G = l >> 4 (columns 16 (G & 1) .., t rows 8 (G >> 1) ..),
    // inside the group lane 4q + p supplies the address of t-row q, columns 4p .. 4p + 3
    const int G = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int trow = 8 * (G >> 1) + q, tcol = 16 * (G & 1) + 4 * pp;
    int fa[WM], fbo[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) fa[i] = trow * SA + (wm0 + i * 32 + tcol) * 2;
#pragma unroll
    for (int j = 0; j < WN; ++j) fbo[j] = trow * SB + (wn0 + j * 32 + tcol) * 2;
    auto frag = [&](const char* base, int off, int stride, int ks) -> bf16x8 {
        const s16x4 lo = tn_trread(base + off + (ks * 16) * stride);
        const s16x4 hi = tn_trread(base + off + (ks * 16 + 4) * stride);
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto mfma_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[WM][NPL], b[WN][NPL];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) a[i][pl] = frag(As + pl * PA, fa[i], SA, ks);
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) b[j][pl] = frag(Bs + pl * PB, fbo[j], SB, ks);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    f32x16 c = acc[i][j];
                    if constexpr (NPL == 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);   // 2^-16 terms
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);   // 2^-8 (2^-9) terms
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);   // leading term
                    acc[i][j] = c;
                }
        }
    };
    if (kt_begin < kt_end) {
        gload(kt_begin);
        lstore();
    }
    __syncthreads();
    for (int kt = kt_begin; kt + 1 < kt_end; ++kt) {          // branch-free body, last k-tile peeled (see gemm_x6.h)
        gload(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        lstore();
        __syncthreads();
    }
    if (kt_begin < kt_end) mfma_tile();

    if (want_colsum) {
        // fixed-order combine: every thread parks its FA column-quad partials, then thread c < BM sums the 32
        // partials of column c (t-row 0 .. 31 of the tile image) in row order
        __syncthreads();
        f32x4* park = reinterpret_cast<f32x4*>(lds);
#pragma unroll
        for (int i = 0; i < FA; ++i) park[i * NT + t] = cs[i];
        __syncthreads();
        if (t < BM) {
            const int c4 = t >> 2, e = t & 3;
            float s = 0.f;
#pragma unroll 8
            for (int row = 0; row < 32; ++row) s += reinterpret_cast<const float*>(&park[row * (BM / 4) + c4])[e];
            g.slab_bias[(long)z * g.slab_bias_stride + bm0 + t] = s;
        }
    }
    // slab[z][n * KI + k]: accumulator register v of tile (i, j) holds row (v & 3) + 8 (v >> 2) + 4 h, column r
    const int r = lane & 31, h = lane >> 5;
    float* sl = g.slab + (long)z * g.slab_stride;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float* d = sl + (long)(bm0 + wm0 + i * 32 + 4 * h) * g.KI + bn0 + wn0 + j * 32 + r;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                *d = acc[i][j][v];
                d += (((v & 3) == 3) ? 5 : 1) * (long)g.KI;
            }
        }
}

}  // namespace vsom
