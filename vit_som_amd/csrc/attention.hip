// Multi-head attention forward / backward for short ViT sequences (N = 5..320 tokens),
// exact fp32 on v_mfma_f32_16x16x4_f32.
//
// One workgroup per (image, head); the two [N, hd] operands every query/key tile needs stay
// in LDS for the whole workgroup; each MFMA wave owns 16-row tiles.  Scores are computed
// TRANSPOSED (rows = the LDS operand's 16 rows, columns = the wave's own 16 rows), so the
// MFMA result layout (column on lane&15, 4 consecutive rows in the 4 registers of lane group
// l>>4) is already the B-operand layout of the second product: probabilities never leave
// registers, and softmax statistics are per-lane-column + two cross-lane-group shuffles.
//
//   fwd  : S^T = K q^T  -> online softmax over key chunks -> O^T += V^T P^T
//   dQ   : S^T = K q^T, dP^T = V dO^T, dS^T = P^T (dP^T - D) -> dQ^T += K^T dS^T   (also emits D)
//   dKV  : S = Q k^T, dP = dO v^T, dS = P (dP - D) -> dV^T += dO^T P, dK^T += Q^T dS
//
// The backward recomputes P from the saved log-sum-exp (nothing of size N x N touches HBM).
//
// EXTRA mode (N = 16 m + 1: the ViT case, m*16 patch tokens + the CLS token).  Padding 65 -> 80
// tokens would cost 800 MFMAs per head for 528 useful and a fifth wave doing 160 MFMAs for ONE
// valid row.  Instead tokens 1..16m run as m full, unmasked MFMA tiles (m balanced waves) and token
// 0 is the "extra" row/column: as a key/query it enters every tile wave through two VALU dot
// products and rank-1 updates of the accumulators; as a row of its own (its output / its gradients) it
// is shared out among the tile waves: each covers the 16-token tiles it owns anyway with VALU (4 lanes
// per token for the dots, lane per channel for the sums), the partial results meet in LDS and wave 0
// adds them in wave order.  (It used to be a fifth, VALU-only wave.  A 5-wave workgroup puts two waves
// on one SIMD, and that SIMD's register file then decides the residency of the whole CU: 3 workgroups
// instead of 4 in the forward, ONE instead of 2 in the fused backward -- measured with
// tools/attn_lab.hip / tools/occupancy_probe.hip; with 4 waves the LDS footprint is the limit again.)
#include "common.h"

#include <atomic>

#include <stdlib.h>

// The general kernels, the fused backward and the lab variants (tools/attention_roll_candidate.hip) must give the
// same bits (the tests flip between them): no implicit mul+add contraction -- where the compiler fuses depends on
// the surrounding code, and two forward variants did differ by 1 ulp in 1 % of the outputs at hd = 32.  fmaf()
// where a fused operation is meant.
#pragma clang fp contract(off)

namespace vsom {

int gemm_grad_products();          // gemm_f32.hip: 3 in VSOM_GEMM_SPLIT_BF16_GRAD3 mode

// test / measurement hook (vsom_set_attention_fused): 0 keeps the short-sequence backward as two launches, 1 is the
// default (one launch; scores shared between its phases where the shape allows; at hd = 64 in the default GEMM mode its
// products run on the two-piece bf16 split), 2 one launch with recomputed scores, 3 = 1 with fp32 products in every mode
static std::atomic<int> g_attn_fused{1};

// tools/attn_lab.hip builds this file with VSOM_ATTN_STAMPS: thread 0 of every workgroup records the 100 MHz
// real-time counter at four points (+ the hardware id of its wave); the library build compiles none of it
#ifdef VSOM_ATTN_STAMPS
__device__ unsigned long long* g_attn_stamps = nullptr;          // [grid][16]
#define ATTN_STAMP(i)                                                                              \
    do {                                                                                           \
        if (threadIdx.x == 0 && g_attn_stamps) g_attn_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define ATTN_STAMP_HWID()                                                                          \
    do {                                                                                           \
        if (threadIdx.x == 0 && g_attn_stamps) {                                                   \
            unsigned hw, xcc;                                                                      \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                       \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                     \
            g_attn_stamps[blockIdx.x * 16 + 4] = hw;                                                \
            g_attn_stamps[blockIdx.x * 16 + 5] = xcc;                                               \
        }                                                                                          \
    } while (0)
#else
#define ATTN_STAMP(i)
#define ATTN_STAMP_HWID()
#endif


// LDS image of a [rows, hd] slice: rows padded by 4 floats (row stride = 4 banks mod 64: the fragment reads
// -- 16 rows x one 16-byte chunk per 16-lane group -- and the accumulate reads -- 4 rows x 16 consecutive
// floats -- are conflict-free, and every address is affine in (row, col): immediates, no address arithmetic).
// An unpadded, XOR-swizzled image was measured too (it is the shape a full-wave LDS-DMA load needs): correct
// and conflict-free as well, but the XOR per access costs VALU in kernels that are VALU-bound (forward
// 36.0 -> 36.6 us, fused backward 87.5 -> 91.5 us at N = 65).
template <int HDP>
struct ACfg {
    static constexpr bool VEC = (HDP % 16 == 0);  // head dim fully valid, 16-B vector accesses
    static constexpr int S = HDP + 4;             // LDS row stride (floats); 16-B aligned rows
    static constexpr int NMM = HDP / 4;           // MFMAs (4 deep) per score tile
    static constexpr int NDT = (HDP + 15) / 16;   // 16-wide output tiles over the head dim
    static __device__ __forceinline__ int off(int row, int col) { return row * S + col; }
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- two-piece bf16 form of the score products (hd = 64; the gradient-GEMM mode of gemm_f32.hip applied to the attention
// backward).  An fp32 value is split round-to-nearest into hi + lo (the dropped rest <= 2^-17 relative), a product of two
// fragments is lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x32_bf16: 6 matrix instructions of 16 cycles per score tile instead
// of 16 of 32.  Reduction slot (lane group qp, j) of MFMA m is d = 32 m + 4 qp + j (j < 4) and d = 32 m + 16 + 4 qp + j - 4
// (j >= 4) for BOTH operands -- the two float4 chunks 2 m and 2 m + 1 a lane loads anyway.
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 abf16x2;
struct SplitFrag { abf16x8 hi[2], lo[2]; };              // hd = 64: two 32-deep MFMAs
__device__ __forceinline__ void split8(const float (&v)[8], abf16x8& hi, abf16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const abf16x2 a = {(__bf16)v[2 * e], (__bf16)v[2 * e + 1]};
        const unsigned u = __builtin_bit_cast(unsigned, a);
        const float r0 = v[2 * e] - __uint_as_float(u << 16), r1 = v[2 * e + 1] - __uint_as_float(u & 0xffff0000u);
        const abf16x2 b = {(__bf16)r0, (__bf16)r1};
        h[e] = u; l[e] = __builtin_bit_cast(unsigned, b);
    }
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    hi = __builtin_bit_cast(abf16x8, (u32x4v){h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(abf16x8, (u32x4v){l[0], l[1], l[2], l[3]});
}
// f[4 g + s] = element s of chunk g (load_frag's order)
__device__ __forceinline__ void split_frag(const float (&f)[16], SplitFrag& o) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[8 * m + j];
        split8(v, o.hi[m], o.lo[m]);
    }
}
// o[dt][d = 16 dt + 4 q' + reg][own row] += sum over the 32 tokens of TWO tiles of Z[token][d] * p[token][own row]: the
// accumulate-type product, two 16-token tiles per 32-deep MFMA (slot (qp, j): token 4 qp + j of tile a for j < 4, token
// 4 qp + j - 4 of tile b for j >= 4, both operands alike); pa / pb = the lane's four values of the two tiles.
__device__ __forceinline__ void accum_x3_pair(f32x4 (&o)[4], const float* Zlds, int rowa, int rowb, int r, int qp, f32x4 pa, f32x4 pb) {
    const float pv[8] = {pa[0], pa[1], pa[2], pa[3], pb[0], pb[1], pb[2], pb[3]};
    abf16x8 bh, bl;
    split8(pv, bh, bl);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        float zv[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            zv[j] = Zlds[ACfg<64>::off(rowa + 4 * qp + j, 16 * dt + r)];
            zv[4 + j] = Zlds[ACfg<64>::off(rowb + 4 * qp + j, 16 * dt + r)];
        }
        abf16x8 ah, al;
        split8(zv, ah, al);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, o[dt], 0, 0, 0);
    }
}
__device__ __forceinline__ f32x4 score_x3(const SplitFrag& a, const SplitFrag& b) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo[m], b.hi[m], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi[m], b.lo[m], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi[m], b.hi[m], acc, 0, 0, 0);
    }
    return acc;
}
// Cross-lane reductions without the LDS crossbar (ds_bpermute: an LDS instruction and its latency per step; these
// sit on the kernels' serial chains).  Over the 4 lane groups (l >> 4): v_permlane16_swap / v_permlane32_swap
// (gfx950) of two copies of v leave the even and the odd partner in the two results; within a group of 16
// lanes: DPP row rotations.  Every lane of the reduced set ends with the same bits.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap16(float v, float& a, float& b) {
    const u32x2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(t[0]); b = __uint_as_float(t[1]);
}
__device__ __forceinline__ void swap32(float v, float& a, float& b) {
    const u32x2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(t[0]); b = __uint_as_float(t[1]);
}
__device__ __forceinline__ float group_sum(float v) {      // over the 4 lane groups (l >> 4)
    float a, b;
    swap16(v, a, b); v = a + b;
    swap32(v, a, b); return a + b;
}
__device__ __forceinline__ float group_max(float v) {
    float a, b;
    swap16(v, a, b); v = fmaxf(a, b);
    swap32(v, a, b); return fmaxf(a, b);
}
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {        // CTRL 0x120 + n: rotate right by n within each 16 lanes
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float tile_sum(float v) {       // over the 16 lanes of a group (same l >> 4)
    v += dpp_row<0x128>(v); v += dpp_row<0x124>(v); v += dpp_row<0x122>(v); v += dpp_row<0x121>(v);
    return v;
}
__device__ __forceinline__ float tile_max(float v) {
    v = fmaxf(v, dpp_row<0x128>(v)); v = fmaxf(v, dpp_row<0x124>(v));
    v = fmaxf(v, dpp_row<0x122>(v)); v = fmaxf(v, dpp_row<0x121>(v));
    return v;
}
__device__ __forceinline__ float wave_sum64(float v) { return group_sum(tile_sum(v)); }
// token index of row r of tile t
template <bool EXTRA>
__device__ __forceinline__ int tok(int t, int r) { return (EXTRA ? 1 : 0) + 16 * t + r; }

// stage rows [0,N) of a [N, hd] slice (row stride `rs`) into lds[nrows][S], zero padded
template <int HDP>
__device__ __forceinline__ void stage_rows(float* lds, const float* __restrict__ src, long rs, int N, int nrows,
                                           int hd) {
    constexpr int S = ACfg<HDP>::S;
    if constexpr (ACfg<HDP>::VEC) {
        constexpr int C4 = HDP / 4;
        for (int idx = threadIdx.x; idx < nrows * C4; idx += blockDim.x) {
            const int row = idx / C4, c4 = idx % C4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < N) v = *reinterpret_cast<const f32x4*>(src + (long)row * rs + 4 * c4);
            *reinterpret_cast<f32x4*>(lds + ACfg<HDP>::off(row, 4 * c4)) = v;
        }
    } else {
        for (int idx = threadIdx.x; idx < nrows * HDP; idx += blockDim.x) {
            const int row = idx / HDP, c = idx % HDP;
            lds[row * S + c] = (row < N && c < hd) ? src[(long)row * rs + c] : 0.f;
        }
    }
}
// Two slices at once.  On the vector path every thread first ISSUES up to 4 + 4 sixteen-byte loads
// and only then stores them: the one-slice loop above keeps a single load in flight per thread (load,
// wait, store, next), i.e. four serialised global-memory latencies per slice at N = 65.
template <int HDP>
__device__ __forceinline__ void stage_rows_pair(float* ldsA, const float* __restrict__ srcA, long rsA, float* ldsB,
                                                const float* __restrict__ srcB, long rsB, int N, int nrows, int hd) {
    if constexpr (ACfg<HDP>::VEC) {
        constexpr int C4 = HDP / 4;
        const int total = nrows * C4, step = blockDim.x;
        for (int base = threadIdx.x; base < total; base += 4 * step) {
            f32x4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * step, row = idx / C4, c4 = idx % C4;
                va[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                vb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < total && row < N) {
                    va[u] = *reinterpret_cast<const f32x4*>(srcA + (long)row * rsA + 4 * c4);
                    vb[u] = *reinterpret_cast<const f32x4*>(srcB + (long)row * rsB + 4 * c4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * step, row = idx / C4, c4 = idx % C4;
                if (idx < total) {
                    *reinterpret_cast<f32x4*>(ldsA + ACfg<HDP>::off(row, 4 * c4)) = va[u];
                    *reinterpret_cast<f32x4*>(ldsB + ACfg<HDP>::off(row, 4 * c4)) = vb[u];
                }
            }
        }
    } else {
        stage_rows<HDP>(ldsA, srcA, rsA, N, nrows, hd);
        stage_rows<HDP>(ldsB, srcB, rsB, N, nrows, hd);
    }
}
// Four slices at once (fused backward): 4 x 4 sixteen-byte loads in flight per thread.
template <int HDP>
__device__ __forceinline__ void stage_rows_quad(float* l0, const float* __restrict__ s0, long r0, float* l1,
                                                const float* __restrict__ s1, long r1, float* l2,
                                                const float* __restrict__ s2, long r2, float* l3,
                                                const float* __restrict__ s3, long r3, int N, int nrows) {
    constexpr int C4 = HDP / 4;
    const int total = nrows * C4, step = blockDim.x;
    for (int base = threadIdx.x; base < total; base += 4 * step) {
        f32x4 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * step, row = idx / C4, c4 = idx % C4;
            const bool ok = idx < total && row < N;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v[0][u] = ok ? *reinterpret_cast<const f32x4*>(s0 + (long)row * r0 + 4 * c4) : z;
            v[1][u] = ok ? *reinterpret_cast<const f32x4*>(s1 + (long)row * r1 + 4 * c4) : z;
            v[2][u] = ok ? *reinterpret_cast<const f32x4*>(s2 + (long)row * r2 + 4 * c4) : z;
            v[3][u] = ok ? *reinterpret_cast<const f32x4*>(s3 + (long)row * r3 + 4 * c4) : z;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * step, row = idx / C4, c4 = idx % C4;
            if (idx < total) {
                *reinterpret_cast<f32x4*>(l0 + ACfg<HDP>::off(row, 4 * c4)) = v[0][u];
                *reinterpret_cast<f32x4*>(l1 + ACfg<HDP>::off(row, 4 * c4)) = v[1][u];
                *reinterpret_cast<f32x4*>(l2 + ACfg<HDP>::off(row, 4 * c4)) = v[2][u];
                *reinterpret_cast<f32x4*>(l3 + ACfg<HDP>::off(row, 4 * c4)) = v[3][u];
            }
        }
    }
}
// one row of hd floats -> lds[HDP] (zero padded)
template <int HDP>
__device__ __forceinline__ void stage_vec(float* lds, const float* __restrict__ src, int hd) {
    for (int c = threadIdx.x; c < HDP; c += blockDim.x) lds[c] = (c < hd) ? src[c] : 0.f;
}

// per-lane operand values of one row for all NMM MFMAs.  Lane group qp supplies reduction index
// d = 16g + 4qp + s (vector path, MFMA 4g+s) or d = 4mm + qp (scalar path); both operands of a
// product use the same map, so the assignment is exact.
template <int HDP>
__device__ __forceinline__ void load_frag(float (&f)[ACfg<HDP>::NMM], const float* rowptr, int qp, bool ok,
                                          int hd) {
    if constexpr (ACfg<HDP>::VEC) {
#pragma unroll
        for (int g = 0; g < HDP / 16; ++g) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(rowptr + 16 * g + 4 * qp);
#pragma unroll
            for (int s = 0; s < 4; ++s) f[4 * g + s] = v[s];
        }
    } else {
#pragma unroll
        for (int mm = 0; mm < ACfg<HDP>::NMM; ++mm) {
            const int d = 4 * mm + qp;
            f[mm] = (ok && d < hd) ? rowptr[d] : 0.f;
        }
    }
}

// the same fragment of row `row` of an LDS slice image
template <int HDP>
__device__ __forceinline__ void load_frag_lds(float (&f)[ACfg<HDP>::NMM], const float* Y, int row, int qp) {
    if constexpr (ACfg<HDP>::VEC) {
#pragma unroll
        for (int g = 0; g < HDP / 16; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(Y + ACfg<HDP>::off(row, 16 * g + 4 * qp));
#pragma unroll
            for (int s = 0; s < 4; ++s) f[4 * g + s] = v[s];
        }
    } else {
        load_frag<HDP>(f, Y + row * ACfg<HDP>::S, qp, true, HDP);
    }
}

// acc[4q'+reg][own row] = sum_d Y[row0 + 4q'+reg][d] * own[row][d]    (row0 = first token of the tile)
template <int HDP>
__device__ __forceinline__ f32x4 score_tile(const float* Ylds, int row0, int r, int qp,
                                            const float (&bf)[ACfg<HDP>::NMM]) {
    float af[ACfg<HDP>::NMM];
    load_frag_lds<HDP>(af, Ylds, row0 + r, qp);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mm = 0; mm < ACfg<HDP>::NMM; ++mm) acc = mfma16(af[mm], bf[mm], acc);
    return acc;
}
// two score tiles with interleaved MFMAs (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle result)
template <int HDP>
__device__ __forceinline__ void score_tile2(const float* Y0, int row0, const float (&b0)[ACfg<HDP>::NMM], const float* Y1,
                                            int row1, const float (&b1)[ACfg<HDP>::NMM], int r, int qp, f32x4& acc0,
                                            f32x4& acc1) {
    float a0[ACfg<HDP>::NMM], a1[ACfg<HDP>::NMM];
    load_frag_lds<HDP>(a0, Y0, row0 + r, qp);
    load_frag_lds<HDP>(a1, Y1, row1 + r, qp);
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mm = 0; mm < ACfg<HDP>::NMM; ++mm) {
        acc0 = mfma16(a0[mm], b0[mm], acc0);
        acc1 = mfma16(a1[mm], b1[mm], acc1);
    }
}

// o[dt][d = 16dt + 4q'+reg][own row] += sum_{j in tile} Z[row0 + j][d] * p[j][own row]
template <int HDP>
__device__ __forceinline__ void accum_tile(f32x4 (&o)[ACfg<HDP>::NDT], const float* Zlds, int row0, int r, int qp,
                                           f32x4 p) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {                // s outer: the NDT accumulators form independent chains
#pragma unroll
        for (int dt = 0; dt < ACfg<HDP>::NDT; ++dt) {
            float a = 0.f;
            if (ACfg<HDP>::VEC || 16 * dt + r < HDP) a = Zlds[ACfg<HDP>::off(row0 + 4 * qp + s, 16 * dt + r)];
            o[dt] = mfma16(a, p[s], o[dt]);
        }
    }
}
template <int HDP>
__device__ __forceinline__ void accum_tile2(f32x4 (&o0)[ACfg<HDP>::NDT], const float* Z0, f32x4 p0,
                                            f32x4 (&o1)[ACfg<HDP>::NDT], const float* Z1, f32x4 p1, int row0, int r,
                                            int qp) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int dt = 0; dt < ACfg<HDP>::NDT; ++dt) {
            float a0 = 0.f, a1 = 0.f;
            if (ACfg<HDP>::VEC || 16 * dt + r < HDP) {
                a0 = Z0[ACfg<HDP>::off(row0 + 4 * qp + s, 16 * dt + r)];
                a1 = Z1[ACfg<HDP>::off(row0 + 4 * qp + s, 16 * dt + r)];
            }
            o0[dt] = mfma16(a0, p0[s], o0[dt]);
            o1[dt] = mfma16(a1, p1[s], o1[dt]);
        }
    }
}

// store o^T tiles to row `dst` (row pointer at column 0 of this head), columns 16dt + 4qp .. +3
template <int HDP>
__device__ __forceinline__ void store_rows(const f32x4 (&o)[ACfg<HDP>::NDT], float* dst, int qp, bool ok, int hd) {
    if (!ok) return;
#pragma unroll
    for (int dt = 0; dt < ACfg<HDP>::NDT; ++dt) {
        const int d0 = 16 * dt + 4 * qp;
        if constexpr (ACfg<HDP>::VEC) {
            *reinterpret_cast<f32x4*>(dst + d0) = o[dt];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (d0 + e < hd) dst[d0 + e] = o[dt][e];
        }
    }
}

// ---- EXTRA-token helpers (VALU) -----------------------------------------------------------------
// full dot product of each of the wave's 16 own rows (fragments in registers) with ONE LDS row
template <int HDP>
__device__ __forceinline__ float frag_dot_row(const float (&f)[ACfg<HDP>::NMM], const float* rowptr, int qp) {
    float y[ACfg<HDP>::NMM];
    load_frag<HDP>(y, rowptr, qp, true, HDP);
    float s = 0.f;
#pragma unroll
    for (int mm = 0; mm < ACfg<HDP>::NMM; ++mm) s = fmaf(f[mm], y[mm], s);
    return group_sum(s);
}
// o[dt][d][own row] += w[own row] * row[d]   (rank-1 update in the accumulator layout)
template <int HDP>
__device__ __forceinline__ void axpy_row(f32x4 (&o)[ACfg<HDP>::NDT], float w, const float* rowptr, int qp) {
#pragma unroll
    for (int dt = 0; dt < ACfg<HDP>::NDT; ++dt) {
        const int d0 = 16 * dt + 4 * qp;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ACfg<HDP>::VEC) {
            v = *reinterpret_cast<const f32x4*>(rowptr + d0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (d0 + e < HDP) v[e] = rowptr[d0 + e];
        }
        o[dt] += w * v;
    }
}
// ---- token 0 as a row of its own, one 16-token tile at a time (any tile wave) ----------------------
// dot of ONE vector x[HDP] (in LDS) with each of the 16 rows row0 + r of Y: the result for row r sits on
// the lanes (r, *).  (x is re-read per tile on purpose: a fragment kept across the tile loop costs 16
// registers of a kernel whose residency is register-bound.)
template <int HDP>
__device__ __forceinline__ float tile_rows_dot(const float* x, const float* Y, int row0, int r, int qp) {
    float xf[ACfg<HDP>::NMM], y[ACfg<HDP>::NMM];
    load_frag<HDP>(xf, x, qp, true, HDP);
    load_frag_lds<HDP>(y, Y, row0 + r, qp);
    float s = 0.f;
#pragma unroll
    for (int mm = 0; mm < ACfg<HDP>::NMM; ++mm) s = fmaf(xf[mm], y[mm], s);
    return group_sum(s);
}
// sum_j w_j * Z[row0 + j][lane] over the 16 rows of the tile (w_j lives on lane j; lane = channel < HDP)
template <int HDP>
__device__ __forceinline__ float tile_wsum(float w, const float* Z, int row0, int lane) {
    float acc = 0.f;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const float wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), j));
        if (HDP >= 64 || lane < HDP) acc = fmaf(wj, Z[ACfg<HDP>::off(row0 + j, lane)], acc);
    }
    return acc;
}
// x . y for two [HDP] vectors in LDS, on every lane
template <int HDP>
__device__ __forceinline__ float vec_dot(const float* x, const float* y, int lane) {
    return wave_sum64((HDP >= 64 || lane < HDP) ? x[lane] * y[lane] : 0.f);
}
// dQ of token 0, the part over the 16 keys of one tile: w_j = p_0j (dP_0j - D_0) scale, sum_j w_j K_j
template <int HDP>
__device__ __forceinline__ float tok0_dq_partial(const float* q0f, const float* do0f, const float* Ks, const float* Vs, int row0, float l0, float D0,
                                                 float scale, int lane, int r, int qp) {
    const float sc = tile_rows_dot<HDP>(q0f, Ks, row0, r, qp);
    const float dp = tile_rows_dot<HDP>(do0f, Vs, row0, r, qp);
    const float w = __expf(sc * scale - l0) * (dp - D0) * scale;
    return tile_wsum<HDP>(w, Ks, row0, lane);
}
// wave 0: key 0's own term + the partials of the tile waves in wave order -> dqkv row 0 (q slice)
template <int HDP>
__device__ __forceinline__ void tok0_dq_combine(const float* q0, const float* do0, const float* k0, const float* v0,
                                                const float* PA, int pa_stride, int nwaves, float l0, float D0,
                                                float scale, float* dst, int hd, int lane) {
    const float s00 = vec_dot<HDP>(q0, k0, lane);
    const float dp00 = vec_dot<HDP>(do0, v0, lane);
    const float w00 = __expf(s00 * scale - l0) * (dp00 - D0) * scale;
    if (HDP >= 64 || lane < HDP) {
        float g = w00 * k0[lane];
        for (int w = 0; w < nwaves; ++w) g += PA[w * pa_stride + lane];
        if (lane < hd) dst[lane] = g;
    }
}
// dK, dV of token 0, the part over the 16 queries of one tile
template <int HDP>
__device__ __forceinline__ void tok0_dkv_partial(const float* k0f, const float* v0f, const float* Qs, const float* Ds, const float* Ls, const float* Es,
                                                 int row0, float scale, int lane, int r, int qp, float& gk, float& gv) {
    const float sc = tile_rows_dot<HDP>(k0f, Qs, row0, r, qp);
    const float dp = tile_rows_dot<HDP>(v0f, Ds, row0, r, qp);
    const float p = __expf(sc * scale - Ls[row0 + r]);
    const float ds = p * (dp - Es[row0 + r]) * scale;
    gv += tile_wsum<HDP>(p, Ds, row0, lane);
    gk += tile_wsum<HDP>(ds, Qs, row0, lane);
}
// wave 0: query 0's own term + the partials (PA rows: [gk | gv], HDP each) -> dqkv row 0 (k and v slices)
template <int HDP>
__device__ __forceinline__ void tok0_dkv_combine(const float* q0, const float* do0, const float* k0, const float* v0,
                                                 const float* PA, int pa_stride, int nwaves, float l0, float D0,
                                                 float scale, float* dk_dst, float* dv_dst, int hd, int lane) {
    const float s00 = vec_dot<HDP>(q0, k0, lane);
    const float dp00 = vec_dot<HDP>(do0, v0, lane);
    const float p00 = __expf(s00 * scale - l0);
    const float ds00 = p00 * (dp00 - D0) * scale;
    if (HDP >= 64 || lane < HDP) {
        float gk = ds00 * q0[lane], gv = p00 * do0[lane];
        for (int w = 0; w < nwaves; ++w) {
            gk += PA[w * pa_stride + lane];
            gv += PA[w * pa_stride + HDP + lane];
        }
        if (lane < hd) {
            dk_dst[lane] = gk;
            dv_dst[lane] = gv;
        }
    }
}

// LDS carve shared by the three two-slice kernels: two [nrows][S] slices, optional row statistics, the two
// token-0 vectors of the EXTRA layout, and the tile waves' token-0 partials (`paw` floats per wave)
template <int HDP, bool EXTRA>
struct Carve {
    int ntile, nrows, nrp;
    float *Y0, *Y1, *L0, *L1, *X0, *X1, *PA;
    __device__ __forceinline__ Carve(float* smem, int N, bool with_stats) {
        constexpr int S = ACfg<HDP>::S;
        ntile = EXTRA ? (N - 1) >> 4 : (N + 15) >> 4;
        nrows = EXTRA ? N : ntile << 4;
        nrp = (nrows + 3) & ~3;
        Y0 = smem;
        Y1 = Y0 + nrows * S;
        L0 = Y1 + nrows * S;
        L1 = L0 + (with_stats ? nrp : 0);
        X0 = L1 + (with_stats ? nrp : 0);
        X1 = X0 + HDP;
        PA = X1 + HDP;
    }
};

// ------------------------------------------------------------------ forward
template <int HDP, bool EXTRA>
__global__ __launch_bounds__(512, 4) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                       float* __restrict__ lse, int N, int H, int hd,
                                                       float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NDT = ACfg<HDP>::NDT;
    constexpr int NMM = ACfg<HDP>::NMM;
    constexpr int S = ACfg<HDP>::S;
    constexpr int PAW = HDP + 2;                   // token-0 partial of a wave: o[HDP], m, l
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * hd, E3 = 3 * E;
    const Carve<HDP, EXTRA> cv(smem, N, false);
    const int ntile = cv.ntile;
    float* Ks = cv.Y0;
    float* Vs = cv.Y1;
    const float* base = qkv + (long)b * N * E3 + h * hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, qp = lane >> 4;
    // the wave's first query fragment is requested BEFORE the K/V staging (latency overlaps it)
    float qf[NMM];
    ATTN_STAMP(0);
    ATTN_STAMP_HWID();
    load_frag<HDP>(qf, base + (long)tok<EXTRA>(wave, r) * E3, qp, tok<EXTRA>(wave, r) < N, hd);
    stage_rows_pair<HDP>(Ks, base + E, E3, Vs, base + 2 * E, E3, N, cv.nrows, hd);
    if (EXTRA) stage_vec<HDP>(cv.X0, base, hd);                       // q of token 0
    __syncthreads();
    ATTN_STAMP(1);

    // token 0 as a query: running softmax state over the key tiles this wave owns (lane = channel for x0o)
    float x0m = -INFINITY, x0l = 0.f, x0o = 0.f;

    for (int qt = wave; qt < ntile; qt += nwaves) {
        const int query = tok<EXTRA>(qt, r);
        const bool qok = query < N;
        if (qt != wave) load_frag<HDP>(qf, base + (long)query * E3, qp, qok, hd);
        float m = -INFINITY, l = 0.f;
        f32x4 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EXTRA) {                                                   // token 0 as a key: running state starts from it
            m = frag_dot_row<HDP>(qf, Ks, qp) * scale;
            l = 1.0f;
            axpy_row<HDP>(o, 1.0f, Vs, qp);
        }
        for (int c0 = 0; c0 < ntile; c0 += 4) {
            f32x4 s[4];
            float cmax = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < 4; tt += 2) {
                const int t = c0 + tt;
                if (t + 1 < ntile) score_tile2<HDP>(Ks, tok<EXTRA>(t, 0), qf, Ks, tok<EXTRA>(t + 1, 0), qf, r, qp, s[tt], s[tt + 1]);
                else if (t < ntile) s[tt] = score_tile<HDP>(Ks, tok<EXTRA>(t, 0), r, qp, qf);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (t + u < ntile) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int key = tok<EXTRA>(t + u, 4 * qp + e);
                            s[tt + u][e] = (EXTRA || key < N) ? s[tt + u][e] * scale : -INFINITY;
                            cmax = fmaxf(cmax, s[tt + u][e]);
                        }
                    } else {
                        s[tt + u] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    }
                }
            }
            cmax = group_max(cmax);
            const float mnew = fmaxf(m, cmax);
            const float alpha = __expf(m - mnew);
            float psum = 0.f;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __expf(s[tt][e] - mnew);
                    s[tt][e] = p;
                    psum += p;
                }
            psum = group_sum(psum);
            l = l * alpha + psum;
            m = mnew;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                if (c0 + tt < ntile) accum_tile<HDP>(o, Vs, tok<EXTRA>(c0 + tt, 0), r, qp, s[tt]);
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] *= inv;
        ATTN_STAMP(2);
        store_rows<HDP>(o, out + ((long)b * N + query) * E + h * hd, qp, qok, hd);
        if (qp == 0 && qok) lse[((long)b * H + h) * N + query] = m + logf(l);

        if (EXTRA) {                                                   // token 0 as a query against this tile's 16 keys
            __builtin_amdgcn_sched_barrier(0);                         // keep its LDS reads out of the tile's register peak
            const int row0 = tok<EXTRA>(qt, 0);
            const float sc = tile_rows_dot<HDP>(cv.X0, Ks, row0, r, qp) * scale;
            const float mnew = fmaxf(x0m, tile_max(sc));
            const float p = __expf(sc - mnew);
            const float alpha = __expf(x0m - mnew);
            x0l = x0l * alpha + tile_sum(p);
            x0o = x0o * alpha + tile_wsum<HDP>(p, Vs, row0, lane);
            x0m = mnew;
        }
    }
    if (EXTRA) {
        float* pa = cv.PA + wave * PAW;
        if (HDP >= 64 || lane < HDP) pa[lane] = x0o;
        if (lane == 0) { pa[HDP] = x0m; pa[HDP + 1] = x0l; }
        const float s00 = (wave == 0) ? vec_dot<HDP>(cv.X0, Ks, lane) * scale : 0.f;
        __syncthreads();
        if (wave == 0) {                                               // key 0 itself, then the waves in order
            float m = s00;
            for (int w = 0; w < nwaves; ++w) m = fmaxf(m, cv.PA[w * PAW + HDP]);
            float l = __expf(s00 - m);
            float o = (HDP >= 64 || lane < HDP) ? l * Vs[lane] : 0.f;
            for (int w = 0; w < nwaves; ++w) {
                const float a = __expf(cv.PA[w * PAW + HDP] - m);
                l = fmaf(cv.PA[w * PAW + HDP + 1], a, l);
                if (HDP >= 64 || lane < HDP) o = fmaf(cv.PA[w * PAW + lane], a, o);
            }
            if (lane < hd) out[((long)b * N) * E + h * hd + lane] = o / l;
            if (lane == 0) lse[((long)b * H + h) * N] = m + logf(l);
        }
    }
    ATTN_STAMP(3);
    (void)S;
}

// ------------------------------------------------------------------ backward: dQ (+ D = rowsum(dO * O))
template <int HDP, bool EXTRA>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const float* __restrict__ qkv,
                                                          const float* __restrict__ out,
                                                          const float* __restrict__ dout,
                                                          const float* __restrict__ lse,
                                                          float* __restrict__ dqkv, float* __restrict__ delta,
                                                          int N, int H, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NDT = ACfg<HDP>::NDT;
    constexpr int NMM = ACfg<HDP>::NMM;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * hd, E3 = 3 * E;
    const Carve<HDP, EXTRA> cv(smem, N, false);
    const int ntile = cv.ntile;
    float* Ks = cv.Y0;
    float* Vs = cv.Y1;
    const float* base = qkv + (long)b * N * E3 + h * hd;
    const long obase = (long)b * N * E + h * hd;
    const long srow0 = ((long)b * H + h) * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, qp = lane >> 4;
    float qf[NMM], dof[NMM], of[NMM];
    float lq_first = 0.f;                          // log-sum-exp of the wave's first query row, requested with the fragments
    {
        const int q0 = tok<EXTRA>(wave, r);
        load_frag<HDP>(qf, base + (long)q0 * E3, qp, q0 < N, hd);
        load_frag<HDP>(dof, dout + obase + (long)q0 * E, qp, q0 < N, hd);
        load_frag<HDP>(of, out + obase + (long)q0 * E, qp, q0 < N, hd);
        if (q0 < N) lq_first = lse[srow0 + q0];
    }
    float o0 = 0.f, l0 = 0.f;                      // token 0: its output row (for D_0) and log-sum-exp
    if (EXTRA) {
        if (lane < hd) o0 = out[obase + lane];
        l0 = lse[srow0];
    }
    stage_rows_pair<HDP>(Ks, base + E, E3, Vs, base + 2 * E, E3, N, cv.nrows, hd);
    if (EXTRA) {
        stage_vec<HDP>(cv.X0, base, hd);                               // q of token 0
        stage_vec<HDP>(cv.X1, dout + obase, hd);                       // dO of token 0
    }
    __syncthreads();

    float D0 = 0.f, gq0 = 0.f;
    if (EXTRA) {
        D0 = wave_sum64((HDP >= 64 || lane < HDP) ? cv.X1[lane] * o0 : 0.f);
        if (wave == 0 && lane == 0) delta[srow0] = D0;
    }

    for (int qt = wave; qt < ntile; qt += nwaves) {
        const int query = tok<EXTRA>(qt, r);
        const bool qok = query < N;
        if (qt != wave) {
            load_frag<HDP>(qf, base + (long)query * E3, qp, qok, hd);
            load_frag<HDP>(dof, dout + obase + (long)query * E, qp, qok, hd);
            load_frag<HDP>(of, out + obase + (long)query * E, qp, qok, hd);
        }
        float D = 0.f;
#pragma unroll
        for (int mm = 0; mm < NMM; ++mm) D = fmaf(dof[mm], of[mm], D);
        D = group_sum(D);
        const long srow = srow0 + query;
        if (qp == 0 && qok) delta[srow] = D;
        const float lq = (qt == wave) ? lq_first : (qok ? lse[srow] : 0.f);
        f32x4 dq[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EXTRA) {                                                   // token 0 as a key
            const float s0 = frag_dot_row<HDP>(qf, Ks, qp) * scale;
            const float dp0 = frag_dot_row<HDP>(dof, Vs, qp);
            const float p0 = __expf(s0 - lq);
            axpy_row<HDP>(dq, p0 * (dp0 - D) * scale, Ks, qp);
        }
        for (int t = 0; t < ntile; ++t) {
            f32x4 s, dp;
            score_tile2<HDP>(Ks, tok<EXTRA>(t, 0), qf, Vs, tok<EXTRA>(t, 0), dof, r, qp, s, dp);
            f32x4 ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = tok<EXTRA>(t, 4 * qp + e);
                const float p = (EXTRA || (key < N && qok)) ? __expf(s[e] * scale - lq) : 0.f;
                ds[e] = p * (dp[e] - D) * scale;
            }
            accum_tile<HDP>(dq, Ks, tok<EXTRA>(t, 0), r, qp, ds);
        }
        store_rows<HDP>(dq, dqkv + ((long)b * N + query) * E3 + h * hd, qp, qok, hd);
        if (EXTRA) gq0 += tok0_dq_partial<HDP>(cv.X0, cv.X1, Ks, Vs, tok<EXTRA>(qt, 0), l0, D0, scale, lane, r, qp);
    }
    if (EXTRA) {
        if (HDP >= 64 || lane < HDP) cv.PA[wave * HDP + lane] = gq0;
        __syncthreads();
        if (wave == 0)
            tok0_dq_combine<HDP>(cv.X0, cv.X1, Ks, Vs, cv.PA, HDP, nwaves, l0, D0, scale,
                                 dqkv + (long)b * N * E3 + h * hd, hd, lane);
    }
}

// ------------------------------------------------------------------ backward: dK, dV
template <int HDP, bool EXTRA>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const float* __restrict__ qkv,
                                                           const float* __restrict__ dout,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta,
                                                           float* __restrict__ dqkv, int N, int H, int hd,
                                                           float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NDT = ACfg<HDP>::NDT;
    constexpr int NMM = ACfg<HDP>::NMM;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * hd, E3 = 3 * E;
    const Carve<HDP, EXTRA> cv(smem, N, true);
    const int ntile = cv.ntile;
    float* Qs = cv.Y0;
    float* Ds = cv.Y1;
    float* Ls = cv.L0;
    float* Es = cv.L1;
    const float* base = qkv + (long)b * N * E3 + h * hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, qp = lane >> 4;
    float kf[NMM], vf[NMM];
    {
        const int k0 = tok<EXTRA>(wave, r);
        load_frag<HDP>(kf, base + (long)k0 * E3 + E, qp, k0 < N, hd);
        load_frag<HDP>(vf, base + (long)k0 * E3 + 2 * E, qp, k0 < N, hd);
    }
    // row statistics: the first blockDim rows are requested before the slice staging (one exposed
    // global latency less), the rest after
    const int i0 = threadIdx.x;
    float l_r = 0.f, e_r = 0.f;
    if (i0 < N) {
        l_r = lse[((long)b * H + h) * N + i0];
        e_r = delta[((long)b * H + h) * N + i0];
    }
    stage_rows_pair<HDP>(Qs, base, E3, Ds, dout + (long)b * N * E + h * hd, E, N, cv.nrows, hd);
    if (i0 < cv.nrp) { Ls[i0] = l_r; Es[i0] = e_r; }
    for (int i = threadIdx.x + blockDim.x; i < cv.nrp; i += blockDim.x) {
        const long srow = ((long)b * H + h) * N + i;
        Ls[i] = (i < N) ? lse[srow] : 0.f;
        Es[i] = (i < N) ? delta[srow] : 0.f;
    }
    if (EXTRA) {
        stage_vec<HDP>(cv.X0, base + E, hd);                           // k of token 0
        stage_vec<HDP>(cv.X1, base + 2 * E, hd);                       // v of token 0
    }
    __syncthreads();

    float gk0 = 0.f, gv0 = 0.f;

    for (int kt = wave; kt < ntile; kt += nwaves) {
        const int key = tok<EXTRA>(kt, r);
        const bool kok = key < N;
        if (kt != wave) {
            load_frag<HDP>(kf, base + (long)key * E3 + E, qp, kok, hd);
            load_frag<HDP>(vf, base + (long)key * E3 + 2 * E, qp, kok, hd);
        }
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (EXTRA) {                                                   // token 0 as a query
            const float s0 = frag_dot_row<HDP>(kf, Qs, qp) * scale;
            const float dp0 = frag_dot_row<HDP>(vf, Ds, qp);
            const float p0 = __expf(s0 - Ls[0]);
            axpy_row<HDP>(dv, p0, Ds, qp);
            axpy_row<HDP>(dk, p0 * (dp0 - Es[0]) * scale, Qs, qp);
        }
        for (int t = 0; t < ntile; ++t) {
            f32x4 s, dp;                                              // rows: queries of tile t, col: own key
            score_tile2<HDP>(Qs, tok<EXTRA>(t, 0), kf, Ds, tok<EXTRA>(t, 0), vf, r, qp, s, dp);
            f32x4 p, ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int query = tok<EXTRA>(t, 4 * qp + e);
                p[e] = (EXTRA || (query < N && kok)) ? __expf(s[e] * scale - Ls[query]) : 0.f;
                ds[e] = p[e] * (dp[e] - Es[query]) * scale;
            }
            accum_tile2<HDP>(dv, Ds, p, dk, Qs, ds, tok<EXTRA>(t, 0), r, qp);
        }
        float* drow = dqkv + ((long)b * N + key) * E3 + h * hd;
        store_rows<HDP>(dk, drow + E, qp, kok, hd);
        store_rows<HDP>(dv, drow + 2 * E, qp, kok, hd);
        if (EXTRA) tok0_dkv_partial<HDP>(cv.X0, cv.X1, Qs, Ds, Ls, Es, tok<EXTRA>(kt, 0), scale, lane, r, qp, gk0, gv0);
    }
    if (EXTRA) {
        if (HDP >= 64 || lane < HDP) {
            cv.PA[wave * 2 * HDP + lane] = gk0;
            cv.PA[wave * 2 * HDP + HDP + lane] = gv0;
        }
        __syncthreads();
        if (wave == 0) {
            float* drow = dqkv + (long)b * N * E3 + h * hd;
            tok0_dkv_combine<HDP>(Qs, Ds, cv.X0, cv.X1, cv.PA, 2 * HDP, nwaves, Ls[0], Es[0], scale, drow + E,
                                  drow + 2 * E, hd, lane);
        }
    }
}

// ------------------------------------------------------------------ backward, fused (short sequences)
// dQ and dK/dV in ONE launch when all four slices (K, V, Q, dO) of an (image, head) fit in LDS next to
// each other twice per CU (N = 65, hd = 64: 74 KB): the slices are staged once, D = rowsum(dO * O) goes
// from the dQ phase to the dK/dV phase through LDS, and the second kernel's launch, staging and prologue
// disappear.  Phase 1 is attn_bwd_dq_kernel's body (waves own query tiles), phase 2
// attn_bwd_dkv_kernel's (waves own key tiles), with the same arithmetic in the same order (the two
// forms give identical bits); the token-0 vectors of the EXTRA path are row 0 of the staged slices.
template <int HDP, bool EXTRA>
__global__ __launch_bounds__(512) void attn_bwd_fused_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                             const float* __restrict__ dout, const float* __restrict__ lse,
                                                             float* __restrict__ dqkv, float* __restrict__ delta, int N,
                                                             int H, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NDT = ACfg<HDP>::NDT;
    constexpr int NMM = ACfg<HDP>::NMM;
    constexpr int S = ACfg<HDP>::S;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * hd, E3 = 3 * E;
    const int ntile = EXTRA ? (N - 1) >> 4 : (N + 15) >> 4;
    const int nrows = EXTRA ? N : ntile << 4;
    const int nrp = (nrows + 3) & ~3;
    float* Ks = smem;
    float* Vs = Ks + nrows * S;
    float* Qs = Vs + nrows * S;
    float* Ds = Qs + nrows * S;
    float* Ls = Ds + nrows * S;
    float* Es = Ls + nrp;
    float* PA = Es + nrp;                                              // [nwaves][3 HDP]: gq | gk | gv of token 0
    const float* base = qkv + (long)b * N * E3 + h * hd;
    const long obase = (long)b * N * E + h * hd;
    const long srow0 = ((long)b * H + h) * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, qp = lane >> 4;
    float qf[NMM], dof[NMM], of[NMM];
    ATTN_STAMP(0);
    ATTN_STAMP_HWID();
    {
        const int q0 = tok<EXTRA>(wave, r);
        load_frag<HDP>(qf, base + (long)q0 * E3, qp, q0 < N, hd);
        load_frag<HDP>(dof, dout + obase + (long)q0 * E, qp, q0 < N, hd);
        load_frag<HDP>(of, out + obase + (long)q0 * E, qp, q0 < N, hd);
    }
    float o0 = 0.f;
    if (EXTRA && lane < hd) o0 = out[obase + lane];
    float l_r = 0.f;
    if ((int)threadIdx.x < N) l_r = lse[srow0 + threadIdx.x];
    stage_rows_quad<HDP>(Ks, base + E, E3, Vs, base + 2 * E, E3, Qs, base, E3, Ds, dout + obase, E, N, nrows);
    for (int i = threadIdx.x; i < nrp; i += blockDim.x) {
        Ls[i] = (i == (int)threadIdx.x) ? l_r : ((i < N) ? lse[srow0 + i] : 0.f);
        Es[i] = 0.f;
    }
    __syncthreads();
    ATTN_STAMP(1);
#ifdef VSOM_ATTN_REPEAT     // lab only: the compute of REPEAT items behind ONE staging (what a perfect prefetch could reach)
    for (int rep_ = 0; rep_ < VSOM_ATTN_REPEAT; ++rep_) {
    __syncthreads();
#endif

    // ---- phase 1: dQ and D
    float D0 = 0.f, l0 = 0.f, gq0 = 0.f;
    if (EXTRA) {                                                       // token 0's vectors are row 0 of the staged slices
        D0 = wave_sum64((HDP >= 64 || lane < HDP) ? Ds[lane] * o0 : 0.f);
        l0 = Ls[0];
        if (wave == 0 && lane == 0) { delta[srow0] = D0; Es[0] = D0; }
    }
    for (int qt = wave; qt < ntile; qt += nwaves) {
        const int query = tok<EXTRA>(qt, r);
        const bool qok = query < N;
        if (qt != wave) {
            load_frag<HDP>(qf, base + (long)query * E3, qp, qok, hd);
            load_frag<HDP>(dof, dout + obase + (long)query * E, qp, qok, hd);
            load_frag<HDP>(of, out + obase + (long)query * E, qp, qok, hd);
        }
        float D = 0.f;
#pragma unroll
        for (int mm = 0; mm < NMM; ++mm) D = fmaf(dof[mm], of[mm], D);
        D = group_sum(D);
        if (qp == 0 && qok) { delta[srow0 + query] = D; Es[query] = D; }
        const float lq = qok ? Ls[query] : 0.f;
        f32x4 dq[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EXTRA) {                                                   // token 0 as a key
            const float s0 = frag_dot_row<HDP>(qf, Ks, qp) * scale;
            const float dp0 = frag_dot_row<HDP>(dof, Vs, qp);
            const float p0 = __expf(s0 - lq);
            axpy_row<HDP>(dq, p0 * (dp0 - D) * scale, Ks, qp);
        }
        for (int t = 0; t < ntile; ++t) {
            f32x4 sc, dp;
            score_tile2<HDP>(Ks, tok<EXTRA>(t, 0), qf, Vs, tok<EXTRA>(t, 0), dof, r, qp, sc, dp);
            f32x4 ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = tok<EXTRA>(t, 4 * qp + e);
                const float p = (EXTRA || (key < N && qok)) ? __expf(sc[e] * scale - lq) : 0.f;
                ds[e] = p * (dp[e] - D) * scale;
            }
            accum_tile<HDP>(dq, Ks, tok<EXTRA>(t, 0), r, qp, ds);
        }
        store_rows<HDP>(dq, dqkv + ((long)b * N + query) * E3 + h * hd, qp, qok, hd);
        if (EXTRA) gq0 += tok0_dq_partial<HDP>(Qs, Ds, Ks, Vs, tok<EXTRA>(qt, 0), l0, D0, scale, lane, r, qp);
    }
    if (EXTRA && (HDP >= 64 || lane < HDP)) PA[wave * 3 * HDP + lane] = gq0;
    __syncthreads();                                                   // Es (D of every row) and the dQ partials complete
    ATTN_STAMP(2);

    // ---- phase 2: dK, dV
    float gk0 = 0.f, gv0 = 0.f;
    if (EXTRA && wave == 0)
        tok0_dq_combine<HDP>(Qs, Ds, Ks, Vs, PA, 3 * HDP, nwaves, l0, D0, scale, dqkv + (long)b * N * E3 + h * hd, hd, lane);
    for (int kt = wave; kt < ntile; kt += nwaves) {
        const int key = tok<EXTRA>(kt, r);
        const bool kok = key < N;
        float kf[NMM], vf[NMM];
        load_frag_lds<HDP>(kf, Ks, key, qp);                           // own rows from the staged slices (rows N .. nrows-1
        load_frag_lds<HDP>(vf, Vs, key, qp);                           // of the padded layout are staged as zeros)
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (EXTRA) {                                                   // token 0 as a query
            const float s0 = frag_dot_row<HDP>(kf, Qs, qp) * scale;
            const float dp0 = frag_dot_row<HDP>(vf, Ds, qp);
            const float p0 = __expf(s0 - Ls[0]);
            axpy_row<HDP>(dv, p0, Ds, qp);
            axpy_row<HDP>(dk, p0 * (dp0 - Es[0]) * scale, Qs, qp);
        }
        for (int t = 0; t < ntile; ++t) {
            f32x4 sc, dp;                                              // rows: queries of tile t, col: own key
            score_tile2<HDP>(Qs, tok<EXTRA>(t, 0), kf, Ds, tok<EXTRA>(t, 0), vf, r, qp, sc, dp);
            f32x4 p, ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int query = tok<EXTRA>(t, 4 * qp + e);
                p[e] = (EXTRA || (query < N && kok)) ? __expf(sc[e] * scale - Ls[query]) : 0.f;
                ds[e] = p[e] * (dp[e] - Es[query]) * scale;
            }
            accum_tile2<HDP>(dv, Ds, p, dk, Qs, ds, tok<EXTRA>(t, 0), r, qp);
        }
        float* drow = dqkv + ((long)b * N + key) * E3 + h * hd;
        store_rows<HDP>(dk, drow + E, qp, kok, hd);
        store_rows<HDP>(dv, drow + 2 * E, qp, kok, hd);
        if (EXTRA) tok0_dkv_partial<HDP>(Ks, Vs, Qs, Ds, Ls, Es, tok<EXTRA>(kt, 0), scale, lane, r, qp, gk0, gv0);
    }
    if (EXTRA) {
        if (HDP >= 64 || lane < HDP) {
            PA[wave * 3 * HDP + HDP + lane] = gk0;
            PA[wave * 3 * HDP + 2 * HDP + lane] = gv0;
        }
        __syncthreads();
        if (wave == 0) {
            float* drow = dqkv + (long)b * N * E3 + h * hd;
            tok0_dkv_combine<HDP>(Qs, Ds, Ks, Vs, PA + HDP, 3 * HDP, nwaves, Ls[0], Es[0], scale, drow + E, drow + 2 * E, hd,
                                  lane);
        }
    }
#ifdef VSOM_ATTN_REPEAT
    }
#endif
    ATTN_STAMP(3);
}

// ------------------------------------------------------------------ backward, fused, scores shared
// attn_bwd_fused_kernel computes the scores and dP twice: transposed per query tile for dQ, and again per key
// tile for dK / dV (7 N x N x hd products for the 5 the mathematics has), and at two workgroups per CU its
// compute phases keep the f32 matrix pipe ~72 % busy.  With ONE 16-token tile per wave (4 tiles, N <= 65) the wave
// keeps the P^T and dS^T blocks it formed for dQ in registers (32 of them); after the dQ phase the K and V regions
// are dead -- every wave first takes what the second phase still needs from them (its own k / v rows for the
// token-0 terms, rows 0 aside) -- and hold P and dS as [query][key] matrices, which the key waves read back
// transposed (conflict-free both ways at a row stride of 16 tiles + 4).  The dK / dV phase is then the two
// accumulations only: 320 MFMAs per wave instead of 448, and no second exp pass.  Needs 2 x (16 tiles)(16 tiles + 4)
// floats <= the K + V regions: hd = 64.  The values are the same bits as the recomputed ones (a product commutes and
// the reduction order over the head dim is the same), so the result is bit-identical to the other forms.
template <int HDP, bool EXTRA, bool BF16X3 = false>
__global__ __launch_bounds__(256) void attn_bwd_shared_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                              const float* __restrict__ dout, const float* __restrict__ lse,
                                                              float* __restrict__ dqkv, float* __restrict__ delta, int N,
                                                              int H, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NDT = ACfg<HDP>::NDT;
    constexpr int NMM = ACfg<HDP>::NMM;
    constexpr int S = ACfg<HDP>::S;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * hd, E3 = 3 * E;
    const int ntile = EXTRA ? (N - 1) >> 4 : (N + 15) >> 4;           // == number of waves, <= 4
    const int nrows = EXTRA ? N : ntile << 4;
    const int nrp = (nrows + 3) & ~3;
    const int PS = 16 * ntile + 4;                                     // row stride of the P / dS matrices
    float* Ks = smem;
    float* Vs = Ks + nrows * S;
    float* Qs = Vs + nrows * S;
    float* Ds = Qs + nrows * S;
    float* Ls = Ds + nrows * S;
    float* Es = Ls + nrp;
    float* PA = Es + nrp;                                              // [nwaves][3 HDP]: gq | gk | gv of token 0
    float* X2 = PA + (blockDim.x >> 6) * 3 * HDP;                      // k, v of token 0 once the K / V regions are reused
    float* X3 = X2 + HDP;
    float* Pm = Ks;                                                    // [16 ntile][PS] after the dQ phase
    float* Dm = Vs;
    const float* base = qkv + (long)b * N * E3 + h * hd;
    const long obase = (long)b * N * E + h * hd;
    const long srow0 = ((long)b * H + h) * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, qp = lane >> 4;
    float qf[NMM], dof[NMM], of[NMM];
    ATTN_STAMP(0);
    ATTN_STAMP_HWID();
    const int own = tok<EXTRA>(wave, r);                               // the wave's query row in phase 1, key row in phase 2
    const bool ook = own < N;
    load_frag<HDP>(qf, base + (long)own * E3, qp, ook, hd);
    load_frag<HDP>(dof, dout + obase + (long)own * E, qp, ook, hd);
    if constexpr (!(BF16X3 && HDP == 64)) load_frag<HDP>(of, out + obase + (long)own * E, qp, ook, hd);     // (the split form takes D from P and dP)
    float o0 = 0.f;
    if (EXTRA && lane < hd) o0 = out[obase + lane];
    float l_r = 0.f;
    if ((int)threadIdx.x < N) l_r = lse[srow0 + threadIdx.x];
    stage_rows_quad<HDP>(Ks, base + E, E3, Vs, base + 2 * E, E3, Qs, base, E3, Ds, dout + obase, E, N, nrows);
    for (int i = threadIdx.x; i < nrp; i += blockDim.x) {
        Ls[i] = (i == (int)threadIdx.x) ? l_r : ((i < N) ? lse[srow0 + i] : 0.f);
        Es[i] = 0.f;
    }
    __syncthreads();
    ATTN_STAMP(1);

    // ---- phase 1: dQ and D; P^T and dS^T of the wave's query tile stay in registers
    float D0 = 0.f, l0 = 0.f, gq0 = 0.f;
    if (EXTRA) {
        D0 = wave_sum64((HDP >= 64 || lane < HDP) ? Ds[lane] * o0 : 0.f);
        l0 = Ls[0];
        if (wave == 0 && lane == 0) { delta[srow0] = D0; Es[0] = D0; }
    }
    f32x4 pT[4], dsT[4];
    SplitFrag qs, dos;
    if constexpr (BF16X3 && HDP == 64) { split_frag(qf, qs); split_frag(dof, dos); }
    {
        const int query = own;
        const bool qok = ook;
        const float lq = qok ? Ls[query] : 0.f;
        f32x4 dq[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (BF16X3 && HDP == 64) {
            // D_i = dO_i . O_i = sum_j p_ij dP_ij: taken from the blocks the wave forms anyway, so the O rows are never read
            // (an eighth of the kernel's HBM traffic).  All tiles' P and dP first, then D, then dS and the accumulation.
            float s0 = 0.f, dp0 = 0.f, p0 = 0.f, dsum = 0.f;
            if (EXTRA) {                                               // token 0 as a key
                s0 = frag_dot_row<HDP>(qf, Ks, qp) * scale;
                dp0 = frag_dot_row<HDP>(dof, Vs, qp);
                p0 = __expf(s0 - lq);
                if (qp == 0) dsum = p0 * dp0;                          // (the four lane groups hold the same p0, dp0: counted once)
            }
            f32x4 dpT[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < ntile) {
                    float kf[16], vf[16];
                    load_frag_lds<HDP>(kf, Ks, tok<EXTRA>(t, 0) + r, qp);
                    load_frag_lds<HDP>(vf, Vs, tok<EXTRA>(t, 0) + r, qp);
                    SplitFrag ka, va;
                    split_frag(kf, ka);
                    split_frag(vf, va);
                    const f32x4 sc = score_x3(ka, qs);
                    dpT[t] = score_x3(va, dos);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = tok<EXTRA>(t, 4 * qp + e);
                        const float p = (EXTRA || (key < N && qok)) ? __expf(sc[e] * scale - lq) : 0.f;
                        pT[t][e] = p;
                        dsum = fmaf(p, dpT[t][e], dsum);
                    }
                }
            }
            const float D = group_sum(dsum);
            if (qp == 0 && qok) { delta[srow0 + query] = D; Es[query] = D; }
            if (EXTRA) axpy_row<HDP>(dq, p0 * (dp0 - D) * scale, Ks, qp);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < ntile) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dsT[t][e] = pT[t][e] * (dpT[t][e] - D) * scale;
                    // two tiles per 32-deep MFMA: a tile waits for its partner, an odd last tile takes the fp32 form
                    if (t & 1) accum_x3_pair(dq, Ks, tok<EXTRA>(t - 1, 0), tok<EXTRA>(t, 0), r, qp, dsT[t - 1], dsT[t]);
                    else if (t + 1 >= ntile) accum_tile<HDP>(dq, Ks, tok<EXTRA>(t, 0), r, qp, dsT[t]);
                }
            }
        } else {
            float D = 0.f;
#pragma unroll
            for (int mm = 0; mm < NMM; ++mm) D = fmaf(dof[mm], of[mm], D);
            D = group_sum(D);
            if (qp == 0 && qok) { delta[srow0 + query] = D; Es[query] = D; }
            if (EXTRA) {                                                   // token 0 as a key
                const float s0 = frag_dot_row<HDP>(qf, Ks, qp) * scale;
                const float dp0 = frag_dot_row<HDP>(dof, Vs, qp);
                const float p0 = __expf(s0 - lq);
                axpy_row<HDP>(dq, p0 * (dp0 - D) * scale, Ks, qp);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < ntile) {
                    f32x4 sc, dp;
                    score_tile2<HDP>(Ks, tok<EXTRA>(t, 0), qf, Vs, tok<EXTRA>(t, 0), dof, r, qp, sc, dp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = tok<EXTRA>(t, 4 * qp + e);
                        const float p = (EXTRA || (key < N && qok)) ? __expf(sc[e] * scale - lq) : 0.f;
                        pT[t][e] = p;
                        dsT[t][e] = p * (dp[e] - D) * scale;
                    }
                    accum_tile<HDP>(dq, Ks, tok<EXTRA>(t, 0), r, qp, dsT[t]);
                }
            }
        }
        store_rows<HDP>(dq, dqkv + ((long)b * N + query) * E3 + h * hd, qp, qok, hd);
        if (EXTRA) gq0 += tok0_dq_partial<HDP>(Qs, Ds, Ks, Vs, tok<EXTRA>(wave, 0), l0, D0, scale, lane, r, qp);
    }
    if (EXTRA && (HDP >= 64 || lane < HDP)) PA[wave * 3 * HDP + lane] = gq0;
    __syncthreads();                                                   // Es (D of every row) and the dQ partials complete
    ATTN_STAMP(2);

    // ---- between the phases: what phase 2 still needs from K and V, then P and dS take their place
    float gk0 = 0.f, gv0 = 0.f, p0k = 0.f, w0k = 0.f;
    if (EXTRA) {
        if (wave == 0)
            tok0_dq_combine<HDP>(Qs, Ds, Ks, Vs, PA, 3 * HDP, nwaves, l0, D0, scale, dqkv + (long)b * N * E3 + h * hd, hd, lane);
        float kf[NMM], vf[NMM];                                        // token 0 as a query against the wave's own keys
        load_frag_lds<HDP>(kf, Ks, own, qp);
        load_frag_lds<HDP>(vf, Vs, own, qp);
        const float s0 = frag_dot_row<HDP>(kf, Qs, qp) * scale;
        const float dp0 = frag_dot_row<HDP>(vf, Ds, qp);
        p0k = __expf(s0 - Ls[0]);
        w0k = p0k * (dp0 - Es[0]) * scale;
        if (wave == nwaves - 1 && (HDP >= 64 || lane < HDP)) { X2[lane] = Ks[lane]; X3[lane] = Vs[lane]; }
    }
    __syncthreads();                                                   // K and V are dead
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < ntile) {
            *reinterpret_cast<f32x4*>(Pm + (16 * wave + r) * PS + 16 * t + 4 * qp) = pT[t];
            *reinterpret_cast<f32x4*>(Dm + (16 * wave + r) * PS + 16 * t + 4 * qp) = dsT[t];
        }
    __syncthreads();

    // ---- phase 2: dK, dV from the stored P and dS (rows: queries of tile t, column: the wave's own key)
    {
        const int key = own;
        const bool kok = ook;
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (EXTRA) {                                                   // token 0 as a query
            axpy_row<HDP>(dv, p0k, Ds, qp);
            axpy_row<HDP>(dk, w0k, Qs, qp);
        }
        for (int t = 0; t < ntile; ++t) {
            f32x4 p, ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p[e] = Pm[(16 * t + 4 * qp + e) * PS + 16 * wave + r];
                ds[e] = Dm[(16 * t + 4 * qp + e) * PS + 16 * wave + r];
            }
            if constexpr (BF16X3 && HDP == 64) {
                if (t + 1 < ntile) {
                    f32x4 p1, ds1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        p1[e] = Pm[(16 * (t + 1) + 4 * qp + e) * PS + 16 * wave + r];
                        ds1[e] = Dm[(16 * (t + 1) + 4 * qp + e) * PS + 16 * wave + r];
                    }
                    accum_x3_pair(dv, Ds, tok<EXTRA>(t, 0), tok<EXTRA>(t + 1, 0), r, qp, p, p1);
                    accum_x3_pair(dk, Qs, tok<EXTRA>(t, 0), tok<EXTRA>(t + 1, 0), r, qp, ds, ds1);
                    ++t;
                    continue;
                }
            }
            accum_tile2<HDP>(dv, Ds, p, dk, Qs, ds, tok<EXTRA>(t, 0), r, qp);
        }
        float* drow = dqkv + ((long)b * N + key) * E3 + h * hd;
        store_rows<HDP>(dk, drow + E, qp, kok, hd);
        store_rows<HDP>(dv, drow + 2 * E, qp, kok, hd);
        if (EXTRA) tok0_dkv_partial<HDP>(X2, X3, Qs, Ds, Ls, Es, tok<EXTRA>(wave, 0), scale, lane, r, qp, gk0, gv0);
    }
    if (EXTRA) {
        if (HDP >= 64 || lane < HDP) {
            PA[wave * 3 * HDP + HDP + lane] = gk0;
            PA[wave * 3 * HDP + 2 * HDP + lane] = gv0;
        }
        __syncthreads();
        if (wave == 0) {
            float* drow = dqkv + (long)b * N * E3 + h * hd;
            tok0_dkv_combine<HDP>(Qs, Ds, X2, X3, PA + HDP, 3 * HDP, nwaves, Ls[0], Es[0], scale, drow + E, drow + 2 * E, hd, lane);
        }
    }
    ATTN_STAMP(3);
}

// ------------------------------------------------------------------ host side
// (A persistent variant -- workgroups looping over (image, head) items with register prefetch of
// the next item's rows -- was measured and rejected: the extra registers drop residency and N = 65 got
// slower, 42 -> 55 us per forward layer.)
static bool use_extra(int N) { return N >= 17 && (N % 16) == 1; }
static int attn_tiles(int N) { return use_extra(N) ? (N - 1) / 16 : cdiv(N, 16); }
static int attn_waves(int N) {          // every wave is an MFMA (tile) wave; 1..4 or 8 of them: a workgroup whose wave count
    const int ntile = attn_tiles(N);    // is not a multiple of 4 puts ceil(w/4) waves on the first SIMDs, and those SIMDs' register
    if (ntile >= 8) return 8;           // files then bound the residency of the CU (tools/occupancy_probe.hip: 5 waves of
    return ntile > 4 ? 4 : ntile;       // 124 VGPRs -> 2 workgroups per CU where 4 waves give 4)
}
static int attn_hdp(int hd) {
    if (hd == 16 || hd == 32 || hd == 64) return hd;
    if (hd >= 1 && hd <= 4) return 4;
    if (hd <= 8) return 8;
    return 0;
}
// paw = token-0 partial floats per wave: forward hdp + 2, dQ hdp, dK/dV 2 hdp
static int attn_stride(int hdp) { return hdp + 4; }        // ACfg<HDP>::S
static size_t attn_lds_bytes(int N, int hdp, bool with_stats, int paw) {
    const int nrows = use_extra(N) ? N : cdiv(N, 16) * 16;
    const int nrp = (nrows + 3) & ~3;
    return ((size_t)2 * nrows * attn_stride(hdp) + (with_stats ? 2 * nrp : 0) + 2 * hdp + (size_t)attn_waves(N) * paw) * sizeof(float);
}
static size_t attn_fused_lds_bytes(int N, int hdp) {
    const int nrows = use_extra(N) ? N : cdiv(N, 16) * 16;
    const int nrp = (nrows + 3) & ~3;
    return ((size_t)4 * nrows * attn_stride(hdp) + 2 * nrp + (size_t)attn_waves(N) * 3 * hdp) * sizeof(float);
}

template <int HDP, bool EXTRA>
static int launch_fwd_t(const float* qkv, float* out, float* lse, int B, int N, int H, int hd, hipStream_t st) {
    const size_t lds = attn_lds_bytes(N, HDP, false, HDP + 2);
    VSOM_LAUNCH((attn_fwd_kernel<HDP, EXTRA>), dim3(B * H), dim3(64 * attn_waves(N)), lds, st, qkv, out, lse, N, H, hd,
                       1.0f / sqrtf((float)hd));
    VSOM_LAUNCH_CHECK("attn_fwd_kernel");
}
template <int HDP, bool EXTRA>
static int launch_bwd_t(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                        float* delta, int B, int N, int H, int hd, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)hd);
    const dim3 block(64 * attn_waves(N));
    // all four slices in LDS and still two workgroups per CU -> one fused launch (vector path only)
    const size_t fused_lds = attn_fused_lds_bytes(N, HDP);
    const int mode = g_attn_fused.load(std::memory_order_relaxed);     // 0: two launches, 1: default, 2: fused with recomputed scores
    if constexpr (ACfg<HDP>::VEC) {
        const int nt = attn_tiles(N), nrows = use_extra(N) ? N : nt * 16;
        const size_t shared_lds = fused_lds + 2 * HDP * sizeof(float);
        if ((mode == 1 || mode == 3) && nt <= 4 && attn_waves(N) == nt && 16 * nt * (16 * nt + 4) <= nrows * (HDP + 4) && shared_lds <= 80 * 1024) {
            if constexpr (HDP == 64) {
                if (gemm_grad_products() == 3 && mode == 1) {   // the mode whose gradient GEMMs run on the two-piece split (hook 3: fp32 products)
                    VSOM_LAUNCH((attn_bwd_shared_kernel<HDP, EXTRA, true>), dim3(B * H), block, shared_lds, st, qkv, out, dout, lse,
                                dqkv, delta, N, H, hd, scale);
                    VSOM_LAUNCH_CHECK("attn_bwd_shared_kernel");
                }
            }
            VSOM_LAUNCH((attn_bwd_shared_kernel<HDP, EXTRA>), dim3(B * H), block, shared_lds, st, qkv, out, dout, lse, dqkv,
                               delta, N, H, hd, scale);
            VSOM_LAUNCH_CHECK("attn_bwd_shared_kernel");
        }
    }
    if (ACfg<HDP>::VEC && fused_lds <= 80 * 1024 && mode) {
        VSOM_LAUNCH((attn_bwd_fused_kernel<HDP, EXTRA>), dim3(B * H), block, fused_lds, st, qkv, out, dout, lse, dqkv,
                           delta, N, H, hd, scale);
        VSOM_LAUNCH_CHECK("attn_bwd_fused_kernel");
    }
    VSOM_LAUNCH((attn_bwd_dq_kernel<HDP, EXTRA>), dim3(B * H), block, attn_lds_bytes(N, HDP, false, HDP), st, qkv, out,
                       dout, lse, dqkv, delta, N, H, hd, scale);
    int rc = hip_status(hipGetLastError(), "attn_bwd_dq_kernel");
    if (rc) return rc;
    VSOM_LAUNCH((attn_bwd_dkv_kernel<HDP, EXTRA>), dim3(B * H), block, attn_lds_bytes(N, HDP, true, 2 * HDP), st, qkv,
                       dout, lse, delta, dqkv, N, H, hd, scale);
    VSOM_LAUNCH_CHECK("attn_bwd_dkv_kernel");
}
template <int HDP>
static int launch_fwd(const float* qkv, float* out, float* lse, int B, int N, int H, int hd, hipStream_t st) {
    return use_extra(N) ? launch_fwd_t<HDP, true>(qkv, out, lse, B, N, H, hd, st)
                        : launch_fwd_t<HDP, false>(qkv, out, lse, B, N, H, hd, st);
}
template <int HDP>
static int launch_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                      float* delta, int B, int N, int H, int hd, hipStream_t st) {
    return use_extra(N) ? launch_bwd_t<HDP, true>(qkv, out, dout, lse, dqkv, delta, B, N, H, hd, st)
                        : launch_bwd_t<HDP, false>(qkv, out, dout, lse, dqkv, delta, B, N, H, hd, st);
}


// ------------------------------------------------------------------ attention maps (return_attn=True, vit.py:33-34,41-42)
// probs[b,h,i,j] = exp(scale * q_i . k_j - lse[b,h,i]) from the qkv buffer and the log-sum-exp the forward saved: the
// softmax probabilities the fused kernels never write.  Visualisation path only (tools/evaluation.py); plain VALU.
__global__ __launch_bounds__(256) void attn_probs_kernel(const float* __restrict__ qkv, const float* __restrict__ lse,
                                                         float* __restrict__ probs, int N, int H, int hd, float scale) {
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int E3 = 3 * H * hd;
    const float* q0 = qkv + (long)b * N * E3 + h * hd;
    const float* k0 = q0 + H * hd;
    for (long idx = (long)blockIdx.y * 256 + threadIdx.x; idx < (long)N * N; idx += (long)gridDim.y * 256) {
        const int i = (int)(idx / N), j = (int)(idx % N);
        const float* q = q0 + (long)i * E3;
        const float* k = k0 + (long)j * E3;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(q[d], k[d], s);
        probs[((long)bh * N + i) * N + j] = __expf(s * scale - lse[(long)bh * N + i]);
    }
}

static int attn_check(const char* who, int B, int N, int H, int hd, int* hdp) {
    VSOM_REQUIRE(B > 0 && N > 0 && H > 0 && hd > 0, VSOM_EINVAL, "%s: bad shape B=%d N=%d H=%d hd=%d", who, B, N, H, hd);
    *hdp = attn_hdp(hd);
    VSOM_REQUIRE(*hdp != 0, VSOM_EUNSUPPORTED, "%s: head dim %d not supported (1..8, 16, 32, 64)", who, hd);
    VSOM_REQUIRE(attn_lds_bytes(N, *hdp, true, 2 * *hdp) <= 160 * 1024, VSOM_EUNSUPPORTED,
                 "%s: N=%d hd=%d needs %zu B of LDS (> 160 KiB)", who, N, hd, attn_lds_bytes(N, *hdp, true, 2 * *hdp));
    return VSOM_OK;
}

}  // namespace vsom

using namespace vsom;

extern "C" {

int vsom_set_attention_fused(int fused) {
    g_attn_fused.store(fused < 0 ? 0 : (fused > 3 ? 3 : fused), std::memory_order_relaxed);
    return VSOM_OK;
}

int vsom_attention_fwd(const float* qkv, float* out, float* lse, int B, int N, int H, int hd, vsom_stream_t stream) {
    VSOM_REQUIRE(qkv && out && lse, VSOM_EINVAL, "attention_fwd: null pointer");
    int hdp;
    int rc = attn_check("attention_fwd", B, N, H, hd, &hdp);
    if (rc) return rc;
    VSOM_REQUIRE(hdp % 16 != 0 || (aligned16(qkv) && aligned16(out)), VSOM_EALIGN, "attention_fwd: 16-byte alignment required");
    switch (hdp) {
        case 4: return launch_fwd<4>(qkv, out, lse, B, N, H, hd, stream);
        case 8: return launch_fwd<8>(qkv, out, lse, B, N, H, hd, stream);
        case 16: return launch_fwd<16>(qkv, out, lse, B, N, H, hd, stream);
        case 32: return launch_fwd<32>(qkv, out, lse, B, N, H, hd, stream);
        default: return launch_fwd<64>(qkv, out, lse, B, N, H, hd, stream);
    }
}

int vsom_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                       float* delta_ws, int B, int N, int H, int hd, vsom_stream_t stream) {
    VSOM_REQUIRE(qkv && out && dout && lse && dqkv && delta_ws, VSOM_EINVAL, "attention_bwd: null pointer");
    int hdp;
    int rc = attn_check("attention_bwd", B, N, H, hd, &hdp);
    if (rc) return rc;
    VSOM_REQUIRE(hdp % 16 != 0 || (aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv)), VSOM_EALIGN,
                 "attention_bwd: 16-byte alignment required");
    switch (hdp) {
        case 4: return launch_bwd<4>(qkv, out, dout, lse, dqkv, delta_ws, B, N, H, hd, stream);
        case 8: return launch_bwd<8>(qkv, out, dout, lse, dqkv, delta_ws, B, N, H, hd, stream);
        case 16: return launch_bwd<16>(qkv, out, dout, lse, dqkv, delta_ws, B, N, H, hd, stream);
        case 32: return launch_bwd<32>(qkv, out, dout, lse, dqkv, delta_ws, B, N, H, hd, stream);
        default: return launch_bwd<64>(qkv, out, dout, lse, dqkv, delta_ws, B, N, H, hd, stream);
    }
}

int vsom_attention_probs(const float* qkv, const float* lse, float* probs, int B, int N, int H, int hd, vsom_stream_t stream) {
    VSOM_REQUIRE(qkv && lse && probs, VSOM_EINVAL, "attention_probs: null pointer");
    VSOM_REQUIRE(B > 0 && N > 0 && H > 0 && hd > 0, VSOM_EINVAL, "attention_probs: bad shape B=%d N=%d H=%d hd=%d", B, N, H, hd);
    const int by = cdiv((long)N * N, 256) < 64 ? cdiv((long)N * N, 256) : 64;
    VSOM_LAUNCH(attn_probs_kernel, dim3(B * H, by), dim3(256), 0, stream, qkv, lse, probs, N, H, hd, 1.0f / sqrtf((float)hd));
    VSOM_LAUNCH_CHECK("attn_probs_kernel");
}

}  // extern "C"
