// CANDIDATE, not part of the library: vit_som_amd/csrc/attention.hip plus the "rolling" kernels
// (attn_fwd_roll_kernel / attn_bwd_roll_kernel: a workgroup stays alive over several (image, head) items and the
// next item's slices arrive by LDS-DMA in the regions the current item has finished with) and hand-pipelined LDS
// operand loads.  Bit-identical to the general kernels on every shape tried, measured with tools/attn_lab.hip
// (-DVSOM_ATTN_CANDIDATE), and NOT adopted: forward 31.3 us against 33.5-36, fused backward 85 against 87.5 at
// N = 65 -- a few per cent for a page of assembly-level hazards (profiles/r02_attention_lab_findings.txt has the
// numbers and the list of what the compiler does to an asynchronous load).  Kept as the starting point of any
// further work on overlapping the staging with the compute.
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

// The general and the rolling / fused kernels must give the same bits (the tests flip between them): no
// implicit mul+add contraction -- where the compiler fuses depends on the surrounding code, and did differ
// between the two forward kernels at hd = 32 by 1 ulp in 1 % of the outputs.  fmaf() where a fused op is meant.
#pragma clang fp contract(off)

namespace vsom {

// test / measurement hook (vsom_set_attention_fused): 0 = the general kernels only (forward one item per
// workgroup, backward as two launches); 1 = default (short sequences: rolling forward, fused backward);
// k >= 2 = the same with k items per workgroup in the rolling kernels
static std::atomic<int> g_attn_fused{1};

// tools/attn_lab.hip builds this file with VSOM_ATTN_STAMPS: thread 0 of every workgroup records the 100 MHz
// real-time counter at four points (+ the hardware id of its wave); the library build compiles none of it
#ifdef VSOM_ATTN_STAMPS
__device__ unsigned long long* g_attn_stamps = nullptr;          // [grid][16]
#define ATTN_STAMP(i)                                                                              \
    do {                                                                                           \
        if (threadIdx.x == 0 && g_attn_stamps) g_attn_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define ATTN_STAMP_ONCE(i)                                                                         \
    do {                                                                                           \
        if (threadIdx.x == 0 && g_attn_stamps && g_attn_stamps[blockIdx.x * 16 + (i)] == 0)        \
            g_attn_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                \
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
#define ATTN_STAMP_ONCE(i)
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

// The same two steps apart, for software pipelining by hand: the compiler emits "LDS read, wait, two MFMAs" per
// step otherwise, i.e. a full LDS latency in front of every pair of MFMAs (a lone wave runs the PV phase at
// 40 % of the MFMA rate).  The operands of a tile are requested a phase ahead and used later.
template <int HDP>
struct AccOp { float a[4 * ACfg<HDP>::NDT]; };
template <int HDP>
__device__ __forceinline__ void accum_load(AccOp<HDP>& v, const float* Zlds, int row0, int r, int qp) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dt = 0; dt < ACfg<HDP>::NDT; ++dt)
            v.a[s * ACfg<HDP>::NDT + dt] = (ACfg<HDP>::VEC || 16 * dt + r < HDP) ? Zlds[ACfg<HDP>::off(row0 + 4 * qp + s, 16 * dt + r)] : 0.f;
}
template <int HDP>
__device__ __forceinline__ void accum_mma(f32x4 (&o)[ACfg<HDP>::NDT], const AccOp<HDP>& v, f32x4 p) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dt = 0; dt < ACfg<HDP>::NDT; ++dt) o[dt] = mfma16(v.a[s * ACfg<HDP>::NDT + dt], p[s], o[dt]);
}
template <int HDP>
__device__ __forceinline__ void accum_mma2(f32x4 (&o0)[ACfg<HDP>::NDT], const AccOp<HDP>& v0, f32x4 p0, f32x4 (&o1)[ACfg<HDP>::NDT],
                                           const AccOp<HDP>& v1, f32x4 p1) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dt = 0; dt < ACfg<HDP>::NDT; ++dt) {
            o0[dt] = mfma16(v0.a[s * ACfg<HDP>::NDT + dt], p0[s], o0[dt]);
            o1[dt] = mfma16(v1.a[s * ACfg<HDP>::NDT + dt], p1[s], o1[dt]);
        }
}
template <int HDP>
struct ScoreOp { float a0[ACfg<HDP>::NMM], a1[ACfg<HDP>::NMM]; };
template <int HDP>
__device__ __forceinline__ void score_load2(ScoreOp<HDP>& v, const float* Y0, int row0, const float* Y1, int row1, int r, int qp) {
    load_frag_lds<HDP>(v.a0, Y0, row0 + r, qp);
    load_frag_lds<HDP>(v.a1, Y1, row1 + r, qp);
}
template <int HDP>
__device__ __forceinline__ void score_mma2(const ScoreOp<HDP>& v, const float (&b0)[ACfg<HDP>::NMM], const float (&b1)[ACfg<HDP>::NMM],
                                           f32x4& acc0, f32x4& acc1) {
    acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mm = 0; mm < ACfg<HDP>::NMM; ++mm) {
        acc0 = mfma16(v.a0[mm], b0[mm], acc0);
        acc1 = mfma16(v.a1[mm], b1[mm], acc1);
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
            AccOp<HDP> av[2];
            accum_load<HDP>(av[0], Vs, tok<EXTRA>(c0, 0), r, qp);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                if (tt < 3 && c0 + tt + 1 < ntile) accum_load<HDP>(av[(tt + 1) & 1], Vs, tok<EXTRA>(c0 + tt + 1, 0), r, qp);
                if (c0 + tt < ntile) accum_mma<HDP>(o, av[tt & 1], s[tt]);
            }
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

// ------------------------------------------------------------------ forward, short sequences, rolling
// At N <= 65 a workgroup's life is: stage K and V (a burst that all resident workgroups issue together, bound
// by HBM: ~7 us), then compute with the memory system idle (~6 us): the kernel costs the SUM of the two
// (tools/attn_lab.hip stamps).  This form keeps the workgroup alive over several (image, head) items and loads
// the next item's slices with LDS-DMA (global -> LDS, no registers, no ds_write) into the regions the current
// item has finished with: K after the score phase (it lands during softmax / PV), V after the PV phase (it
// lands during the next item's score phase).  Same arithmetic in the same order as attn_fwd_kernel (a
// single softmax chunk: at most 4 key tiles), so the two give identical bits.
//
// The loads of the loop are written in assembly on purpose.  The compiler's wait-count pass treats an LDS-DMA load
// as a possible writer of every LDS address (one dynamic LDS array: nothing to tell the regions apart) and puts
// s_waitcnt vmcnt(0) in front of the next LDS read, __syncthreads() carries a release fence that waits for
// vmcnt(0) too, and a wait it inserts for a register load counts only the memory operations it knows about --
// all three turn the asynchronous load into a synchronous one (measured: the next item's K was waited for
// inside the PV phase).  So: LDS-DMA and the query-fragment loads as asm (invisible to that pass), explicit
// s_waitcnt vmcnt(n) placed by hand (the memory pipeline retires in order, n = the instructions issued later
// that may still fly), and s_waitcnt lgkmcnt(0) + s_barrier as the workgroup barrier.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long a = (unsigned long)p;
    return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffff)),
                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}
// LDS byte address of a float inside the dynamic LDS array `smem` (which starts after the static LDS, none here).
// (Through offsets: casting the flat pointer back to the LDS address space trips the compiler's verifier.)
__device__ __forceinline__ unsigned lds_addr(const float* p, const float* smem_base) {
    return __builtin_amdgcn_groupstaticsize() + (unsigned)(p - smem_base) * 4u;
}
__device__ __forceinline__ void dma16(i32x4 rs, unsigned lds, unsigned voff) {      // LDS address = lds + 16 * lane
    const i32x4 u = {__builtin_amdgcn_readfirstlane(rs[0]), __builtin_amdgcn_readfirstlane(rs[1]),
                     __builtin_amdgcn_readfirstlane(rs[2]), __builtin_amdgcn_readfirstlane(rs[3])};
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(__builtin_amdgcn_readfirstlane((int)lds)), "v"(voff), "s"(u) : "memory");
}
// rows [0, N) of one slice: one LDS-DMA instruction per row, issued by the waves in turn, lanes 0 .. HDP/4-1
// move 16 bytes each (the padded rows are not contiguous, so a wave cannot move four rows with one instruction)
template <int HDP>
__device__ __forceinline__ void dma_rows(i32x4 rs, unsigned lds, unsigned byte0, unsigned row_bytes, int N, int wave,
                                         int nwaves, int lane) {
    if (lane < HDP / 4)
        for (int row = wave; row < N; row += nwaves)
            dma16(rs, lds + row * ACfg<HDP>::S * 4, byte0 + row * row_bytes + lane * 16);
}
// A whole padded slice image with full-wave instructions.  One DMA instruction per ROW (16 active lanes, 256 B)
// turned out to be bound by the instruction rate of the DMA path: two workgroups per CU issuing 68 of them per
// wave and item made every item 8 us longer.  A wave's instruction writes 64 consecutive 16-byte units of LDS, so
// it is aimed at the padded image itself, 1 KB at a time: unit p of the image is chunk p % (C4 + 1) of row
// p / (C4 + 1) (the last chunk of every row is the padding: its lane reads past the buffer and writes a zero).
// The per-lane source offsets depend on the wave and the lane only: computed once, kept in registers; the slice
// (q / k / v, head) enters through the scalar offset of the instruction.
constexpr int ROLL_PIECES = 5;          // 1 KB pieces per wave: (16 waves + 1) rows x 17 units <= 64 x 5 x waves
struct ImgPlan { unsigned v[ROLL_PIECES]; };
template <int HDP>
__device__ __forceinline__ void img_plan(ImgPlan& pl, unsigned row_bytes, int N, int wave, int nwaves, int lane) {
    constexpr int CPR = HDP / 4 + 1;                                   // 16-byte units per padded row
#pragma unroll
    for (int k = 0; k < ROLL_PIECES; ++k) {
        const int p = 64 * (wave + k * nwaves) + lane, row = p / CPR, ch = p % CPR;
        pl.v[k] = (row < N && ch < HDP / 4) ? (unsigned)row * row_bytes + (unsigned)ch * 16u : 0x7FFFFFF0u;
    }
}
__device__ __forceinline__ void dma16s(i32x4 rs, unsigned lds, unsigned voff, unsigned soff) {
    const i32x4 u = {__builtin_amdgcn_readfirstlane(rs[0]), __builtin_amdgcn_readfirstlane(rs[1]),
                     __builtin_amdgcn_readfirstlane(rs[2]), __builtin_amdgcn_readfirstlane(rs[3])};
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(__builtin_amdgcn_readfirstlane((int)lds)), "v"(voff), "s"(u), "s"(__builtin_amdgcn_readfirstlane((int)soff)) : "memory");
}
// exactly ROLL_PIECES instructions per wave (a piece past the image repeats the wave's first piece: the same bytes
// to the same place), so that s_waitcnt vmcnt(ROLL_PIECES) means "everything older than this image has landed"
template <int HDP>
__device__ __forceinline__ void dma_image(i32x4 rs, unsigned lds, const ImgPlan& pl, unsigned byte0, int N, int wave, int nwaves,
                                          int lane) {
    constexpr int CPR = HDP / 4 + 1;
    const int total = N * CPR;
#pragma unroll
    for (int k = 0; k < ROLL_PIECES; ++k) {
        const bool in = 64 * (wave + k * nwaves) < total;              // wave-uniform
        const int i = in ? wave + k * nwaves : wave;
        const unsigned vo = in ? pl.v[k] : pl.v[0];
        if (64 * i + lane < total) dma16s(rs, lds + 1024u * i, vo, byte0);
    }
}
// The query rows of the NEXT item travel while the current one is computed, and they go through LDS as well.
// (A load into registers cannot be used for that: the compiler is free to copy "loaded" registers around before
// a hand-placed wait -- it did, at hd = 32 --, its own wait for a load it can see counts only the younger
// operations it knows of, which forces the younger LDS-DMA loads to land too, the pipeline retiring in order;
// and accumulation registers named in asm are not safe either, the allocator spills into them between the
// statements.)  Their LDS image is unpadded and XOR-swizzled, 16 rows per tile wave: one full-wave DMA
// instruction moves 1 KB, and the one fragment read per item (16 rows x one 16-byte chunk) is conflict-free.
template <int HDP>
__device__ __forceinline__ int q_swz(int row) { return (row / (64 / HDP)) & (HDP / 4 - 1); }
// rows [0, 16 ntile) of the image <- tokens tok0 + row of the q slice; rows past the tensor read as zeros (the
// buffer's range check)
template <int HDP>
__device__ __forceinline__ void dma_q(i32x4 rs, unsigned lds, unsigned byte0, unsigned row_bytes, int ntile, int tok0, int wave,
                                      int nwaves, int lane) {
    constexpr int C4 = HDP / 4;
    const int ninst = ntile * C4 / 4;                                  // 16 ntile rows x C4 chunks, 64 chunks per instruction
    for (int i = wave; i < ninst; i += nwaves) {
        const int p = 64 * i + lane, row = p / C4, pc = p % C4;
        dma16(rs, lds + 1024u * i, byte0 + (unsigned)(tok0 + row) * row_bytes + (unsigned)((pc ^ q_swz<HDP>(row)) * 16));
    }
}
template <int HDP>
__device__ __forceinline__ void load_frag_q(float (&f)[HDP / 4], const float* Qs, int row, int qp) {
#pragma unroll
    for (int g = 0; g < HDP / 16; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(Qs + row * HDP + (((4 * g + qp) ^ q_swz<HDP>(row)) << 2));
#pragma unroll
        for (int e = 0; e < 4; ++e) f[4 * g + e] = v[e];
    }
}
__device__ __forceinline__ void dma_wait_older() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }    // ROLL_PIECES
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// at most KEEP memory operations may still be in flight: the KEEP youngest, the pipeline retires in order (used
// where the youngest are this wave's own output stores, whose acknowledgement nobody has to wait for)
template <int KEEP>
__device__ __forceinline__ void dma_wait_keep() {
    static_assert(KEEP == 2 || KEEP == 3 || KEEP == 4 || KEEP == 5 || KEEP == 8, "add the immediate");
    if constexpr (KEEP == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (KEEP == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (KEEP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (KEEP == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if constexpr (KEEP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int HDP, bool EXTRA>
__global__ __launch_bounds__(256, 3) void attn_fwd_roll_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                               float* __restrict__ lse, int N, int H, int hd, float scale,
                                                               int nitems) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NDT = ACfg<HDP>::NDT;
    constexpr int NMM = ACfg<HDP>::NMM;
    constexpr int PAW = HDP + 2;
    const int E = H * hd, E3 = 3 * E;
    const Carve<HDP, EXTRA> cv(smem, N, false);
    const int ntile = cv.ntile;                                        // == number of waves, <= 4
    float* Ks = cv.Y0;
    float* Vs = cv.Y1;
    // the wave index as a scalar: everything the DMA issue derives from it (rows, LDS addresses) stays in SGPRs
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int r = lane & 15, qp = lane >> 4;
    const unsigned img_bytes = (unsigned)N * E3 * 4, row_bytes = (unsigned)E3 * 4;
    // rows N .. nrows-1 of the padded layout stay zero for the whole kernel (the DMA never touches them)
    for (int idx = threadIdx.x; idx < (cv.nrows - N) * ACfg<HDP>::S; idx += blockDim.x) {
        Ks[N * ACfg<HDP>::S + idx] = 0.f;
        Vs[N * ACfg<HDP>::S + idx] = 0.f;
    }
    int item = blockIdx.x;
    ATTN_STAMP(0);
    ATTN_STAMP_HWID();
    float* Qs = cv.PA + ((nwaves * PAW + 3) & ~3);                     // [16 ntile][HDP], swizzled: the tile waves' query rows
    const unsigned ks_a = lds_addr(Ks, smem), vs_a = lds_addr(Vs, smem), x0_a = lds_addr(cv.X0, smem), qs_a = lds_addr(Qs, smem);
    ImgPlan plan;
    img_plan<HDP>(plan, row_bytes, N, wave, nwaves, lane);
    const int query = tok<EXTRA>(wave, r);
    const bool qok = query < N;
    {
        const int b = item / H, h = item % H;
        const i32x4 rs = make_rsrc(qkv + (long)b * N * E3, img_bytes);
        dma_image<HDP>(rs, ks_a, plan, (unsigned)(E + h * hd) * 4, N, wave, nwaves, lane);
        dma_image<HDP>(rs, vs_a, plan, (unsigned)(2 * E + h * hd) * 4, N, wave, nwaves, lane);
        if (EXTRA && wave == 0) dma_rows<HDP>(rs, x0_a, (unsigned)(h * hd) * 4, row_bytes, 1, 0, 1, lane);
        dma_q<HDP>(rs, qs_a, (unsigned)(h * hd) * 4, row_bytes, ntile, EXTRA ? 1 : 0, wave, nwaves, lane);
    }
    for (;;) {
        const int b = item / H, h = item % H;
        const int next = item + gridDim.x;
        const bool has_next = next < nitems;
        const int nb = next / H, nh = next % H;
        const i32x4 nrs = make_rsrc(qkv + (long)(has_next ? nb : b) * N * E3, img_bytes);
        // K, the query rows and q of token 0 have landed; V (issued last, ROLL_PIECES instructions) may still fly
        if (item == (int)blockIdx.x) dma_wait(); else dma_wait_older();
        if (item != (int)blockIdx.x) ATTN_STAMP_ONCE(12);
        lds_barrier();
        ATTN_STAMP(1);
        ATTN_STAMP_ONCE(6);
        float qf[NMM];
        load_frag_q<HDP>(qf, Qs, 16 * wave + r, qp);

        // ---- score phase (K)
        float m = -INFINITY, l = 0.f;
        if (EXTRA) {
            m = frag_dot_row<HDP>(qf, Ks, qp) * scale;
            l = 1.0f;
        }
        f32x4 s[4];
        float cmax = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < 4; tt += 2) {
            if (tt + 1 < ntile) score_tile2<HDP>(Ks, tok<EXTRA>(tt, 0), qf, Ks, tok<EXTRA>(tt + 1, 0), qf, r, qp, s[tt], s[tt + 1]);
            else if (tt < ntile) s[tt] = score_tile<HDP>(Ks, tok<EXTRA>(tt, 0), r, qp, qf);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (tt + u < ntile) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = tok<EXTRA>(tt + u, 4 * qp + e);
                        s[tt + u][e] = (EXTRA || key < N) ? s[tt + u][e] * scale : -INFINITY;
                        cmax = fmaxf(cmax, s[tt + u][e]);
                    }
                } else {
                    s[tt + u] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
            }
        }
        float sc0 = 0.f, s00 = 0.f;
        if (EXTRA) {
            sc0 = tile_rows_dot<HDP>(cv.X0, Ks, tok<EXTRA>(wave, 0), r, qp) * scale;      // token 0 as a query, own 16 keys
            if (wave == 0) s00 = vec_dot<HDP>(cv.X0, Ks, lane) * scale;
        }
        dma_wait();                                                    // this wave's rows of V have landed
        lds_barrier();                                                 // K, q0 and the query image are free, V is complete
        ATTN_STAMP_ONCE(7);
        if (has_next) {                                                // the next item's K, q0 and query rows: in flight during PV
            dma_image<HDP>(nrs, ks_a, plan, (unsigned)(E + nh * hd) * 4, N, wave, nwaves, lane);
            if (EXTRA && wave == 0) dma_rows<HDP>(nrs, x0_a, (unsigned)(nh * hd) * 4, row_bytes, 1, 0, 1, lane);
            dma_q<HDP>(nrs, qs_a, (unsigned)(nh * hd) * 4, row_bytes, ntile, EXTRA ? 1 : 0, wave, nwaves, lane);
        }

        // ---- softmax + PV phase (V)
        f32x4 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EXTRA) axpy_row<HDP>(o, 1.0f, Vs, qp);
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
        {
            AccOp<HDP> av[2];
            accum_load<HDP>(av[0], Vs, tok<EXTRA>(0, 0), r, qp);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                if (tt < 3 && tt + 1 < ntile) accum_load<HDP>(av[(tt + 1) & 1], Vs, tok<EXTRA>(tt + 1, 0), r, qp);
                if (tt < ntile) accum_mma<HDP>(o, av[tt & 1], s[tt]);
            }
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] *= inv;
        ATTN_STAMP(2);
        ATTN_STAMP_ONCE(8);
        store_rows<HDP>(o, out + ((long)b * N + query) * E + h * hd, qp, qok, hd);
        if (qp == 0 && qok) lse[((long)b * H + h) * N + query] = m + logf(l);

        float v0 = 0.f;
        if (EXTRA) {
            __builtin_amdgcn_sched_barrier(0);
            const int row0 = tok<EXTRA>(wave, 0);
            const float xm = tile_max(sc0);                            // one tile per wave: the running state starts here
            const float p = __expf(sc0 - xm);
            const float xl = tile_sum(p);
            const float xo = tile_wsum<HDP>(p, Vs, row0, lane);
            float* pa = cv.PA + wave * PAW;
            if (HDP >= 64 || lane < HDP) pa[lane] = xo;
            if (lane == 0) { pa[HDP] = xm; pa[HDP + 1] = xl; }
            if (wave == 0 && (HDP >= 64 || lane < HDP)) v0 = Vs[lane];
        }
        ATTN_STAMP_ONCE(9);
        lds_barrier();                                                 // V is free, the token-0 partials are complete
        ATTN_STAMP_ONCE(10);
        if (has_next) dma_image<HDP>(nrs, vs_a, plan, (unsigned)(2 * E + nh * hd) * 4, N, wave, nwaves, lane);
        if (EXTRA && wave == 0) {                                      // key 0 itself, then the waves in order
            float mm = s00;
            for (int w = 0; w < nwaves; ++w) mm = fmaxf(mm, cv.PA[w * PAW + HDP]);
            float ll = __expf(s00 - mm);
            float oo = ll * v0;
            for (int w = 0; w < nwaves; ++w) {
                const float a = __expf(cv.PA[w * PAW + HDP] - mm);
                ll = fmaf(cv.PA[w * PAW + HDP + 1], a, ll);
                if (HDP >= 64 || lane < HDP) oo = fmaf(cv.PA[w * PAW + lane], a, oo);
            }
            if (lane < hd) out[((long)b * N) * E + h * hd + lane] = oo / ll;
            if (lane == 0) lse[((long)b * H + h) * N] = mm + logf(ll);
        }
        if (!has_next) break;
        item = next;
        ATTN_STAMP_ONCE(11);
    }
    ATTN_STAMP(3);
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
        ScoreOp<HDP> sv;
        score_load2<HDP>(sv, Ks, tok<EXTRA>(0, 0), Vs, tok<EXTRA>(0, 0), r, qp);
        for (int t = 0; t < ntile; ++t) {
            AccOp<HDP> av;                                             // lands during the score MFMAs
            accum_load<HDP>(av, Ks, tok<EXTRA>(t, 0), r, qp);
            f32x4 s, dp;
            score_mma2<HDP>(sv, qf, dof, s, dp);
            if (t + 1 < ntile) score_load2<HDP>(sv, Ks, tok<EXTRA>(t + 1, 0), Vs, tok<EXTRA>(t + 1, 0), r, qp);   // during exp + dQ MFMAs
            f32x4 ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = tok<EXTRA>(t, 4 * qp + e);
                const float p = (EXTRA || (key < N && qok)) ? __expf(s[e] * scale - lq) : 0.f;
                ds[e] = p * (dp[e] - D) * scale;
            }
            accum_mma<HDP>(dq, av, ds);
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
            ScoreOp<HDP> sv;
            score_load2<HDP>(sv, Qs, tok<EXTRA>(t, 0), Ds, tok<EXTRA>(t, 0), r, qp);
            AccOp<HDP> avd, avq;                                       // land during the score MFMAs
            accum_load<HDP>(avd, Ds, tok<EXTRA>(t, 0), r, qp);
            accum_load<HDP>(avq, Qs, tok<EXTRA>(t, 0), r, qp);
            f32x4 s, dp;                                              // rows: queries of tile t, col: own key
            score_mma2<HDP>(sv, kf, vf, s, dp);
            f32x4 p, ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int query = tok<EXTRA>(t, 4 * qp + e);
                p[e] = (EXTRA || (query < N && kok)) ? __expf(s[e] * scale - Ls[query]) : 0.f;
                ds[e] = p[e] * (dp[e] - Es[query]) * scale;
            }
            accum_mma2<HDP>(dv, avd, p, dk, avq, ds);
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
        ScoreOp<HDP> sv;
        score_load2<HDP>(sv, Ks, tok<EXTRA>(0, 0), Vs, tok<EXTRA>(0, 0), r, qp);
        for (int t = 0; t < ntile; ++t) {
            AccOp<HDP> av;                                             // lands during the score MFMAs
            accum_load<HDP>(av, Ks, tok<EXTRA>(t, 0), r, qp);
            f32x4 sc, dp;
            score_mma2<HDP>(sv, qf, dof, sc, dp);
            if (t + 1 < ntile) score_load2<HDP>(sv, Ks, tok<EXTRA>(t + 1, 0), Vs, tok<EXTRA>(t + 1, 0), r, qp);   // during exp + dQ MFMAs
            f32x4 ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = tok<EXTRA>(t, 4 * qp + e);
                const float p = (EXTRA || (key < N && qok)) ? __expf(sc[e] * scale - lq) : 0.f;
                ds[e] = p * (dp[e] - D) * scale;
            }
            accum_mma<HDP>(dq, av, ds);
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
            ScoreOp<HDP> sv;
            score_load2<HDP>(sv, Qs, tok<EXTRA>(t, 0), Ds, tok<EXTRA>(t, 0), r, qp);
            AccOp<HDP> avd, avq;                                       // land during the score MFMAs
            accum_load<HDP>(avd, Ds, tok<EXTRA>(t, 0), r, qp);
            accum_load<HDP>(avq, Qs, tok<EXTRA>(t, 0), r, qp);
            f32x4 sc, dp;                                              // rows: queries of tile t, col: own key
            score_mma2<HDP>(sv, kf, vf, sc, dp);
            f32x4 p, ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int query = tok<EXTRA>(t, 4 * qp + e);
                p[e] = (EXTRA || (query < N && kok)) ? __expf(sc[e] * scale - Ls[query]) : 0.f;
                ds[e] = p[e] * (dp[e] - Es[query]) * scale;
            }
            accum_mma2<HDP>(dv, avd, p, dk, avq, ds);
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

// ------------------------------------------------------------------ backward, fused, rolling
// attn_bwd_fused_kernel kept alive over several items, the next item's slices arriving by LDS-DMA in the
// regions the current item has finished with: the fused kernel holds two workgroups per CU and each spends
// 7 of its 26 us staging (HBM-bound burst) with the matrix cores idle; the compute of three items behind one
// staging measured 64 us against 87.5 (tools/attn_lab.hip, VSOM_ATTN_REPEAT).  Per item:
//   barrier A   K, V, the token-0 rows q0 / dO0 and the log-sum-exps are in LDS, the wave's own q / dO / O
//               fragments in registers;  -> DMA Q, dO into their regions (nothing reads them in phase 1)
//   phase 1     dQ and D (as attn_bwd_fused_kernel)
//   barrier B   Q, dO landed; D of every row, the dQ partials of token 0
//   (wave 0 finishes token 0's dQ; every wave takes its own k / v rows; rows 0 of K, V are copied aside)
//   barrier C   K, V are free  -> DMA the NEXT item's K, V, q0, dO0, log-sum-exps; request its q / dO / O
//               fragments (ordinary loads: they are issued AFTER those DMAs and BEFORE the next item's, so the
//               compiler's wait for them is exactly "everything of the next item has landed")
//   phase 2     dK, dV
//   barrier D   token 0's dK / dV partials; wave 0 finishes them
// Same arithmetic in the same order as the general kernels: identical bits.
// make the compiler finish the loads behind these registers HERE (its own wait goes in front of this statement)
template <int NV>
__device__ __forceinline__ void touch(float (&f)[NV]) {
    static_assert(NV == 4 || NV == 8 || NV == 16, "fragment sizes");
    if constexpr (NV == 16)
        asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]),
                          "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]));
    else if constexpr (NV == 8)
        asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
    else
        asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
}
__device__ __forceinline__ void dma4(i32x4 rs, unsigned lds, unsigned voff) {       // one float per lane: LDS address = lds + 4 * lane
    const i32x4 u = {__builtin_amdgcn_readfirstlane(rs[0]), __builtin_amdgcn_readfirstlane(rs[1]),
                     __builtin_amdgcn_readfirstlane(rs[2]), __builtin_amdgcn_readfirstlane(rs[3])};
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :: "s"(__builtin_amdgcn_readfirstlane((int)lds)), "v"(voff), "s"(u) : "memory");
}
template <int HDP, bool EXTRA>
__global__ __launch_bounds__(256) void attn_bwd_roll_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                            const float* __restrict__ dout, const float* __restrict__ lse,
                                                            float* __restrict__ dqkv, float* __restrict__ delta, int N, int H,
                                                            int hd, float scale, int nitems, unsigned lse_bytes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NDT = ACfg<HDP>::NDT;
    constexpr int NMM = ACfg<HDP>::NMM;
    constexpr int S = ACfg<HDP>::S;
    const int E = H * hd, E3 = 3 * E;
    const int ntile = EXTRA ? (N - 1) >> 4 : (N + 15) >> 4;           // == number of waves, <= 4
    const int nrows = EXTRA ? N : ntile << 4;
    const int nrp = (nrows + 3) & ~3;
    float* Ks = smem;
    float* Vs = Ks + nrows * S;
    float* Qs = Vs + nrows * S;
    float* Ds = Qs + nrows * S;
    float* Lb = Ds + nrows * S;                                        // [2][nrp] log-sum-exps, by item parity
    float* Es = Lb + 2 * nrp;
    float* PA = Es + nrp;                                              // [nwaves][3 HDP]: gq | gk | gv of token 0
    float* X0 = PA + (blockDim.x >> 6) * 3 * HDP;                      // q, dO of token 0 (phase 1); k, v of token 0 (phase 2)
    float* X1 = X0 + HDP;
    float* X2 = X1 + HDP;
    float* X3 = X2 + HDP;
    const unsigned ks_a = lds_addr(Ks, smem), vs_a = lds_addr(Vs, smem), qs_a = lds_addr(Qs, smem), ds_a = lds_addr(Ds, smem),
                   lb_a = lds_addr(Lb, smem), x0_a = lds_addr(X0, smem), x1_a = lds_addr(X1, smem);
    // the wave index as a scalar: everything the DMA issue derives from it (rows, LDS addresses) stays in SGPRs
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int r = lane & 15, qp = lane >> 4;
    const unsigned img_bytes = (unsigned)N * E3 * 4, row_bytes = (unsigned)E3 * 4;
    const unsigned oimg_bytes = (unsigned)N * E * 4, orow_bytes = (unsigned)E * 4;
    const int own = tok<EXTRA>(wave, r);                               // the wave's own query / key row
    const bool ook = own < N;
    // the padding rows of the slices, of the statistics, and D stay zero where no item writes them
    for (int idx = threadIdx.x; idx < (nrows - N) * S; idx += blockDim.x) {
        Ks[N * S + idx] = 0.f; Vs[N * S + idx] = 0.f; Qs[N * S + idx] = 0.f; Ds[N * S + idx] = 0.f;
    }
    for (int i = threadIdx.x; i < nrp; i += blockDim.x) { Lb[i] = 0.f; Lb[nrp + i] = 0.f; Es[i] = 0.f; }
    lds_barrier();                                                     // before the DMA writes next to them

    ImgPlan plan3, plan1;                                              // rows of qkv (stride 3E) and of dout (stride E)
    img_plan<HDP>(plan3, row_bytes, N, wave, nwaves, lane);
    img_plan<HDP>(plan1, orow_bytes, N, wave, nwaves, lane);
    // next item's K, V, token-0 rows and log-sum-exps by DMA; its own fragments by ordinary loads AFTER them
    float qf[NMM], dof[NMM], of[NMM];
    float o0 = 0.f;
    auto request = [&](int it, int parity) {
        const int b = it / H, h = it % H;
        const i32x4 rq = make_rsrc(qkv + (long)b * N * E3, img_bytes);
        const i32x4 ro = make_rsrc(dout + (long)b * N * E, oimg_bytes);
        const i32x4 rl = make_rsrc(lse, lse_bytes);
        dma_image<HDP>(rq, ks_a, plan3, (unsigned)(E + h * hd) * 4, N, wave, nwaves, lane);
        dma_image<HDP>(rq, vs_a, plan3, (unsigned)(2 * E + h * hd) * 4, N, wave, nwaves, lane);
        if (EXTRA && wave == 0) dma_rows<HDP>(rq, x0_a, (unsigned)(h * hd) * 4, row_bytes, 1, 0, 1, lane);
        if (EXTRA && wave == 1 % nwaves) dma_rows<HDP>(ro, x1_a, (unsigned)(h * hd) * 4, orow_bytes, 1, 0, 1, lane);
        if (wave == 2 % nwaves)
            for (int i0 = 0; i0 < N; i0 += 64)
                if (i0 + lane < N) dma4(rl, lb_a + (unsigned)(parity * nrp + i0) * 4, (unsigned)(((long)b * H + h) * N + i0 + lane) * 4);
        const float* base = qkv + (long)b * N * E3 + h * hd;
        const long obase = (long)b * N * E + h * hd;
        // straight-line loads (a predicated load is a branch and a merge, and the merge drags the wait up to the
        // load): rows past N read row 0 and are zeroed after the wait; hd == HDP on this path
        const int orow = ook ? own : 0;
        load_frag<HDP>(qf, base + (long)orow * E3, qp, true, HDP);
        load_frag<HDP>(dof, dout + obase + (long)orow * E, qp, true, HDP);
        load_frag<HDP>(of, out + obase + (long)orow * E, qp, true, HDP);
        if (EXTRA) o0 = out[obase + (lane < HDP ? lane : 0)];
    };
    int item = blockIdx.x, parity = 0;
    ATTN_STAMP(0);
    ATTN_STAMP_HWID();
    request(item, 0);
    for (;;) {
        const int b = item / H, h = item % H;
        const int next = item + gridDim.x;
        const bool has_next = next < nitems;
        const long srow0 = ((long)b * H + h) * N;
        float* Ls = Lb + parity * nrp;
        // this wave's share of K, V, q0, dO0 and the log-sum-exps has landed; its dK / dV stores (2 NDT, the
        // youngest) need not have
        if (item == (int)blockIdx.x) dma_wait(); else dma_wait_keep<2 * NDT>();
        // the wave's own fragments: the compiler's wait for them must sit HERE, in front of the DMA issue below (left
        // to the first use it lands inside phase 1 as vmcnt(0) and takes this item's Q / dO loads with it).  It is a
        // vmcnt(0): the loop's two entries merge to the conservative count, so the dK / dV stores are waited for too;
        // moving the statement to the end of the loop body made the compiler put a vmcnt(0) back into phase 1.
        touch<NMM>(qf); touch<NMM>(dof); touch<NMM>(of);
        asm volatile("" : "+v"(o0));
        if (!EXTRA) {
#pragma unroll
            for (int mm = 0; mm < NMM; ++mm) { qf[mm] = ook ? qf[mm] : 0.f; dof[mm] = ook ? dof[mm] : 0.f; of[mm] = ook ? of[mm] : 0.f; }
        }
        lds_barrier();                                                 // ---- A
        ATTN_STAMP(1);
        {   // Q, dO of THIS item travel during phase 1
            const i32x4 rq = make_rsrc(qkv + (long)b * N * E3, img_bytes);
            const i32x4 ro = make_rsrc(dout + (long)b * N * E, oimg_bytes);
            dma_image<HDP>(rq, qs_a, plan3, (unsigned)(h * hd) * 4, N, wave, nwaves, lane);
            dma_image<HDP>(ro, ds_a, plan1, (unsigned)(h * hd) * 4, N, wave, nwaves, lane);
        }

        // ---- phase 1: dQ and D
        float D0 = 0.f, l0 = 0.f, gq0 = 0.f;
        if (EXTRA) {
            D0 = wave_sum64((HDP >= 64 || lane < HDP) ? X1[lane] * o0 : 0.f);
            l0 = Ls[0];
            if (wave == 0 && lane == 0) { delta[srow0] = D0; Es[0] = D0; }
        }
        {
            const int query = own;
            const bool qok = ook;
            float D = 0.f;
#pragma unroll
            for (int mm = 0; mm < NMM; ++mm) D = fmaf(dof[mm], of[mm], D);
            D = group_sum(D);
            if (qp == 0 && qok) { delta[srow0 + query] = D; Es[query] = D; }
            const float lq = qok ? Ls[query] : 0.f;
            f32x4 dq[NDT];
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (EXTRA) {                                               // token 0 as a key
                const float s0 = frag_dot_row<HDP>(qf, Ks, qp) * scale;
                const float dp0 = frag_dot_row<HDP>(dof, Vs, qp);
                const float p0 = __expf(s0 - lq);
                axpy_row<HDP>(dq, p0 * (dp0 - D) * scale, Ks, qp);
            }
            ScoreOp<HDP> sv;
            score_load2<HDP>(sv, Ks, tok<EXTRA>(0, 0), Vs, tok<EXTRA>(0, 0), r, qp);
            for (int t = 0; t < ntile; ++t) {
                AccOp<HDP> av;
                accum_load<HDP>(av, Ks, tok<EXTRA>(t, 0), r, qp);
                f32x4 sc, dp;
                score_mma2<HDP>(sv, qf, dof, sc, dp);
                if (t + 1 < ntile) score_load2<HDP>(sv, Ks, tok<EXTRA>(t + 1, 0), Vs, tok<EXTRA>(t + 1, 0), r, qp);
                f32x4 ds;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = tok<EXTRA>(t, 4 * qp + e);
                    const float p = (EXTRA || (key < N && qok)) ? __expf(sc[e] * scale - lq) : 0.f;
                    ds[e] = p * (dp[e] - D) * scale;
                }
                accum_mma<HDP>(dq, av, ds);
            }
            store_rows<HDP>(dq, dqkv + ((long)b * N + query) * E3 + h * hd, qp, qok, hd);
            if (EXTRA) gq0 += tok0_dq_partial<HDP>(X0, X1, Ks, Vs, tok<EXTRA>(wave, 0), l0, D0, scale, lane, r, qp);
        }
        if (EXTRA && (HDP >= 64 || lane < HDP)) PA[wave * 3 * HDP + lane] = gq0;
        dma_wait_keep<NDT + 1>();                                      // this wave's rows of Q, dO (older than its D and dQ stores)
        lds_barrier();                                                 // ---- B
        ATTN_STAMP(2);

        // ---- between the phases: token 0's dQ, the wave's own k / v rows, rows 0 of K and V aside
        float gk0 = 0.f, gv0 = 0.f;
        if (EXTRA && wave == 0)
            tok0_dq_combine<HDP>(X0, X1, Ks, Vs, PA, 3 * HDP, nwaves, l0, D0, scale, dqkv + (long)b * N * E3 + h * hd, hd, lane);
        const int key = own;
        const bool kok = ook;
        float kf[NMM], vf[NMM];
        load_frag_lds<HDP>(kf, Ks, key, qp);
        load_frag_lds<HDP>(vf, Vs, key, qp);
        if (EXTRA && wave == nwaves - 1 && (HDP >= 64 || lane < HDP)) { X2[lane] = Ks[lane]; X3[lane] = Vs[lane]; }
        lds_barrier();                                                 // ---- C: K, V, X0, X1 are free
        if (has_next) request(next, parity ^ 1);

        // ---- phase 2: dK, dV
        {
            f32x4 dk[NDT], dv[NDT];
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (EXTRA) {                                               // token 0 as a query
                const float s0 = frag_dot_row<HDP>(kf, Qs, qp) * scale;
                const float dp0 = frag_dot_row<HDP>(vf, Ds, qp);
                const float p0 = __expf(s0 - Ls[0]);
                axpy_row<HDP>(dv, p0, Ds, qp);
                axpy_row<HDP>(dk, p0 * (dp0 - Es[0]) * scale, Qs, qp);
            }
            for (int t = 0; t < ntile; ++t) {
                ScoreOp<HDP> sv;
                score_load2<HDP>(sv, Qs, tok<EXTRA>(t, 0), Ds, tok<EXTRA>(t, 0), r, qp);
                AccOp<HDP> avd, avq;
                accum_load<HDP>(avd, Ds, tok<EXTRA>(t, 0), r, qp);
                accum_load<HDP>(avq, Qs, tok<EXTRA>(t, 0), r, qp);
                f32x4 sc, dp;                                          // rows: queries of tile t, col: own key
                score_mma2<HDP>(sv, kf, vf, sc, dp);
                f32x4 p, ds;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int query = tok<EXTRA>(t, 4 * qp + e);
                    p[e] = (EXTRA || (query < N && kok)) ? __expf(sc[e] * scale - Ls[query]) : 0.f;
                    ds[e] = p[e] * (dp[e] - Es[query]) * scale;
                }
                accum_mma2<HDP>(dv, avd, p, dk, avq, ds);
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
            lds_barrier();                                             // ---- D
            if (wave == 0) {
                float* drow = dqkv + (long)b * N * E3 + h * hd;
                tok0_dkv_combine<HDP>(Qs, Ds, X2, X3, PA + HDP, 3 * HDP, nwaves, Ls[0], Es[0], scale, drow + E, drow + 2 * E, hd, lane);
            }
        }
        if (!has_next) break;
        item = next;
        parity ^= 1;
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
    const int mode = g_attn_fused.load(std::memory_order_relaxed);
    if constexpr (ACfg<HDP>::VEC) {
        if (mode && attn_tiles(N) <= 4 && hd == HDP && (long)N * 3 * H * hd * 4 < (1l << 31)) {
            const int nitems = B * H, ipw = mode >= 2 ? mode : 2;
            const size_t roll_lds = ((lds + 15) & ~(size_t)15) + (size_t)attn_tiles(N) * 16 * HDP * sizeof(float);   // + the query image
            hipLaunchKernelGGL((attn_fwd_roll_kernel<HDP, EXTRA>), dim3(cdiv(nitems, ipw)), dim3(64 * attn_waves(N)), roll_lds, st, qkv,
                               out, lse, N, H, hd, 1.0f / sqrtf((float)hd), nitems);
            VSOM_LAUNCH_CHECK("attn_fwd_roll_kernel");
        }
    }
    hipLaunchKernelGGL((attn_fwd_kernel<HDP, EXTRA>), dim3(B * H), dim3(64 * attn_waves(N)), lds, st, qkv, out, lse, N, H, hd,
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
    const int mode = g_attn_fused.load(std::memory_order_relaxed);
    if constexpr (ACfg<HDP>::VEC) {
        const int nrp = ((use_extra(N) ? N : cdiv(N, 16) * 16) + 3) & ~3;
        const size_t roll_lds = fused_lds + ((size_t)nrp + 4 * HDP) * sizeof(float);           // second log-sum-exp buffer, 4 token-0 rows
        if (mode && attn_tiles(N) <= 4 && hd == HDP && roll_lds <= 80 * 1024 && (long)N * 3 * H * hd * 4 < (1l << 31) &&
            (long)B * H * N * 4 < (1l << 31)) {
            const int nitems = B * H, ipw = mode >= 2 ? mode : 3;
            hipLaunchKernelGGL((attn_bwd_roll_kernel<HDP, EXTRA>), dim3(cdiv(nitems, ipw)), block, roll_lds, st, qkv, out, dout, lse,
                               dqkv, delta, N, H, hd, scale, nitems, (unsigned)((long)B * H * N * 4));
            VSOM_LAUNCH_CHECK("attn_bwd_roll_kernel");
        }
    }
    if (ACfg<HDP>::VEC && fused_lds <= 80 * 1024 && mode) {
        hipLaunchKernelGGL((attn_bwd_fused_kernel<HDP, EXTRA>), dim3(B * H), block, fused_lds, st, qkv, out, dout, lse, dqkv,
                           delta, N, H, hd, scale);
        VSOM_LAUNCH_CHECK("attn_bwd_fused_kernel");
    }
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HDP, EXTRA>), dim3(B * H), block, attn_lds_bytes(N, HDP, false, HDP), st, qkv, out,
                       dout, lse, dqkv, delta, N, H, hd, scale);
    int rc = hip_status(hipGetLastError(), "attn_bwd_dq_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDP, EXTRA>), dim3(B * H), block, attn_lds_bytes(N, HDP, true, 2 * HDP), st, qkv,
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
    g_attn_fused.store(fused < 0 ? 0 : fused, std::memory_order_relaxed);
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

}  // extern "C"
