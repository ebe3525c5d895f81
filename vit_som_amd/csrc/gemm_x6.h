// fp32-accurate GEMM on the bf16 matrix cores: "split-bf16" (3 pieces per operand, 6 products).
//
// An fp32 value has a 24-bit significand; bf16 keeps fp32's exponent range and 8 significand
// bits.  Truncating to bf16 three times,
//     a1 = trunc16(a),  a2 = trunc16(a - a1),  a3 = a - a1 - a2
// gives a = a1 + a2 + a3 EXACTLY (8 + 8 + 8 bits; both subtractions are exact in fp32), each
// piece a bf16 number, |a2| < 2^-7 |a|, |a3| < 2^-15 |a|.  A product a*b is the 9 terms ai*bj;
// every ai*bj is exact in the MFMA's fp32 accumulator (8x8 -> 16 significand bits).  The six
// terms with i + j <= 4 are accumulated on v_mfma_f32_32x32x16_bf16, smallest first; the three
// dropped terms are below 2^-22 |a b| together, i.e. the size of the fp32 rounding an fmaf chain
// commits at every step anyway.  Measured against fp64 on the qkv GEMM of the step: relative
// error 1.3e-7 here vs 1.9e-7 for the v_mfma_f32_32x32x2_f32 engine.  The MFMA time per k drops
// from 32 to 12 cycles per 32x32 tile; range and denormal behaviour are fp32's (no scaling).
//
// Layouts, per operand: k-contiguous (both: "NT", Y = X W^T; the Linear input-gradient GEMM uses a
// transposed weight copy so that it is NT too) or k-strided (both: "TN", dW = dY^T X, reduction
// over the token rows; B only: "NN", the SOM input gradient coef W).  For a k-strided operand a
// thread loads a 4(k) x 4(cols) block, transposes it in registers and writes the same
// k-contiguous bf16 planes, so the MFMA loop is shared.
// Tiles 128 x 64 x 32 (4 waves of 32 x 64) or 64 x 64 x 32 (2 x 2 waves of 32 x 32), register-staged like
// gemm_f32.h; the split happens between the global load and the LDS store (4 and / 4 sub / 3
// perm per pair of elements).  LDS holds three bf16 planes per operand, rows of 32 bf16 (64 B) with an
// XOR chunk swizzle (x6_chunk_off: conflict-free reads AND stores).  Fragment layout of the 32x32x16
// MFMA: lane (r = l & 31, h = l >> 5) supplies A[row r][k = 8h + j] and B[k = 8h + j][col r],
// j = 0..7 -- one 16-byte LDS read per plane per 16-k step; the accumulator layout equals the
// fp32 MFMA's, so the epilogues are shared.
#pragma once
#include "gemm_f32.h"

namespace vsom {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ unsigned x6_bits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float x6_float(unsigned x) { return __builtin_bit_cast(float, x); }

// 4 floats -> 3 planes of 4 bf16 (8 bytes per plane), element e at bytes 2e..2e+1
__device__ __forceinline__ void x6_split(f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
    const unsigned HI = 0xffff0000u, SEL = 0x07060302u;      // perm: high halves of (S1, S0) -> (lo, hi)
    float r[4], s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = v[e] - x6_float(x6_bits(v[e]) & HI);
        s[e] = r[e] - x6_float(x6_bits(r[e]) & HI);
    }
    p1.x = __builtin_amdgcn_perm(x6_bits(v[1]), x6_bits(v[0]), SEL); p1.y = __builtin_amdgcn_perm(x6_bits(v[3]), x6_bits(v[2]), SEL);
    p2.x = __builtin_amdgcn_perm(x6_bits(r[1]), x6_bits(r[0]), SEL); p2.y = __builtin_amdgcn_perm(x6_bits(r[3]), x6_bits(r[2]), SEL);
    p3.x = __builtin_amdgcn_perm(x6_bits(s[1]), x6_bits(s[0]), SEL); p3.y = __builtin_amdgcn_perm(x6_bits(s[3]), x6_bits(s[2]), SEL);
}

// 4 floats -> TWO planes of 4 bf16 (round to nearest even, v_cvt_pk_bf16_f32): a = a1 + a2 + r with |a2| <= 2^-9 |a|,
// |r| <= 2^-17 |a|.  Three products a2 b1 + a1 b2 + a1 b1 then carry |error| <= 3 * 2^-16 |a||b| per term in the worst case
// (random signs: ~2^-18 on a long sum): the BMU contraction (bmu_x3.hip) and the weight-gradient GEMM (gemm_x6_tn.h).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ void x3_split(f32x4 v, uint2& p1, uint2& p2) {
    const bf16x2_t a01 = {(__bf16)v[0], (__bf16)v[1]}, a23 = {(__bf16)v[2], (__bf16)v[3]};
    const unsigned u01 = __builtin_bit_cast(unsigned, a01), u23 = __builtin_bit_cast(unsigned, a23);
    const float r0 = v[0] - x6_float(u01 << 16), r1 = v[1] - x6_float(u01 & 0xffff0000u);
    const float r2 = v[2] - x6_float(u23 << 16), r3 = v[3] - x6_float(u23 & 0xffff0000u);
    const bf16x2_t b01 = {(__bf16)r0, (__bf16)r1}, b23 = {(__bf16)r2, (__bf16)r3};
    p1.x = u01; p1.y = u23;
    p2.x = __builtin_bit_cast(unsigned, b01); p2.y = __builtin_bit_cast(unsigned, b23);
}

// LDS plane image: [row][64 B] = 32 bf16 of one k-tile, no padding; the 16-byte chunk c (8 consecutive k) of row R sits
// at chunk position c ^ ((R >> 2) & 3).  With that XOR both access patterns are bank-conflict-free: the fragment reads
// (ds_read_b128, lane = row: its 16-lane groups hit 16 different 16-byte slots of the 256-byte bank row) AND the staging
// stores (ds_write_b64, 8 lanes per row: a 16-lane group covers two whole rows = 128 contiguous bytes).  The padded
// [row][80 B] image of round 1 was conflict-free for the reads only -- every store was 2-way (SQ_LDS_BANK_CONFLICT = a
// third of the LDS cycles of the kernel).
constexpr int X6_RS = 64;        // bytes per plane row
__device__ __forceinline__ int x6_chunk_off(int row, int chunk) { return row * X6_RS + ((chunk ^ ((row >> 2) & 3)) << 4); }
// byte offset of the 8-byte piece kq (4 consecutive k, kq = 0..7) of row `row`
__device__ __forceinline__ int x6_piece_off(int row, int kq) { return x6_chunk_off(row, kq >> 1) + ((kq & 1) << 3); }

template <int ROWS, int NPL = 3>
__device__ __forceinline__ void x6_store(const StageRegs<ROWS>& s, char* planes, int t) {
    constexpr int PL = ROWS * X6_RS;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        uint2 p1, p2, p3;
        if constexpr (NPL == 3) x6_split(s.v[p], p1, p2, p3); else x3_split(s.v[p], p1, p2);
        char* dst = planes + x6_piece_off(p * 32 + (t >> 3), t & 7);
        *reinterpret_cast<uint2*>(dst) = p1;
        *reinterpret_cast<uint2*>(dst + PL) = p2;
        if constexpr (NPL == 3) *reinterpret_cast<uint2*>(dst + 2 * PL) = p3;
    }
}

// ---- k-strided ("TN") staging: a task = 4 consecutive reduction rows x 4 consecutive columns.
// Lane order inside a task group is k-group fastest (kg = t % 8, column quad = t / 8): a wave's
// load touches 8 rows x 128 contiguous bytes, and its 8-byte LDS writes are conflict-free.
struct X6Blk { f32x4 v[4]; };
__device__ __forceinline__ void x6_load_ks(X6Blk& b, __amdgpu_buffer_rsrc_t rsrc, unsigned colbytes, unsigned ld4, int k0,
                                           int K, int kg, int seg, int stride, int off0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + 4 * kg + i;
        const unsigned row = seg ? (unsigned)((k / seg) * stride + off0 + (k % seg)) : (unsigned)k;
        b.v[i] = bload4(rsrc, (k < K && colbytes != OOB) ? row * ld4 + colbytes : OOB);
    }
}
template <int NPL = 3>
__device__ __forceinline__ void x6_store_ks(const X6Blk& b, char* planes, int plane_bytes, int mq, int kg) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const f32x4 col = {b.v[0][e], b.v[1][e], b.v[2][e], b.v[3][e]};      // 4 consecutive k of column 4mq+e
        uint2 p1, p2, p3;
        if constexpr (NPL == 3) x6_split(col, p1, p2, p3); else x3_split(col, p1, p2);
        char* dst = planes + x6_piece_off(4 * mq + e, kg);
        *reinterpret_cast<uint2*>(dst) = p1;
        *reinterpret_cast<uint2*>(dst + plane_bytes) = p2;
        if constexpr (NPL == 3) *reinterpret_cast<uint2*>(dst + 2 * plane_bytes) = p3;
    }
}

// NPL = 3: exact three-piece split, six products; NPL = 2: two-piece round-to-nearest split, three products (x3_split above)
template <bool A_KC, bool B_KC, int WM, int WN, int WAVES_M, int WAVES_N, int EPI, int NPL = 3>
__global__ __launch_bounds__(256) void gemm_x6_kernel(const GemmP g) {
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int PA = BM * X6_RS, PB = BN * X6_RS;
    __shared__ __attribute__((aligned(16))) char lds[NPL * (PA + PB)];
    char* As = lds;
    char* Bs = lds + NPL * PA;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WAVES_N) * (WM * 32);
    const int wn0 = (wave % WAVES_N) * (WN * 32);

    // same (split, tile) -> workgroup order as gemm_f32_kernel
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles_m = (g.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int z = lid / ntiles;
    const int rem = lid - z * ntiles;
    const int tm = g.n_major ? rem % tiles_m : rem / tiles_n;
    const int tn = g.n_major ? rem / tiles_m : rem % tiles_n;
    const int bm0 = tm * BM;
    const int bn0 = tn * BN;

    const int ktiles = (g.K + 31) >> 5;
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

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (int)g.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (int)g.b_bytes, 0x00020000);
    // per-operand staging state: k-contiguous operands use StageRegs / OffKC (all 256 threads),
    // k-strided ones the 4x4 task map (A tasks on threads [0, 2 BM), B tasks on [256 - 2 BN, 256))
    StageRegs<A_KC ? BM : 32> sa;
    StageRegs<B_KC ? BN : 32> sb;
    OffKC<A_KC ? BM : 32> oa; OffKC<B_KC ? BN : 32> ob;
    constexpr int TA = 2 * BM, TB0 = 256 - 2 * BN;
    static_assert(TA <= 256 && TB0 >= 0, "tile too large for the k-strided task map");
    const bool has_a = !A_KC && t < TA, has_b = !B_KC && t >= TB0;
    const int kga = t & 7, mqa = t >> 3, kgb = (t - TB0) & 7, mqb = (t - TB0) >> 3;
    X6Blk ba, bb;
    unsigned cola = OOB, colb = OOB;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};          // EPI_SLAB bias partial: column sums of A over this thread's k rows
    const bool want_colsum = (EPI == EPI_SLAB) && !A_KC && g.slab_bias != nullptr && bn0 == 0;
    if constexpr (A_KC) init_kc<BM>(oa, g.lda, bm0, g.M, t);
    else if (has_a && bm0 + 4 * mqa < g.M) cola = (unsigned)(bm0 + 4 * mqa) << 2;
    if constexpr (B_KC) init_kc<BN>(ob, g.ldb, bn0, g.N, t);
    else if (has_b && bn0 + 4 * mqb < g.N) colb = (unsigned)(bn0 + 4 * mqb) << 2;
    auto gload = [&](int kt) {
        if constexpr (A_KC) load_kc_fast<BM>(sa, rsA, oa, kt << 5, g.K, t);
        else if (has_a) x6_load_ks(ba, rsA, cola, (unsigned)g.lda << 2, kt << 5, g.K, kga, g.a_seg, g.a_stride, g.a_off);
        if constexpr (B_KC) load_kc_fast<BN>(sb, rsB, ob, kt << 5, g.K, t);
        else if (has_b) x6_load_ks(bb, rsB, colb, (unsigned)g.ldb << 2, kt << 5, g.K, kgb, 0, 0, 0);
    };
    auto lstore = [&]() {
        if constexpr (A_KC) {
            x6_store<BM, NPL>(sa, As, t);
        } else if (has_a) {
            if (want_colsum) {
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[e] += (ba.v[0][e] + ba.v[1][e]) + (ba.v[2][e] + ba.v[3][e]);
            }
            x6_store_ks<NPL>(ba, As, PA, mqa, kga);
        }
        if constexpr (B_KC) x6_store<BN, NPL>(sb, Bs, t);
        else if (has_b) x6_store_ks<NPL>(bb, Bs, PB, mqb, kgb);
    };

    if (kt_begin < kt_end) {
        gload(kt_begin);
        lstore();
    }
    __syncthreads();

    auto mfma_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[WM][NPL], b[WN][NPL];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    a[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PA + x6_chunk_off(wm0 + i * 32 + r, 2 * ks + h));
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    b[j][pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * PB + x6_chunk_off(wn0 + j * 32 + r, 2 * ks + h));
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
    // The steady-state loop body is branch-free (the last k-tile is peeled): with a conditional
    // prefetch inside, hipcc carries the accumulators through VGPRs and copies all of them
    // AGPR -> VGPR -> AGPR on every iteration.
    for (int kt = kt_begin; kt + 1 < kt_end; ++kt) {
        gload(kt + 1);
        __builtin_amdgcn_sched_barrier(0);      // loads first, then the whole MFMA phase, and only then the split
        mfma_tile();                            // (left alone, hipcc interleaves load -> vmcnt(0) -> split into the MFMAs)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        lstore();
        __syncthreads();
    }
    if (kt_begin < kt_end) mfma_tile();
    if constexpr (EPI == EPI_SLAB && !A_KC) {
        if (want_colsum && has_a) {              // the 8 k-groups of a column quad are 8 consecutive lanes
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = cs[e];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
                const int m = bm0 + 4 * mqa + e;
                if (kga == 0 && m < g.M) g.slab_bias[(long)z * g.slab_bias_stride + m] = v;
            }
        }
    }
    gemm_epilogue<WM, WN, EPI>(g, acc, bm0 + wm0, bn0 + wn0, r, h, z);
}

}  // namespace vsom
