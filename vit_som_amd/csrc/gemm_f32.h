// Exact-fp32 MFMA GEMM engine for gfx950 (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain).
//
//   C[M,N] = epilogue( sum_k opA(m,k) * opB(k,n) )
//
// Operand storage modes (the reduction index is "k"):
//   A_KC  : A element (m,k) at A[m*lda + k]   (k contiguous)      else A[k*lda + m] (k strided)
//   B_KC  : B element (k,n) at B[n*ldb + k]   (k contiguous)      else B[k*ldb + n] (k strided)
//   forward  Linear  Y = X W^T      : A_KC,  B_KC   ("NT")
//   input    grad    dX = dY W      : A_KC, !B_KC   ("NN")
//   weight   grad    dW = dY^T X    : !A_KC, !B_KC  ("TN", split over the reduction -> slabs)
//
// Tiling: 256 threads = 4 waves; block tile 128 x 64, BK = 32; each wave owns WM x WN = 1 x 2
// tiles of 32x32 (16 accumulator VGPRs each).  Operand tiles are staged global ->
// registers -> LDS (one LDS buffer, next tile's global loads in flight during the MFMAs); with
// f32 MFMA at 64 cycles per instruction the 2-3 co-resident workgroups per CU cover the staging.
// One workgroup per output tile (a persistent / cross-tile-prefetch variant measured no faster:
// hipcc then shuttles the accumulators AGPR<->VGPR every k-tile and drains vmcnt early).
// LDS images: k-contiguous tiles are [rows][36] (row stride 144 B: conflict-free ds_read_b128 of
// 4 consecutive k per lane), k-strided tiles are [32][cols+4] (ds_read_b32 across 32 consecutive
// columns).  Lane l = (r = l & 31, h = l >> 5) feeds MFMA step s of an 8-deep group with
// k = kb + 4h + s for BOTH operands, so any consistent assignment is exact.
#pragma once
#include "common.h"

namespace vsom {

enum Epi : int {
    EPI_NONE = 0,       // v = alpha*acc (+ C if accumulate)
    EPI_BIAS = 1,       // v = acc + bias[n]
    EPI_BIAS_GELU = 2,  // pre = acc + bias[n]; C = gelu'(pre); C2 = gelu(pre)
    EPI_BIAS_RES = 3,   // v = acc + bias[n] + R[(m % r_mod + r_off)*ldr + n]; output row map
    EPI_ROWAXPY = 4,    // v = acc + rowscale[m]*R[m*ldr + n] (+ C if accumulate)
    EPI_GELU_BWD = 5,   // v = acc * R[m*ldr + n] (R = gelu'(pre) saved by the forward) (+ C if accumulate)
    EPI_SLAB = 6,       // split-k partial: slab[z][m*N + n] = acc; optional column sums of A
    EPI_BIAS_RELU = 7,  // pre = acc + bias[n]; C = (pre > 0) as 1.0/0.0 (the derivative); C2 = max(pre, 0)
};

struct GemmP {
    const float* A; const float* B; float* C;
    long lda, ldb, ldc;
    int M, N, K;
    int ktiles_per_split;        // k-tiles (of 32) handled by one blockIdx.z
    // reduction-row map for a k-strided A (rows of A are reduction indices):
    //   row(k) = (k / a_seg)*a_stride + a_off + k % a_seg ; a_seg == 0 -> identity
    int a_seg, a_stride, a_off;
    // output-row map, same form (c_seg == 0 -> identity)
    int c_seg, c_stride, c_off;
    const float* bias;
    const float* R; long ldr; int r_mod, r_off;
    float* C2; long ldc2;
    const float* rowscale;
    float alpha; int accumulate;
    float* slab; long slab_stride;   // EPI_SLAB
    float* slab_bias; long slab_bias_stride;   // per-split column sums of A (bias gradient), or null
    int n_major;                 // tile order inside a split: 1 = consecutive ids walk the m-tiles of one n-tile
    int a_vec, b_vec;            // 16-byte vector loads legal for A / B
    unsigned a_bytes, b_bytes;   // extent of A / B for the bounds-checked buffer loads (FAST path)
    int products;                // split-bf16 engine: 6 (exact three-piece split; default when 0) or 3 (two-piece split)
};

template <int R_>
struct StageRegs { f32x4 v[R_ / 32]; };

constexpr unsigned OOB = 0xFFFFFFF0u;      // byte offset past any buffer: raw buffer loads return 0

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
}

// same with a wave-uniform byte offset in the SGPR operand (it takes part in the range check: num_records - soffset)
__device__ __forceinline__ f32x4 bload4s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

// ---- global -> register staging ------------------------------------------------------------
// FAST path: 16-byte hardware-bounds-checked buffer loads.  Each thread's byte offsets into the
// operand are computed ONCE per output tile (OOB when its row / column is outside the matrix);
// inside the k-loop a load costs one add and one select.
// k-contiguous tile: ROWS x 32, thread t loads float4 at (row = p*32 + t/8, k = (t%8)*4)
template <int ROWS>
struct OffKC { unsigned off[ROWS / 32]; };
template <int ROWS>
__device__ __forceinline__ void init_kc(OffKC<ROWS>& o, long ld, int row0, int nrows, int t) {
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        const int row = row0 + p * 32 + (t >> 3);
        o.off[p] = (row < nrows) ? (unsigned)(((long)row * ld + ((t & 7) << 2)) << 2) : OOB;
    }
}
template <int ROWS>
__device__ __forceinline__ void load_kc_fast(StageRegs<ROWS>& s, __amdgpu_buffer_rsrc_t rsrc, const OffKC<ROWS>& o,
                                             int k0, int K, int t) {
    const bool kok = k0 + ((t & 7) << 2) < K;
    const unsigned kbytes = (unsigned)k0 << 2;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) s.v[p] = bload4(rsrc, (kok && o.off[p] != OOB) ? o.off[p] + kbytes : OOB);
}
// k-strided tile: 32 x COLS, thread t loads float4 at (k = p*KPP + t/(COLS/4), col = (t%(COLS/4))*4)
template <int COLS>
struct OffKS { unsigned col; unsigned ldb4; };
template <int COLS>
__device__ __forceinline__ void init_ks(OffKS<COLS>& o, long ld, int col0, int ncols, int t) {
    constexpr int TPR = COLS / 4;
    const int col = col0 + ((t % TPR) << 2);
    o.col = (col < ncols) ? (unsigned)col << 2 : OOB;
    o.ldb4 = (unsigned)ld << 2;
}
template <int COLS>
__device__ __forceinline__ void load_ks_fast(StageRegs<COLS>& s, __amdgpu_buffer_rsrc_t rsrc, const OffKS<COLS>& o,
                                             int k0, int K, int t, int seg, int stride, int off0) {
    constexpr int TPR = COLS / 4;
    constexpr int KPP = 256 / TPR;
    const int kk = k0 + t / TPR;
#pragma unroll
    for (int p = 0; p < COLS / 32; ++p) {
        const int k = kk + p * KPP;
        const unsigned row = seg ? (unsigned)((k / seg) * stride + off0 + (k % seg)) : (unsigned)k;
        s.v[p] = bload4(rsrc, (k < K && o.col != OOB) ? row * o.ldb4 + o.col : OOB);
    }
}
// generic (any alignment / ragged K) variants
template <int ROWS>
__device__ __forceinline__ void load_kc(StageRegs<ROWS>& s, const float* __restrict__ base, long ld,
                                        int row0, int nrows, int k0, int K, int vec, int t) {
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        const int row = row0 + p * 32 + (t >> 3);
        const int k = k0 + ((t & 7) << 2);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < nrows && k < K) {
            const float* src = base + (long)row * ld + k;
            if (vec && k + 3 < K) {
                v = *reinterpret_cast<const f32x4*>(src);
            } else {
                v[0] = src[0];
                if (k + 1 < K) v[1] = src[1];
                if (k + 2 < K) v[2] = src[2];
                if (k + 3 < K) v[3] = src[3];
            }
        }
        s.v[p] = v;
    }
}
template <int ROWS>
__device__ __forceinline__ void store_kc(const StageRegs<ROWS>& s, float* lds, int t) {
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p)
        *reinterpret_cast<f32x4*>(lds + (p * 32 + (t >> 3)) * 36 + ((t & 7) << 2)) = s.v[p];
}
template <int COLS>
__device__ __forceinline__ void load_ks(StageRegs<COLS>& s, const float* __restrict__ base, long ld,
                                        int col0, int ncols, int k0, int K, int vec, int t, int seg,
                                        int stride, int off) {
    constexpr int TPR = COLS / 4;        // threads per k-row
    constexpr int KPP = 256 / TPR;       // k-rows per pass
#pragma unroll
    for (int p = 0; p < COLS / 32; ++p) {
        const int k = k0 + p * KPP + t / TPR;
        const int col = col0 + ((t % TPR) << 2);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < K && col < ncols) {
            const long row = seg ? (long)(k / seg) * stride + off + (k % seg) : (long)k;
            const float* src = base + row * ld + col;
            if (vec && col + 3 < ncols) {
                v = *reinterpret_cast<const f32x4*>(src);
            } else {
                v[0] = src[0];
                if (col + 1 < ncols) v[1] = src[1];
                if (col + 2 < ncols) v[2] = src[2];
                if (col + 3 < ncols) v[3] = src[3];
            }
        }
        s.v[p] = v;
    }
}
template <int COLS>
__device__ __forceinline__ void store_ks(const StageRegs<COLS>& s, float* lds, int t) {
    constexpr int TPR = COLS / 4;
    constexpr int KPP = 256 / TPR;
#pragma unroll
    for (int p = 0; p < COLS / 32; ++p)
        *reinterpret_cast<f32x4*>(lds + (p * KPP + t / TPR) * (COLS + 4) + ((t % TPR) << 2)) = s.v[p];
}

// ---- epilogue (shared by the exact-f32 and the split-bf16 kernels) ----------------------------
// accumulator register v of a 32x32 tile holds row (v&3) + 8*(v>>2) + 4*h, column r: walk the
// rows with pointer increments (no per-element index arithmetic); the rare output-row map
// (patch embedding) takes the indexed path.  (m0, n0) = first row / column of this WAVE's tiles.
template <int WM, int WN, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmP& g, const f32x16 (&acc)[WM][WN], int m0, int n0, int r, int h, int z) {
    const bool rowmap = (EPI == EPI_BIAS_RES) && g.c_seg != 0;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + j * 32 + r;
            if (n >= g.N) continue;
            const int mb = m0 + i * 32 + 4 * h;
            float bn = 0.f;
            if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_RES) bn = g.bias ? g.bias[n] : 0.f;
            // Fast path: all 32 rows of this tile are inside the matrix (every tile of the step's
            // GEMMs): no per-element guards, rows walked with running pointers (register v -> v+1 is
            // the next row, every 4th step skips to the lane group's next 4-row band).
            const bool plain_rows = (m0 + i * 32 + 31 < g.M) && !rowmap &&
                                    !((EPI == EPI_BIAS_RES) && !(g.r_mod >= g.M && g.r_off == 0));
            if (plain_rows) {
                float* d = (EPI == EPI_SLAB) ? g.slab + (long)z * g.slab_stride + (long)mb * g.N + n : g.C + (long)mb * g.ldc + n;
                const long dstep = (EPI == EPI_SLAB) ? (long)g.N : g.ldc;
                const float* rp = nullptr;
                float* d2 = nullptr;
                if constexpr (EPI == EPI_ROWAXPY || EPI == EPI_GELU_BWD || EPI == EPI_BIAS_RES) rp = g.R + (long)mb * g.ldr + n;
                if constexpr (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU) d2 = g.C2 + (long)mb * g.ldc2 + n;
                const float* rs = nullptr;
                if constexpr (EPI == EPI_ROWAXPY) rs = g.rowscale + mb;
                // every input of the 16 rows is loaded BEFORE the first store: with loads and stores
                // interleaved hipcc must assume aliasing and waits for each load (vmcnt(0)) before the next
                // store -- 16 (or 32) serialised global-memory latencies per tile
                constexpr bool READS_R = (EPI == EPI_ROWAXPY || EPI == EPI_GELU_BWD || EPI == EPI_BIAS_RES);
                constexpr bool MAY_ACC = (EPI == EPI_NONE || EPI == EPI_ROWAXPY || EPI == EPI_GELU_BWD);
                float rv[16], dv[16];
                if constexpr (READS_R) {
                    const float* q = rp;
#pragma unroll
                    for (int v = 0; v < 16; ++v) { rv[v] = *q; q += (((v & 3) == 3) ? 5 : 1) * g.ldr; }
                }
                float sv[16];
                if constexpr (EPI == EPI_ROWAXPY) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) sv[v] = rs[(v & 3) + 8 * (v >> 2)];
                }
                const bool accum = MAY_ACC && g.accumulate;
                if (accum) {
                    const float* q = d;
#pragma unroll
                    for (int v = 0; v < 16; ++v) { dv[v] = *q; q += (((v & 3) == 3) ? 5 : 1) * dstep; }
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float a = acc[i][j][v];
                    if constexpr (EPI == EPI_SLAB) {
                        *d = a;
                    } else if constexpr (EPI == EPI_NONE) {
                        float val = g.alpha * a;
                        if (accum) val += dv[v];
                        *d = val;
                    } else if constexpr (EPI == EPI_BIAS) {
                        *d = a + bn;
                    } else if constexpr (EPI == EPI_BIAS_GELU) {
                        float act, grad;
                        gelu_erf_both(a + bn, act, grad);
                        *d = grad;
                        *d2 = act;
                    } else if constexpr (EPI == EPI_BIAS_RELU) {
                        const float pre = a + bn;
                        *d = pre > 0.f ? 1.0f : 0.f;
                        *d2 = fmaxf(pre, 0.f);
                    } else if constexpr (EPI == EPI_BIAS_RES) {
                        *d = a + bn + rv[v];
                    } else if constexpr (EPI == EPI_ROWAXPY) {
                        float val = a + sv[v] * rv[v];
                        if (accum) val += dv[v];
                        *d = val;
                    } else if constexpr (EPI == EPI_GELU_BWD) {
                        float val = a * rv[v];
                        if (accum) val += dv[v];
                        *d = val;
                    }
                    const long adv = ((v & 3) == 3) ? 5 : 1;          // rows (v&3) + 8 (v>>2): +1, +1, +1, +5
                    d += adv * dstep;
                    if constexpr (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU) d2 += adv * g.ldc2;
                }
                continue;
            }
            if constexpr (EPI == EPI_SLAB) {
                float* dst = g.slab + (long)z * g.slab_stride + (long)mb * g.N + n;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int dm = (v & 3) + 8 * (v >> 2);
                    if (mb + dm < g.M) dst[(long)dm * g.N] = acc[i][j][v];
                }
            } else if (rowmap) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = mb + (v & 3) + 8 * (v >> 2);
                    if (m >= g.M) continue;
                    const long orow = (long)(m / g.c_seg) * g.c_stride + g.c_off + (m % g.c_seg);
                    const long rr = (long)(m % g.r_mod) + g.r_off;
                    g.C[orow * g.ldc + n] = acc[i][j][v] + bn + g.R[rr * g.ldr + n];
                }
            } else {
                float* dst = g.C + (long)mb * g.ldc + n;
                const float* rsrc = nullptr;
                float* dst2 = nullptr;
                if constexpr (EPI == EPI_ROWAXPY || EPI == EPI_GELU_BWD) rsrc = g.R + (long)mb * g.ldr + n;
                if constexpr (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU) dst2 = g.C2 + (long)mb * g.ldc2 + n;
                const bool plain_res = (EPI == EPI_BIAS_RES) && g.r_mod >= g.M && g.r_off == 0;
                if constexpr (EPI == EPI_BIAS_RES) rsrc = g.R + (long)mb * g.ldr + n;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int dm = (v & 3) + 8 * (v >> 2);
                    const int m = mb + dm;
                    if (m >= g.M) continue;
                    const float a = acc[i][j][v];
                    float* d = dst + (long)dm * g.ldc;
                    if constexpr (EPI == EPI_NONE) {
                        float val = g.alpha * a;
                        if (g.accumulate) val += *d;
                        *d = val;
                    } else if constexpr (EPI == EPI_BIAS) {
                        *d = a + bn;
                    } else if constexpr (EPI == EPI_BIAS_GELU) {
                        float act, grad;
                        gelu_erf_both(a + bn, act, grad);
                        *d = grad;
                        dst2[(long)dm * g.ldc2] = act;
                    } else if constexpr (EPI == EPI_BIAS_RELU) {
                        const float pre = a + bn;
                        *d = pre > 0.f ? 1.0f : 0.f;
                        dst2[(long)dm * g.ldc2] = fmaxf(pre, 0.f);
                    } else if constexpr (EPI == EPI_BIAS_RES) {
                        const float rv = plain_res ? rsrc[(long)dm * g.ldr] : g.R[((long)(m % g.r_mod) + g.r_off) * g.ldr + n];
                        *d = a + bn + rv;
                    } else if constexpr (EPI == EPI_ROWAXPY) {
                        float val = a + g.rowscale[m] * rsrc[(long)dm * g.ldr];
                        if (g.accumulate) val += *d;
                        *d = val;
                    } else if constexpr (EPI == EPI_GELU_BWD) {
                        float val = a * rsrc[(long)dm * g.ldr];
                        if (g.accumulate) val += *d;
                        *d = val;
                    }
                }
            }
        }
    }
}

template <bool A_KC, bool B_KC, int WM, int WN, int WAVES_M, int WAVES_N, int EPI, bool FAST>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmP g) {
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int A_LDS = A_KC ? BM * 36 : 32 * (BM + 4);
    constexpr int B_LDS = B_KC ? BN * 36 : 32 * (BN + 4);
    __shared__ __attribute__((aligned(16))) float lds[A_LDS + B_LDS];
    float* As = lds;
    float* Bs = lds + A_LDS;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WAVES_N) * (WM * 32);
    const int wn0 = (wave % WAVES_N) * (WN * 32);

    // 1-D grid of (split, tile) pairs.  After the XCD remap each XCD owns a CONTIGUOUS range of
    // logical ids; split-major order gives every XCD its own slice of the reduction range (it
    // reads that slice of both operands once, its private L2 serves the re-use), and inside a
    // split tiles are ordered so that neighbours share the LARGER operand's panel.
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

    float colsum = 0.f;                       // EPI_SLAB bias partial (thread t < BM owns A column t)
    const bool want_colsum = (EPI == EPI_SLAB) && !A_KC && g.slab_bias != nullptr && bn0 == 0;

    StageRegs<BM> sa;
    StageRegs<BN> sb;
    __amdgpu_buffer_rsrc_t rsA, rsB;
    OffKC<BM> oa_kc; OffKS<BM> oa_ks; OffKC<BN> ob_kc; OffKS<BN> ob_ks;
    if constexpr (FAST) {
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (int)g.a_bytes, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (int)g.b_bytes, 0x00020000);
        if constexpr (A_KC) init_kc<BM>(oa_kc, g.lda, bm0, g.M, t); else init_ks<BM>(oa_ks, g.lda, bm0, g.M, t);
        if constexpr (B_KC) init_kc<BN>(ob_kc, g.ldb, bn0, g.N, t); else init_ks<BN>(ob_ks, g.ldb, bn0, g.N, t);
    }
    auto gload = [&](int kt) {
        const int k0 = kt << 5;
        if constexpr (FAST) {
            if constexpr (A_KC) load_kc_fast<BM>(sa, rsA, oa_kc, k0, g.K, t);
            else load_ks_fast<BM>(sa, rsA, oa_ks, k0, g.K, t, g.a_seg, g.a_stride, g.a_off);
            if constexpr (B_KC) load_kc_fast<BN>(sb, rsB, ob_kc, k0, g.K, t);
            else load_ks_fast<BN>(sb, rsB, ob_ks, k0, g.K, t, 0, 0, 0);
        } else {
            if constexpr (A_KC) load_kc<BM>(sa, g.A, g.lda, bm0, g.M, k0, g.K, g.a_vec, t);
            else load_ks<BM>(sa, g.A, g.lda, bm0, g.M, k0, g.K, g.a_vec, t, g.a_seg, g.a_stride, g.a_off);
            if constexpr (B_KC) load_kc<BN>(sb, g.B, g.ldb, bn0, g.N, k0, g.K, g.b_vec, t);
            else load_ks<BN>(sb, g.B, g.ldb, bn0, g.N, k0, g.K, g.b_vec, t, 0, 0, 0);
        }
    };
    auto lstore = [&]() {
        if constexpr (A_KC) store_kc<BM>(sa, As, t); else store_ks<BM>(sa, As, t);
        if constexpr (B_KC) store_kc<BN>(sb, Bs, t); else store_ks<BN>(sb, Bs, t);
    };

    if (kt_begin < kt_end) {
        gload(kt_begin);
        lstore();
    }
    __syncthreads();

    auto mfma_tile = [&]() {
        if (want_colsum && t < BM) {
            float s = 0.f;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) s += As[kk * (BM + 4) + t];
            colsum += s;
        }
#pragma unroll
        for (int kb = 0; kb < 32; kb += 8) {
            f32x4 a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                if constexpr (A_KC) {
                    a[i] = *reinterpret_cast<const f32x4*>(As + (wm0 + i * 32 + r) * 36 + kb + 4 * h);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) a[i][s] = As[(kb + 4 * h + s) * (BM + 4) + wm0 + i * 32 + r];
                }
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if constexpr (B_KC) {
                    b[j] = *reinterpret_cast<const f32x4*>(Bs + (wn0 + j * 32 + r) * 36 + kb + 4 * h);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[j][s] = Bs[(kb + 4 * h + s) * (BN + 4) + wn0 + j * 32 + r];
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    };
    // Branch-free steady-state body with the last k-tile peeled: a conditional prefetch inside the
    // loop makes hipcc carry the accumulators through VGPRs (all of them copied AGPR -> VGPR ->
    // AGPR every iteration).  The next tile's global loads stay in flight under the MFMAs.
    for (int kt = kt_begin; kt + 1 < kt_end; ++kt) {
        gload(kt + 1);
        __builtin_amdgcn_sched_barrier(0);      // keep the loads ABOVE the MFMAs (the scheduler otherwise sinks them to their use)
        mfma_tile();
        __syncthreads();
        lstore();
        __syncthreads();
    }
    if (kt_begin < kt_end) mfma_tile();

    if constexpr (EPI == EPI_SLAB) {
        if (want_colsum && t < BM && bm0 + t < g.M) g.slab_bias[(long)z * g.slab_bias_stride + bm0 + t] = colsum;
    }
    gemm_epilogue<WM, WN, EPI>(g, acc, bm0 + wm0, bn0 + wn0, r, h, z);
}

// host-side launcher (gemm_f32.hip)
int launch_gemm(bool a_kc, bool b_kc, int epi, GemmP g, int splits, hipStream_t stream);
int linear_bwd_weight_impl(const float* dY, long lddy, const float* X, long ldx, float* dW, float* db, int M, int N,
                           int K, int a_seg, int a_stride, int a_off, void* ws, size_t ws_bytes,
                           hipStream_t stream);
// out1[j] = sum_s slabs[s*stride + j]            (j < n1)
// out2[j] = sum_s slabs[s*stride + off2 + j]     (j < n2; optional second segment, e.g. the bias)
// A workgroup owns 256 consecutive columns (64 lanes x float4); its WAVES waves stride over the
// slabs with four independent accumulators each (loads in flight instead of one dependent chain)
// and are combined through LDS in wave order: the summation order is fixed -> bitwise reproducible.
template <int WAVES, int VEC>
__device__ __forceinline__ void reduce_slabs_body(f32x4 (*sh)[64], int bx, const float* __restrict__ slabs, long stride,
                                                  int nslabs, float* __restrict__ out1, long n1,
                                                  float* __restrict__ out2, long off2, long n2,
                                                  int nb1, int vec) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool seg2 = bx >= nb1;
    const long col = ((long)(seg2 ? bx - nb1 : bx) * 64 + lane) * VEC;
    const long n = seg2 ? n2 : n1;
    const float* src = slabs + (seg2 ? off2 : 0) + col;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (col < n) {
        if (VEC == 4 && vec && col + 3 < n) {
            int s = wave;
            for (; s + 3 * WAVES < nslabs; s += 4 * WAVES) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] += *reinterpret_cast<const f32x4*>(src + (long)(s + u * WAVES) * stride);
            }
            for (; s < nslabs; s += WAVES) acc[0] += *reinterpret_cast<const f32x4*>(src + (long)s * stride);
        } else if (VEC == 1) {                           // narrow outputs (LayerNorm / bias): one column per lane
            int s = wave;
            for (; s + 3 * WAVES < nslabs; s += 4 * WAVES) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u][0] += src[(long)(s + u * WAVES) * stride];
            }
            for (; s < nslabs; s += WAVES) acc[0][0] += src[(long)s * stride];
        } else {
            for (int s = wave; s < nslabs; s += WAVES)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < n) acc[0][e] += src[(long)s * stride + e];
        }
    }
    sh[wave][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (wave == 0 && col < n) {
        f32x4 t = sh[0][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) t += sh[w][lane];
        float* dst = (seg2 ? out2 : out1) + col;
        if (VEC == 1) {
            dst[0] = t[0];
        } else if (col + 3 < n && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
            *reinterpret_cast<f32x4*>(dst) = t;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < n) dst[e] = t[e];
        }
    }
}

int reduce_slabs_internal(const float* slabs, long stride, int nslabs, float* out, long n, hipStream_t stream);
int reduce_slabs2_internal(const float* slabs, long stride, int nslabs, float* out1, long n1, float* out2, long off2,
                           long n2, hipStream_t stream);
int sum_partials(const float* part, int n, float* out, hipStream_t stream);
int choose_splits(int tiles, int ktiles, int max_splits, bool prefer_xcd_multiple = false);
int gemm_tile_m(bool a_kc, bool b_kc, int M);
int gemm_grad_products();     // 3 in VSOM_GEMM_SPLIT_BF16_GRAD3 mode (gradient GEMMs on the two-piece split), else 6

}  // namespace vsom
