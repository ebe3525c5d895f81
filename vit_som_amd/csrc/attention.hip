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
// products and rank-1 updates of the accumulators; as the wave's own row it is handled by one
// additional wave with VALU only (lane per key for the dots, lane per channel for the sums).
#include "common.h"

#include <atomic>

#include <stdlib.h>
namespace vsom {

// test hook (vsom_set_attention_fused): 0 keeps the short-sequence backward as two launches
static std::atomic<int> g_attn_fused{1};

constexpr int MAXCH = 5;   // EXTRA mode: the VALU wave walks the rows in chunks of 64 -> N <= 320

template <int HDP>
struct ACfg {
    static constexpr int S = HDP + 4;             // LDS row stride (floats); 16-B aligned rows
    static constexpr int NMM = HDP / 4;           // MFMAs (4 deep) per score tile
    static constexpr int NDT = (HDP + 15) / 16;   // 16-wide output tiles over the head dim
    static constexpr bool VEC = (HDP % 16 == 0);  // head dim fully valid, 16-B vector accesses
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float group_sum(float v) {      // over the 4 lane groups (l >> 4)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
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
            *reinterpret_cast<f32x4*>(lds + row * S + 4 * c4) = v;
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
        constexpr int S = ACfg<HDP>::S, C4 = HDP / 4;
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
                    *reinterpret_cast<f32x4*>(ldsA + row * S + 4 * c4) = va[u];
                    *reinterpret_cast<f32x4*>(ldsB + row * S + 4 * c4) = vb[u];
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
    constexpr int S = ACfg<HDP>::S, C4 = HDP / 4;
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
                *reinterpret_cast<f32x4*>(l0 + row * S + 4 * c4) = v[0][u];
                *reinterpret_cast<f32x4*>(l1 + row * S + 4 * c4) = v[1][u];
                *reinterpret_cast<f32x4*>(l2 + row * S + 4 * c4) = v[2][u];
                *reinterpret_cast<f32x4*>(l3 + row * S + 4 * c4) = v[3][u];
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

// acc[4q'+reg][own row] = sum_d Y[row0 + 4q'+reg][d] * own[row][d]    (row0 = first token of the tile)
template <int HDP>
__device__ __forceinline__ f32x4 score_tile(const float* Ylds, int row0, int r, int qp,
                                            const float (&bf)[ACfg<HDP>::NMM]) {
    float af[ACfg<HDP>::NMM];
    load_frag<HDP>(af, Ylds + (row0 + r) * ACfg<HDP>::S, qp, true, HDP);
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
    load_frag<HDP>(a0, Y0 + (row0 + r) * ACfg<HDP>::S, qp, true, HDP);
    load_frag<HDP>(a1, Y1 + (row1 + r) * ACfg<HDP>::S, qp, true, HDP);
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
    constexpr int S = ACfg<HDP>::S;
#pragma unroll
    for (int s = 0; s < 4; ++s) {                // s outer: the NDT accumulators form independent chains
#pragma unroll
        for (int dt = 0; dt < ACfg<HDP>::NDT; ++dt) {
            float a = 0.f;
            if (ACfg<HDP>::VEC || 16 * dt + r < HDP) a = Zlds[(row0 + 4 * qp + s) * S + 16 * dt + r];
            o[dt] = mfma16(a, p[s], o[dt]);
        }
    }
}
template <int HDP>
__device__ __forceinline__ void accum_tile2(f32x4 (&o0)[ACfg<HDP>::NDT], const float* Z0, f32x4 p0,
                                            f32x4 (&o1)[ACfg<HDP>::NDT], const float* Z1, f32x4 p1, int row0, int r,
                                            int qp) {
    constexpr int S = ACfg<HDP>::S;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int dt = 0; dt < ACfg<HDP>::NDT; ++dt) {
            float a0 = 0.f, a1 = 0.f;
            if (ACfg<HDP>::VEC || 16 * dt + r < HDP) {
                a0 = Z0[(row0 + 4 * qp + s) * S + 16 * dt + r];
                a1 = Z1[(row0 + 4 * qp + s) * S + 16 * dt + r];
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
// s[c] = x . Y[64c + lane]   (lane per row, rows beyond nrows give 0)
template <int HDP>
__device__ __forceinline__ void rows_dot(float (&s)[MAXCH], const float* x, const float* Y, int nrows, int lane) {
    constexpr int S = ACfg<HDP>::S;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        float acc = 0.f;
        const int row = 64 * c + lane;
        if (64 * c < nrows && row < nrows) {
#pragma unroll
            for (int d4 = 0; d4 < HDP / 4; ++d4) {
                const f32x4 y = *reinterpret_cast<const f32x4*>(Y + row * S + 4 * d4);
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + 4 * d4);
                acc = fmaf(y[0], xv[0], acc); acc = fmaf(y[1], xv[1], acc);
                acc = fmaf(y[2], xv[2], acc); acc = fmaf(y[3], xv[3], acc);
            }
        }
        s[c] = acc;
    }
}
// sum_row w[row] * Z[row][d]   (lane per channel d < HDP)
template <int HDP>
__device__ __forceinline__ float rows_wsum(const float* w, const float* Z, int nrows, int d) {
    constexpr int S = ACfg<HDP>::S;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int row = 0;
    for (; row + 3 < nrows; row += 4) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + row);
        a0 = fmaf(wv[0], Z[(row + 0) * S + d], a0);
        a1 = fmaf(wv[1], Z[(row + 1) * S + d], a1);
        a2 = fmaf(wv[2], Z[(row + 2) * S + d], a2);
        a3 = fmaf(wv[3], Z[(row + 3) * S + d], a3);
    }
    for (; row < nrows; ++row) a0 = fmaf(w[row], Z[row * S + d], a0);
    return (a0 + a1) + (a2 + a3);
}
__device__ __forceinline__ void lds_fence_wave() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// LDS carve shared by the three kernels
template <int HDP, bool EXTRA>
struct Carve {
    int ntile, nrows, nrp;
    float *Y0, *Y1, *L0, *L1, *X0, *X1, *W1, *W2;
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
        W1 = X1 + HDP;
        W2 = W1 + nrp;
    }
};

// ------------------------------------------------------------------ forward
template <int HDP, bool EXTRA>
__global__ __launch_bounds__(576) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                       float* __restrict__ lse, int N, int H, int hd,
                                                       float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NDT = ACfg<HDP>::NDT;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * hd, E3 = 3 * E;
    const Carve<HDP, EXTRA> cv(smem, N, false);
    const int ntile = cv.ntile;
    float* Ks = cv.Y0;
    float* Vs = cv.Y1;
    const float* base = qkv + (long)b * N * E3 + h * hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int tile_waves = nwaves - (EXTRA ? 1 : 0);
    const bool extra_wave = EXTRA && wave == tile_waves;
    const int r = lane & 15, qp = lane >> 4;
    // the wave's first query fragment is requested BEFORE the K/V staging (latency overlaps it)
    float qf[ACfg<HDP>::NMM];
    if (!extra_wave) load_frag<HDP>(qf, base + (long)tok<EXTRA>(wave, r) * E3, qp, tok<EXTRA>(wave, r) < N, hd);
    stage_rows_pair<HDP>(Ks, base + E, E3, Vs, base + 2 * E, E3, N, cv.nrows, hd);
    if (EXTRA) stage_vec<HDP>(cv.X0, base, hd);                       // q of token 0
    __syncthreads();

    if (extra_wave) {                                                  // token 0 as a query: VALU only
        float s[MAXCH];
        rows_dot<HDP>(s, cv.X0, Ks, N, lane);
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) { s[c] = (64 * c + lane < N) ? s[c] * scale : -INFINITY; m = fmaxf(m, s[c]); }
        m = wave_max(m);
        float l = 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const float p = __expf(s[c] - m);
            l += p;
            if (64 * c + lane < N) cv.W1[64 * c + lane] = p;
        }
        l = wave_sum(l);
        lds_fence_wave();
        if (lane < HDP) {
            const float o = rows_wsum<HDP>(cv.W1, Vs, N, lane) / l;
            if (lane < hd) out[((long)b * N) * E + h * hd + lane] = o;
        }
        if (lane == 0) lse[((long)b * H + h) * N] = m + logf(l);
        return;
    }

    for (int qt = wave; qt < ntile; qt += tile_waves) {
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
        store_rows<HDP>(o, out + ((long)b * N + query) * E + h * hd, qp, qok, hd);
        if (qp == 0 && qok) lse[((long)b * H + h) * N + query] = m + logf(l);
    }
}

// ------------------------------------------------------------------ backward: dQ (+ D = rowsum(dO * O))
template <int HDP, bool EXTRA>
__global__ __launch_bounds__(576) void attn_bwd_dq_kernel(const float* __restrict__ qkv,
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int tile_waves = nwaves - (EXTRA ? 1 : 0);
    const bool extra_wave = EXTRA && wave == tile_waves;
    const int r = lane & 15, qp = lane >> 4;
    float qf[NMM], dof[NMM], of[NMM];
    float lq_first = 0.f;                          // log-sum-exp of the wave's first query row, requested with the fragments
    if (!extra_wave) {
        const int q0 = tok<EXTRA>(wave, r);
        load_frag<HDP>(qf, base + (long)q0 * E3, qp, q0 < N, hd);
        load_frag<HDP>(dof, dout + obase + (long)q0 * E, qp, q0 < N, hd);
        load_frag<HDP>(of, out + obase + (long)q0 * E, qp, q0 < N, hd);
        if (q0 < N) lq_first = lse[((long)b * H + h) * N + q0];
    }
    stage_rows_pair<HDP>(Ks, base + E, E3, Vs, base + 2 * E, E3, N, cv.nrows, hd);
    if (EXTRA) {
        stage_vec<HDP>(cv.X0, base, hd);                               // q of token 0
        stage_vec<HDP>(cv.X1, dout + obase, hd);                       // dO of token 0
    }
    __syncthreads();

    if (extra_wave) {                                                  // dQ of token 0: VALU only
        float d0 = (lane < hd) ? cv.X1[lane] * out[obase + lane] : 0.f;
        const float D0 = wave_sum(d0);
        const long srow = ((long)b * H + h) * N;
        if (lane == 0) delta[srow] = D0;
        const float l0 = lse[srow];
        float s[MAXCH], dp[MAXCH];
        rows_dot<HDP>(s, cv.X0, Ks, N, lane);
        rows_dot<HDP>(dp, cv.X1, Vs, N, lane);
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            if (64 * c + lane < N) {
                const float p = __expf(s[c] * scale - l0);
                cv.W1[64 * c + lane] = p * (dp[c] - D0) * scale;
            }
        }
        lds_fence_wave();
        if (lane < HDP) {
            const float g = rows_wsum<HDP>(cv.W1, Ks, N, lane);
            if (lane < hd) dqkv[(long)b * N * E3 + h * hd + lane] = g;
        }
        return;
    }

    for (int qt = wave; qt < ntile; qt += tile_waves) {
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
        const long srow = ((long)b * H + h) * N + query;
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
    }
}

// ------------------------------------------------------------------ backward: dK, dV
template <int HDP, bool EXTRA>
__global__ __launch_bounds__(576) void attn_bwd_dkv_kernel(const float* __restrict__ qkv,
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
    const int tile_waves = nwaves - (EXTRA ? 1 : 0);
    const bool extra_wave = EXTRA && wave == tile_waves;
    const int r = lane & 15, qp = lane >> 4;
    float kf[NMM], vf[NMM];
    if (!extra_wave) {
        const int k0 = tok<EXTRA>(wave, r);
        load_frag<HDP>(kf, base + (long)k0 * E3 + E, qp, k0 < N, hd);
        load_frag<HDP>(vf, base + (long)k0 * E3 + 2 * E, qp, k0 < N, hd);
    }
    // row statistics: the first blockDim rows are requested before the slice staging (one exposed
    // global latency less), the rest (N > blockDim never happens for the supported shapes) after
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

    if (extra_wave) {                                                  // dK, dV of token 0: VALU only
        float s[MAXCH], dp[MAXCH];
        rows_dot<HDP>(s, cv.X0, Qs, N, lane);                          // s_j = q_j . k_0
        rows_dot<HDP>(dp, cv.X1, Ds, N, lane);                         // dp_j = dO_j . v_0
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const int j = 64 * c + lane;
            if (j < N) {
                const float p = __expf(s[c] * scale - Ls[j]);
                cv.W1[j] = p;
                cv.W2[j] = p * (dp[c] - Es[j]) * scale;
            }
        }
        lds_fence_wave();
        if (lane < HDP) {
            const float gv = rows_wsum<HDP>(cv.W1, Ds, N, lane);
            const float gk = rows_wsum<HDP>(cv.W2, Qs, N, lane);
            if (lane < hd) {
                float* drow = dqkv + (long)b * N * E3 + h * hd;
                drow[E + lane] = gk;
                drow[2 * E + lane] = gv;
            }
        }
        return;
    }

    for (int kt = wave; kt < ntile; kt += tile_waves) {
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
    }
}

// ------------------------------------------------------------------ backward, fused (short sequences)
// dQ and dK/dV in ONE launch when all four slices (K, V, Q, dO) of an (image, head) fit in LDS next to
// each other (N = 65, hd = 64: 72 KB): the slices are staged once, D = rowsum(dO * O) goes from the dQ
// phase to the dK/dV phase through LDS, and the second kernel's launch, staging and prologue
// disappear.  Phase 1 is attn_bwd_dq_kernel's body (waves own query tiles), phase 2
// attn_bwd_dkv_kernel's (waves own key tiles); the token-0 vectors of the EXTRA path are row 0 of the
// staged slices.
template <int HDP, bool EXTRA>
__global__ __launch_bounds__(576) void attn_bwd_fused_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
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
    float* W1 = Es + nrp;
    float* W2 = W1 + nrp;
    const float* base = qkv + (long)b * N * E3 + h * hd;
    const long obase = (long)b * N * E + h * hd;
    const long srow0 = ((long)b * H + h) * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int tile_waves = nwaves - (EXTRA ? 1 : 0);
    const bool extra_wave = EXTRA && wave == tile_waves;
    const int r = lane & 15, qp = lane >> 4;
    float qf[NMM], dof[NMM], of[NMM];
    if (!extra_wave) {
        const int q0 = tok<EXTRA>(wave, r);
        load_frag<HDP>(qf, base + (long)q0 * E3, qp, q0 < N, hd);
        load_frag<HDP>(dof, dout + obase + (long)q0 * E, qp, q0 < N, hd);
        load_frag<HDP>(of, out + obase + (long)q0 * E, qp, q0 < N, hd);
    }
    float l_r = 0.f;
    if ((int)threadIdx.x < N) l_r = lse[srow0 + threadIdx.x];
    stage_rows_quad<HDP>(Ks, base + E, E3, Vs, base + 2 * E, E3, Qs, base, E3, Ds, dout + obase, E, N, nrows);
    for (int i = threadIdx.x; i < nrp; i += blockDim.x) {
        Ls[i] = (i == (int)threadIdx.x) ? l_r : ((i < N) ? lse[srow0 + i] : 0.f);
        Es[i] = 0.f;
    }
    __syncthreads();

    // ---- phase 1: dQ and D
    if (extra_wave) {                                                  // token 0 as a query: VALU only
        const float* q0v = Qs;                                         // row 0 of the staged slices
        const float* do0 = Ds;
        float d0 = (lane < hd) ? do0[lane] * out[obase + lane] : 0.f;
        const float D0 = wave_sum(d0);
        if (lane == 0) { delta[srow0] = D0; Es[0] = D0; }
        const float l0 = Ls[0];
        float sc[MAXCH], dp[MAXCH];
        rows_dot<HDP>(sc, q0v, Ks, N, lane);
        rows_dot<HDP>(dp, do0, Vs, N, lane);
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            if (64 * c + lane < N) {
                const float p = __expf(sc[c] * scale - l0);
                W1[64 * c + lane] = p * (dp[c] - D0) * scale;
            }
        }
        lds_fence_wave();
        if (lane < HDP) {
            const float gq = rows_wsum<HDP>(W1, Ks, N, lane);
            if (lane < hd) dqkv[(long)b * N * E3 + h * hd + lane] = gq;
        }
    } else {
        for (int qt = wave; qt < ntile; qt += tile_waves) {
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
            if (EXTRA) {                                               // token 0 as a key
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
        }
    }
    __syncthreads();                                                   // Es (D of every row) complete; W1 free again

    // ---- phase 2: dK, dV
    if (extra_wave) {                                                  // token 0 as a key: VALU only
        const float* k0v = Ks;
        const float* v0v = Vs;
        float sc[MAXCH], dp[MAXCH];
        rows_dot<HDP>(sc, k0v, Qs, N, lane);                           // s_j = q_j . k_0
        rows_dot<HDP>(dp, v0v, Ds, N, lane);                           // dp_j = dO_j . v_0
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const int j = 64 * c + lane;
            if (j < N) {
                const float p = __expf(sc[c] * scale - Ls[j]);
                W1[j] = p;
                W2[j] = p * (dp[c] - Es[j]) * scale;
            }
        }
        lds_fence_wave();
        if (lane < HDP) {
            const float gv = rows_wsum<HDP>(W1, Ds, N, lane);
            const float gk = rows_wsum<HDP>(W2, Qs, N, lane);
            if (lane < hd) {
                float* drow = dqkv + (long)b * N * E3 + h * hd;
                drow[E + lane] = gk;
                drow[2 * E + lane] = gv;
            }
        }
        return;
    }
    for (int kt = wave; kt < ntile; kt += tile_waves) {
        const int key = tok<EXTRA>(kt, r);
        const bool kok = key < N;
        float kf[NMM], vf[NMM];
        load_frag<HDP>(kf, Ks + (long)(kok ? key : 0) * S, qp, kok, HDP);      // own rows from the staged slices
        load_frag<HDP>(vf, Vs + (long)(kok ? key : 0) * S, qp, kok, HDP);
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
    }
}

// ------------------------------------------------------------------ host side
// (A persistent variant -- workgroups looping over (image, head) items with register prefetch of
// the next item's rows -- was measured and rejected: the extra registers drop residency from 3-4 to
// 2 workgroups per CU and N = 65 got slower, 42 -> 55 us per forward layer.)
static bool use_extra(int N) { return N >= 17 && (N % 16) == 1 && N <= 64 * MAXCH; }
static int attn_tiles(int N) { return use_extra(N) ? (N - 1) / 16 : cdiv(N, 16); }
static int attn_waves(int N) {          // MFMA (tile) waves; EXTRA mode adds one VALU wave
    const int ntile = attn_tiles(N);
    const int rounds = cdiv(ntile, 8);
    return cdiv(ntile, rounds);
}
static int attn_hdp(int hd) {
    if (hd == 16 || hd == 32 || hd == 64) return hd;
    if (hd >= 1 && hd <= 4) return 4;
    if (hd <= 8) return 8;
    return 0;
}
static size_t attn_lds_bytes(int N, int hdp, bool with_stats) {
    const int nrows = use_extra(N) ? N : cdiv(N, 16) * 16;
    const int nrp = (nrows + 3) & ~3;
    return ((size_t)2 * nrows * (hdp + 4) + (with_stats ? 2 * nrp : 0) + 2 * hdp + 2 * nrp) * sizeof(float);
}

static size_t attn_fused_lds_bytes(int N, int hdp) {
    const int nrows = use_extra(N) ? N : cdiv(N, 16) * 16;
    const int nrp = (nrows + 3) & ~3;
    return ((size_t)4 * nrows * (hdp + 4) + 4 * nrp) * sizeof(float);
}

template <int HDP, bool EXTRA>
static int launch_fwd_t(const float* qkv, float* out, float* lse, int B, int N, int H, int hd, hipStream_t st) {
    const size_t lds = attn_lds_bytes(N, HDP, false);
    hipLaunchKernelGGL((attn_fwd_kernel<HDP, EXTRA>), dim3(B * H), dim3(64 * (attn_waves(N) + (EXTRA ? 1 : 0))), lds, st, qkv,
                       out, lse, N, H, hd, 1.0f / sqrtf((float)hd));
    VSOM_LAUNCH_CHECK("attn_fwd_kernel");
}
template <int HDP, bool EXTRA>
static int launch_bwd_t(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                        float* delta, int B, int N, int H, int hd, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)hd);
    const dim3 block(64 * (attn_waves(N) + (EXTRA ? 1 : 0)));
    // all four slices in LDS and still two workgroups per CU -> one fused launch (vector path only)
    const size_t fused_lds = attn_fused_lds_bytes(N, HDP);
    if (ACfg<HDP>::VEC && fused_lds <= 80 * 1024 && g_attn_fused.load(std::memory_order_relaxed)) {
        hipLaunchKernelGGL((attn_bwd_fused_kernel<HDP, EXTRA>), dim3(B * H), block, fused_lds, st, qkv, out, dout, lse, dqkv,
                           delta, N, H, hd, scale);
        VSOM_LAUNCH_CHECK("attn_bwd_fused_kernel");
    }
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HDP, EXTRA>), dim3(B * H), block, attn_lds_bytes(N, HDP, false), st, qkv, out, dout,
                       lse, dqkv, delta, N, H, hd, scale);
    int rc = hip_status(hipGetLastError(), "attn_bwd_dq_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<HDP, EXTRA>), dim3(B * H), block, attn_lds_bytes(N, HDP, true), st, qkv, dout, lse,
                       delta, dqkv, N, H, hd, scale);
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
    VSOM_REQUIRE(attn_lds_bytes(N, *hdp, true) <= 160 * 1024, VSOM_EUNSUPPORTED,
                 "%s: N=%d hd=%d needs %zu B of LDS (> 160 KiB)", who, N, hd, attn_lds_bytes(N, *hdp, true));
    return VSOM_OK;
}

}  // namespace vsom

using namespace vsom;

extern "C" {

int vsom_set_attention_fused(int fused) {
    g_attn_fused.store(fused ? 1 : 0, std::memory_order_relaxed);
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
