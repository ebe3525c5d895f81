// Multi-head attention forward / backward for short ViT sequences (N = 17..257 tokens),
// exact fp32 on v_mfma_f32_16x16x4_f32.
//
// One workgroup per (image, head); the two [N, hd] operands every query/key tile needs stay
// in LDS for the whole workgroup; each wave owns 16-row tiles.  Scores are computed
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
#include "common.h"

namespace vsom {

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

// stage rows [0,N) of a [N, hd] slice (row stride `rs`) into lds[Np][S], zero padded
template <int HDP>
__device__ __forceinline__ void stage_rows(float* lds, const float* __restrict__ src, long rs, int N, int Np,
                                           int hd) {
    constexpr int S = ACfg<HDP>::S;
    if constexpr (ACfg<HDP>::VEC) {
        constexpr int C4 = HDP / 4;
        for (int idx = threadIdx.x; idx < Np * C4; idx += blockDim.x) {
            const int row = idx / C4, c4 = idx % C4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < N) v = *reinterpret_cast<const f32x4*>(src + (long)row * rs + 4 * c4);
            *reinterpret_cast<f32x4*>(lds + row * S + 4 * c4) = v;
        }
    } else {
        for (int idx = threadIdx.x; idx < Np * HDP; idx += blockDim.x) {
            const int row = idx / HDP, c = idx % HDP;
            lds[row * S + c] = (row < N && c < hd) ? src[(long)row * rs + c] : 0.f;
        }
    }
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

// acc[4q'+reg][own row] = sum_d Y[16t + 4q'+reg][d] * own[row][d]
template <int HDP>
__device__ __forceinline__ f32x4 score_tile(const float* Ylds, int t, int r, int qp,
                                            const float (&bf)[ACfg<HDP>::NMM]) {
    float af[ACfg<HDP>::NMM];
    load_frag<HDP>(af, Ylds + (16 * t + r) * ACfg<HDP>::S, qp, true, HDP);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mm = 0; mm < ACfg<HDP>::NMM; ++mm) acc = mfma16(af[mm], bf[mm], acc);
    return acc;
}

// o[dt][d = 16dt + 4q'+reg][own row] += sum_{j in tile t} Z[j][d] * p[j][own row]
template <int HDP>
__device__ __forceinline__ void accum_tile(f32x4 (&o)[ACfg<HDP>::NDT], const float* Zlds, int t, int r, int qp,
                                           f32x4 p) {
    constexpr int S = ACfg<HDP>::S;
#pragma unroll
    for (int dt = 0; dt < ACfg<HDP>::NDT; ++dt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float a = 0.f;
            if (ACfg<HDP>::VEC || 16 * dt + r < HDP) a = Zlds[(16 * t + 4 * qp + s) * S + 16 * dt + r];
            o[dt] = mfma16(a, p[s], o[dt]);
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

// ------------------------------------------------------------------ forward
template <int HDP>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                       float* __restrict__ lse, int N, int H, int hd,
                                                       float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = ACfg<HDP>::S;
    constexpr int NDT = ACfg<HDP>::NDT;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * hd, E3 = 3 * E;
    const int ntile = (N + 15) >> 4, Np = ntile << 4;
    float* Ks = smem;
    float* Vs = smem + Np * S;
    const float* base = qkv + (long)b * N * E3 + h * hd;
    stage_rows<HDP>(Ks, base + E, E3, N, Np, hd);
    stage_rows<HDP>(Vs, base + 2 * E, E3, N, Np, hd);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, qp = lane >> 4;
    for (int qt = wave; qt < ntile; qt += nwaves) {
        const int query = 16 * qt + r;
        const bool qok = query < N;
        float qf[ACfg<HDP>::NMM];
        load_frag<HDP>(qf, base + (long)query * E3, qp, qok, hd);
        float m = -INFINITY, l = 0.f;
        f32x4 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < ntile; c0 += 4) {
            f32x4 s[4];
            float cmax = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = c0 + tt;
                if (t < ntile) {
                    s[tt] = score_tile<HDP>(Ks, t, r, qp, qf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = 16 * t + 4 * qp + e;
                        s[tt][e] = (key < N) ? s[tt][e] * scale : -INFINITY;
                        cmax = fmaxf(cmax, s[tt][e]);
                    }
                } else {
                    s[tt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
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
                if (c0 + tt < ntile) accum_tile<HDP>(o, Vs, c0 + tt, r, qp, s[tt]);
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] *= inv;
        store_rows<HDP>(o, out + ((long)b * N + query) * E + h * hd, qp, qok, hd);
        if (qp == 0 && qok) lse[((long)b * H + h) * N + query] = m + logf(l);
    }
}

// ------------------------------------------------------------------ backward: dQ (+ D = rowsum(dO * O))
template <int HDP>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(const float* __restrict__ qkv,
                                                          const float* __restrict__ out,
                                                          const float* __restrict__ dout,
                                                          const float* __restrict__ lse,
                                                          float* __restrict__ dqkv, float* __restrict__ delta,
                                                          int N, int H, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = ACfg<HDP>::S;
    constexpr int NDT = ACfg<HDP>::NDT;
    constexpr int NMM = ACfg<HDP>::NMM;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * hd, E3 = 3 * E;
    const int ntile = (N + 15) >> 4, Np = ntile << 4;
    float* Ks = smem;
    float* Vs = smem + Np * S;
    const float* base = qkv + (long)b * N * E3 + h * hd;
    stage_rows<HDP>(Ks, base + E, E3, N, Np, hd);
    stage_rows<HDP>(Vs, base + 2 * E, E3, N, Np, hd);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, qp = lane >> 4;
    for (int qt = wave; qt < ntile; qt += nwaves) {
        const int query = 16 * qt + r;
        const bool qok = query < N;
        float qf[NMM], dof[NMM], of[NMM];
        load_frag<HDP>(qf, base + (long)query * E3, qp, qok, hd);
        const long orow = ((long)b * N + query) * E + h * hd;
        load_frag<HDP>(dof, dout + orow, qp, qok, hd);
        load_frag<HDP>(of, out + orow, qp, qok, hd);
        float D = 0.f;
#pragma unroll
        for (int mm = 0; mm < NMM; ++mm) D = fmaf(dof[mm], of[mm], D);
        D = group_sum(D);
        const long srow = ((long)b * H + h) * N + query;
        if (qp == 0 && qok) delta[srow] = D;
        const float lq = qok ? lse[srow] : 0.f;
        f32x4 dq[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < ntile; ++t) {
            const f32x4 s = score_tile<HDP>(Ks, t, r, qp, qf);
            const f32x4 dp = score_tile<HDP>(Vs, t, r, qp, dof);
            f32x4 ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = 16 * t + 4 * qp + e;
                const float p = (key < N && qok) ? __expf(s[e] * scale - lq) : 0.f;
                ds[e] = p * (dp[e] - D) * scale;
            }
            accum_tile<HDP>(dq, Ks, t, r, qp, ds);
        }
        store_rows<HDP>(dq, dqkv + ((long)b * N + query) * E3 + h * hd, qp, qok, hd);
    }
}

// ------------------------------------------------------------------ backward: dK, dV
template <int HDP>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(const float* __restrict__ qkv,
                                                           const float* __restrict__ dout,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta,
                                                           float* __restrict__ dqkv, int N, int H, int hd,
                                                           float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = ACfg<HDP>::S;
    constexpr int NDT = ACfg<HDP>::NDT;
    constexpr int NMM = ACfg<HDP>::NMM;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int E = H * hd, E3 = 3 * E;
    const int ntile = (N + 15) >> 4, Np = ntile << 4;
    float* Qs = smem;
    float* Ds = smem + Np * S;
    float* Ls = smem + 2 * Np * S;
    float* Es = Ls + Np;
    const float* base = qkv + (long)b * N * E3 + h * hd;
    stage_rows<HDP>(Qs, base, E3, N, Np, hd);
    stage_rows<HDP>(Ds, dout + (long)b * N * E + h * hd, E, N, Np, hd);
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
        const long srow = ((long)b * H + h) * N + i;
        Ls[i] = (i < N) ? lse[srow] : 0.f;
        Es[i] = (i < N) ? delta[srow] : 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane & 15, qp = lane >> 4;
    for (int kt = wave; kt < ntile; kt += nwaves) {
        const int key = 16 * kt + r;
        const bool kok = key < N;
        float kf[NMM], vf[NMM];
        load_frag<HDP>(kf, base + (long)key * E3 + E, qp, kok, hd);
        load_frag<HDP>(vf, base + (long)key * E3 + 2 * E, qp, kok, hd);
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int t = 0; t < ntile; ++t) {
            const f32x4 s = score_tile<HDP>(Qs, t, r, qp, kf);      // rows: queries of tile t, col: own key
            const f32x4 dp = score_tile<HDP>(Ds, t, r, qp, vf);
            f32x4 p, ds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int query = 16 * t + 4 * qp + e;
                p[e] = (query < N && kok) ? __expf(s[e] * scale - Ls[query]) : 0.f;
                ds[e] = p[e] * (dp[e] - Es[query]) * scale;
            }
            accum_tile<HDP>(dv, Ds, t, r, qp, p);
            accum_tile<HDP>(dk, Qs, t, r, qp, ds);
        }
        float* drow = dqkv + ((long)b * N + key) * E3 + h * hd;
        store_rows<HDP>(dk, drow + E, qp, kok, hd);
        store_rows<HDP>(dv, drow + 2 * E, qp, kok, hd);
    }
}

static int attn_waves(int N) {
    const int ntile = cdiv(N, 16);
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
    const int Np = cdiv(N, 16) * 16;
    return ((size_t)2 * Np * (hdp + 4) + (with_stats ? 2 * Np : 0)) * sizeof(float);
}

template <int HDP>
static int launch_fwd(const float* qkv, float* out, float* lse, int B, int N, int H, int hd, hipStream_t st) {
    const size_t lds = attn_lds_bytes(N, HDP, false);
    hipLaunchKernelGGL(attn_fwd_kernel<HDP>, dim3(B * H), dim3(64 * attn_waves(N)), lds, st, qkv, out, lse, N, H, hd,
                       1.0f / sqrtf((float)hd));
    VSOM_LAUNCH_CHECK("attn_fwd_kernel");
}
template <int HDP>
static int launch_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                      float* delta, int B, int N, int H, int hd, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)hd);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<HDP>, dim3(B * H), dim3(64 * attn_waves(N)), attn_lds_bytes(N, HDP, false),
                       st, qkv, out, dout, lse, dqkv, delta, N, H, hd, scale);
    int rc = hip_status(hipGetLastError(), "attn_bwd_dq_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<HDP>, dim3(B * H), dim3(64 * attn_waves(N)), attn_lds_bytes(N, HDP, true),
                       st, qkv, dout, lse, delta, dqkv, N, H, hd, scale);
    VSOM_LAUNCH_CHECK("attn_bwd_dkv_kernel");
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
