// Row-panel Linear GEMMs on pre-split weight images ("rp" engine; the default for every nn.Linear-shaped forward / input-gradient
// GEMM whose shape it takes -- vsom_linear_planes).
//
//   Y[M,N] = epilogue( A[M,K] . B[K,N] ),   A = X or LayerNorm(X) (fp32 activations),   B = a weight (or its transpose)
//
// Same arithmetic as gemm_x6.h -- every fp32 operand is split EXACTLY into three bf16 pieces and the six leading cross
// products are accumulated on v_mfma_f32_32x32x16_bf16, smallest first -- but the data movement is turned around:
//
//  * a wave OWNS 32 rows of the output.  Its A operand never touches LDS: the lane (r = lane & 31, h = lane >> 5) loads
//    the 8 consecutive floats of row r it feeds to the MFMA (k = 16 s + 8 h .. + 8) straight into registers and splits them
//    there -- once per element per workgroup, instead of once per 64-column tile of the output (9 times for the qkv GEMM)
//    plus an LDS round trip;
//  * the weight operand is split ONCE PER OPTIMIZER STEP by weight_image_kernel into an "image" of MFMA B fragments
//    (3 planes x [K/16 steps] x [N/32 tiles] x 64 lanes x 16 bytes, in the order the kernel consumes them), so that a
//    workgroup streams it global -> LDS with LDS-DMA (buffer_load ... lds: no registers, no VALU, no bank conflicts --
//    a fragment is 1 KB, lane-linear) and reads it back with one ds_read_b128 per plane;
//  * "panel" form (K <= 192, e.g. qkv / proj / fc1 and the input gradient of fc2): the whole K extent of the wave's 32
//    rows stays in registers (K/16 x 3 planes x 4 VGPRs = 144 at K = 192) while the workgroup walks the 32-column tiles
//    of the weight; because the whole row is there, LayerNorm is applied in this prologue (row statistics from the
//    registers, mean / rstd saved for the backward): no LayerNorm kernel, no normalised copy of the activations;
//  * "stream" form (N <= 192, any K % 32 == 0, e.g. fc2 and the input gradients of fc1 / qkv): the wave holds the 32 x N
//    accumulators and walks K, A fragments prefetched one k-tile ahead.
// Both forms: WAVES x 32 rows per workgroup, one 36 KB block of the image per step (double-buffered, one barrier per
// step), 72 MFMAs per wave per step at K or N = 192.  Epilogues are gemm_f32.h's (a guard-free copy for whole tiles).
#include "gemm_f32.h"
#include "gemm_x6.h"

#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>

namespace vsom {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int RP_FRAG = 1024;          // one B fragment: 64 lanes x 16 bytes

struct RpP {
    GemmP g;                 // A (= X) / lda, C / ldc, bias, R, C2, accumulate, M, N, K for the shared epilogue
    const char* img;         // weight image
    unsigned img_bytes;
    int nt_total;            // 32-column tiles of the output
    int nsplit, tiles_per_split;    // panel form: column ranges per row panel
    const float* gamma; const float* beta; float eps;       // LayerNorm prologue (panel form) or null
    float* mean; float* rstd;
    float* ln_out; long ld_ln;                               // optional copy of LayerNorm(X) (the weight-gradient GEMM reads it)
    int ablate;
    unsigned long long* stamps;   // lab only (VSOM_RP_STAMPS): 8 words per wave, see rp_stamp_report
};

__device__ __forceinline__ bf16x8 rp_pack(uint2 lo, uint2 hi) {
    const uint4 v = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(bf16x8, v);
}
// 8 consecutive floats -> the three bf16 planes of an MFMA A fragment
__device__ __forceinline__ void rp_split8(f32x4 lo, f32x4 hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    uint2 a1, a2, a3, b1, b2, b3;
    x6_split(lo, a1, a2, a3);
    x6_split(hi, b1, b2, b3);
    p1 = rp_pack(a1, b1); p2 = rp_pack(a2, b2); p3 = rp_pack(a3, b3);
}

__device__ __forceinline__ f32x16 rp_mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);   // 2^-16 terms
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);   // 2^-8 terms
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);   // leading term
    return c;
}

// The three planes of B fragment number F of a block (lane-linear: conflict-free ds_read_b128).  Written in assembly, with the
// waits placed by hand (rp_frags_wait): left to hipcc, every read is sunk to just before the MFMA that consumes it -- however
// the source orders them, sched_group_barrier included -- and the dependent MFMA chain eats the LDS latency once per step.
template <int F>
__device__ __forceinline__ void rp_frags(bf16x8 (&bf)[3], unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6"
                 : "=&v"(bf[0]), "=&v"(bf[1]), "=&v"(bf[2])
                 : "v"(lds_addr), "n"(F * 3 * RP_FRAG), "n"((F * 3 + 1) * RP_FRAG), "n"((F * 3 + 2) * RP_FRAG));
}
// wait until at most N LDS reads of this wave are outstanding (they return in order), then fence: an MFMA does not touch
// memory, so nothing else keeps hipcc from hoisting it above the wait
template <int N>
__device__ __forceinline__ void rp_frags_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// compile-time loop (the fragment number is an immediate of the read)
template <int I, int N, typename F>
__device__ __forceinline__ void rp_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rp_static_for<I + 1, N>(f);
    }
}

// LDS-DMA of one image block (NPIECE fragments of 1 KB, contiguous in the image) by the 8 waves of the workgroup.
// Reads past the end of the image (the prefetch of a block that does not exist) are range-checked away.
// The instruction is written in assembly so that hipcc does not know it writes LDS: with the builtin form it orders every
// later ds_read behind the transfer (s_waitcnt vmcnt(0) right after the issue -- the whole latency exposed once per step).
// The waits are ours: rp_wait_block() before the barrier that precedes the first read of the block.
typedef int rp_i32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor (base, stride 0, num_records = bytes, DATA_FORMAT = 32: the flags word of make_buffer_rsrc above)
__device__ __forceinline__ rp_i32x4 rp_srd(const void* base, unsigned bytes) {
    const unsigned long a = (unsigned long)base;
    return rp_i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void rp_dma16(rp_i32x4 rs, unsigned lds_dst, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_dst), "s"(rs), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ unsigned rp_lds_addr(const char* p) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)p;
}
template <int NPIECE, int WAVES>
__device__ __forceinline__ void rp_dma_block(rp_i32x4 rs, char* dst, unsigned block_off, int wave, int lane) {
    const unsigned base = rp_lds_addr(dst);
#pragma unroll
    for (int i = 0; i < NPIECE / WAVES; ++i) {
        const int q = i * WAVES + wave;
        rp_dma16(rs, base + q * RP_FRAG, (unsigned)(q * RP_FRAG + lane * 16), block_off);
    }
    if constexpr (NPIECE % WAVES != 0) {
        const int q = (NPIECE / WAVES) * WAVES + wave;
        if (q < NPIECE) rp_dma16(rs, base + q * RP_FRAG, (unsigned)(q * RP_FRAG + lane * 16), block_off);
    }
}
// Wait until this wave's pieces of the block issued one step ago have landed.  `younger` = the vector-memory instructions the
// wave is KNOWN to have issued after them (the epilogue stores of a tile that lies wholly inside the matrix): they may stay in
// flight.  A wave that cannot vouch for that count waits for everything.
template <int YOUNGER>
__device__ __forceinline__ void rp_wait_block(bool counted) {
    if (counted) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(YOUNGER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// gemm_epilogue's whole-tile path for ONE 32 x 32 tile, cut into row ranges so that the panel kernel can issue the stores of
// tile t under the MFMAs of tile t + 1 (a wave's stores then overlap its own matrix work: no other wave has to be in a
// different phase for the memory system and the matrix cores to be busy together).  Accumulator register v holds row
// (v & 3) + 8 (v >> 2) + 4 h of the tile, column r.  `bn` = bias of the column, `rv` = the 16 values of the R operand
// (residual / gelu'), both loaded a tile early.
__device__ __forceinline__ long rp_row_of(int v) { return (v & 3) + 8 * (v >> 2); }
template <int EPI>
__device__ __forceinline__ void rp_load_r(const GemmP& g, float (&rv)[16], int mb, int n) {
    const float* q = g.R + (long)mb * g.ldr + n;
#pragma unroll
    for (int v = 0; v < 16; ++v) rv[v] = q[rp_row_of(v) * g.ldr];
}
template <int EPI, int V0, int V1>
__device__ __forceinline__ void rp_store_rows(const GemmP& g, const f32x16& acc, const float (&rv)[16], int mb, int n, float bn) {
#pragma unroll
    for (int v = V0; v < V1; ++v) {
        const float a = acc[v];
        float* d = g.C + ((long)mb + rp_row_of(v)) * g.ldc + n;
        if constexpr (EPI == EPI_NONE) {
            *d = a;
        } else if constexpr (EPI == EPI_BIAS) {
            *d = a + bn;
        } else if constexpr (EPI == EPI_BIAS_GELU) {
            float act, grad;
            gelu_erf_both(a + bn, act, grad);
            *d = grad;
            g.C2[((long)mb + rp_row_of(v)) * g.ldc2 + n] = act;
        } else if constexpr (EPI == EPI_BIAS_RELU) {
            const float pre = a + bn;
            *d = pre > 0.f ? 1.0f : 0.f;
            g.C2[((long)mb + rp_row_of(v)) * g.ldc2 + n] = fmaxf(pre, 0.f);
        } else if constexpr (EPI == EPI_BIAS_RES) {
            *d = a + bn + rv[v];
        } else if constexpr (EPI == EPI_GELU_BWD) {
            *d = a * rv[v];
        }
    }
}

// ------------------------------------------------------------------------------------------------ panel form
__device__ __forceinline__ unsigned long long rp_clock() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
__device__ __forceinline__ unsigned long long rp_realtime() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

template <int KS, int EPI, bool LN, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void linear_rp_panel_kernel(const RpP p) {
    constexpr int RP_ROWS = 32 * WAVES;
    const bool st = p.stamps != nullptr;
    unsigned long long t_real0 = 0, t0 = 0, t1 = 0, t_wait = 0, t_mma = 0, t_a = 0;
    if (st) { t_real0 = rp_realtime(); t0 = rp_clock(); }
    constexpr int NPIECE = KS * 3;
    constexpr int BLK = NPIECE * RP_FRAG;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int panel = lid / p.nsplit, split = lid - panel * p.nsplit;
    const int m0 = panel * RP_ROWS + wave * 32;
    const int nt_begin = split * p.tiles_per_split;
    int nt_end = nt_begin + p.tiles_per_split;
    if (nt_end > p.nt_total) nt_end = p.nt_total;

    const rp_i32x4 rsI = rp_srd(p.img, p.img_bytes);
    rp_dma_block<NPIECE, WAVES>(rsI, lds, (unsigned)nt_begin * BLK, wave, lane);

    // ---- A: this lane's share of row m0 + r (k = 16 s + 8 h .. + 8 for every step s), split once
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g.A), 0, (int)p.g.a_bytes, 0x00020000);
    const int row = m0 + r;
    const unsigned rowoff = row < p.g.M ? (unsigned)(((long)row * p.g.lda + 8 * h) << 2) : OOB;
    bf16x8 a[KS][3];
    {
        f32x4 raw[KS][2];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            raw[s][0] = bload4(rsX, rowoff != OOB ? rowoff + 64 * s : OOB);
            raw[s][1] = bload4(rsX, rowoff != OOB ? rowoff + 64 * s + 16 : OOB);
        }
        if constexpr (LN) {
            const float inv_k = 1.0f / (float)(16 * KS);
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                sum += ((raw[s][0][0] + raw[s][0][1]) + (raw[s][0][2] + raw[s][0][3])) + ((raw[s][1][0] + raw[s][1][1]) + (raw[s][1][2] + raw[s][1][3]));
            sum += __shfl_xor(sum, 32, 64);
            const float mu = sum * inv_k;
            float q = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = raw[s][c][e] - mu; q = fmaf(d, d, q); }
            q += __shfl_xor(q, 32, 64);
            const float rs = rsqrtf(q * inv_k + p.eps);
            if (split == 0 && h == 0 && row < p.g.M) { p.mean[row] = mu; p.rstd[row] = rs; }
            const f32x4* g4 = reinterpret_cast<const f32x4*>(p.gamma) + 2 * h;
            const f32x4* b4 = reinterpret_cast<const f32x4*>(p.beta) + 2 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x4 gg = g4[4 * s + c], bb = b4[4 * s + c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) raw[s][c][e] = (raw[s][c][e] - mu) * rs * gg[e] + bb[e];
                }
            if (p.ln_out != nullptr && split == 0 && row < p.g.M) {
                f32x4* o = reinterpret_cast<f32x4*>(p.ln_out + (long)row * p.ld_ln + 8 * h);
#pragma unroll
                for (int s = 0; s < KS; ++s) { o[4 * s] = raw[s][0]; o[4 * s + 1] = raw[s][1]; }
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) rp_split8(raw[s][0], raw[s][1], a[s][0], a[s][1], a[s][2]);
    }

    if (st) { asm volatile("" :: "v"(a[0][0]), "v"(a[KS - 1][2])); t1 = rp_clock(); }
    // ---- walk the column tiles: block nt of the image = [K/16 steps][3 planes][64 lanes][16 B]
    // The epilogue of a whole tile is DEFERRED: its accumulators are parked and its stores go out one row group per step under
    // the MFMAs of the next tile.  Vector-memory instructions this wave issues after the DMA of a step, all of known count:
    // the parked tile's stores, then the next R operand's 16 loads -- they may stay in flight when the next step begins
    // (rp_wait_block).  Tiles cut by the matrix edge (or accumulating into the output) take gemm_epilogue's guarded path at
    // once and the wave then waits for everything.
    constexpr int EPI_STORES = (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU) ? 32 : 16;
    constexpr bool HAS_BIAS = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_RES);
    constexpr bool READS_R = (EPI == EPI_GELU_BWD || EPI == EPI_BIAS_RES);
    constexpr int PF = (KS >= 12) ? 1 : 2;               // B fragments this many steps ahead of their MFMAs (register budget at K = 192)
    const bool whole_rows = __builtin_amdgcn_readfirstlane(m0 + 31 < p.g.M) && !(EPI == EPI_BIAS_RES && !(p.g.r_mod >= p.g.M)) &&
                            !((EPI == EPI_NONE || EPI == EPI_GELU_BWD) && p.g.accumulate);
    auto bias_of = [&](int nt) { return (HAS_BIAS && p.g.bias && nt * 32 + r < p.g.N) ? p.g.bias[nt * 32 + r] : 0.f; };
    const int mb = m0 + 4 * h;
    float bn = bias_of(nt_begin), pbn = 0.f;
    f32x16 pacc;                                         // the parked tile
    float rv[16];
    int pn = 0;
    bool parked = false;
    int younger = 0;                                     // known vector-memory instructions issued after the last DMA
#pragma unroll
    for (int v = 0; v < 16; ++v) { pacc[v] = 0.f; rv[v] = 0.f; }
    for (int nt = nt_begin; nt < nt_end; ++nt) {
        const int b = (nt - nt_begin) & 1;
        if (st) t_a = rp_clock();
        if (younger == EPI_STORES + 16) rp_wait_block<EPI_STORES + 16>(true);
        else if (younger == EPI_STORES) rp_wait_block<EPI_STORES>(true);
        else if (younger == 16) rp_wait_block<16>(true);
        else rp_wait_block<0>(false);
        __builtin_amdgcn_s_barrier();                         // everybody's pieces have landed; everybody is done reading the other buffer
        if (st) { const unsigned long long t = rp_clock(); t_wait += t - t_a; t_a = t; }
        const float bn_next = bias_of(nt + 1 < nt_end ? nt + 1 : nt);      // before the DMA: hipcc's own wait for it then never waits for the DMA
        rp_dma_block<NPIECE, WAVES>(rsI, lds + (b ^ 1) * BLK, (unsigned)(nt + 1) * BLK, wave, lane);
        younger = parked ? EPI_STORES : 0;
        const unsigned bb = rp_lds_addr(lds) + b * BLK + lane * 16;
        f32x16 acc[1][1];
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[0][0][v] = 0.f;
        bf16x8 bf[PF + 1][3];
        rp_frags<0>(bf[0], bb);
        if constexpr (KS > 1 && PF > 1) rp_frags<1>(bf[1], bb);
        rp_static_for<0, KS>([&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s + PF < KS) { rp_frags<s + PF>(bf[(s + PF) % (PF + 1)], bb); rp_frags_wait<3 * PF>(); }
            else if constexpr (s + 1 < KS) rp_frags_wait<3 * (KS - 1 - s)>();
            else rp_frags_wait<0>();
            acc[0][0] = rp_mfma6(a[s], bf[s % (PF + 1)], acc[0][0]);
            if (parked) rp_store_rows<EPI, (16 * s) / KS, (16 * (s + 1)) / KS>(p.g, pacc, rv, mb, pn, pbn);
        });
        if (st) { asm volatile("" :: "v"(acc[0][0])); t_mma += rp_clock() - t_a; }
        if (whole_rows && nt * 32 + 32 <= p.g.N) {
            pacc = acc[0][0]; pn = nt * 32 + r; pbn = bn; parked = true;
            if constexpr (READS_R) { rp_load_r<EPI>(p.g, rv, mb, pn); younger += 16; }
        } else {
            parked = false; younger = -1;
            gemm_epilogue<1, 1, EPI>(p.g, acc, m0, nt * 32, r, h, 0);
        }
        bn = bn_next;
    }
    if (parked) rp_store_rows<EPI, 0, 16>(p.g, pacc, rv, mb, pn, pbn);
    if (st && lane == 0) {
        unsigned long long* o = p.stamps + ((long)blockIdx.x * WAVES + wave) * 8;
        const unsigned long long t3 = rp_clock();
        o[0] = t_real0; o[1] = t1 - t0; o[2] = t3 - t1; o[3] = t_wait; o[4] = t_mma; o[5] = rp_realtime();
        o[6] = __builtin_amdgcn_s_getreg((3 << 11) | 4) /* HW_ID */; o[7] = nt_end - nt_begin;
    }
}

// ------------------------------------------------------------------------------------------------ stream form
template <int NT, int EPI, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void linear_rp_stream_kernel(const RpP p) {
    constexpr int RP_ROWS = 32 * WAVES;
    constexpr int NPIECE = NT * 6;
    constexpr int BLK = NPIECE * RP_FRAG;             // one k-tile (32 k) of the image: [N/32 tiles][2 steps][3 planes]
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0 = (int)blockIdx.x * RP_ROWS + wave * 32;
    const int ktiles = p.g.K >> 5;

    const rp_i32x4 rsI = rp_srd(p.img, p.img_bytes);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g.A), 0, (int)p.g.a_bytes, 0x00020000);
    const int row = m0 + r;
    const unsigned rowoff = row < p.g.M ? (unsigned)(((long)row * p.g.lda + 8 * h) << 2) : OOB;
    // this lane's A floats of a k-tile: k = 32 kt + 16 s + 8 h + 4 c .. + 4  ->  raw[2 s + c]; two k-tiles in flight (the rows
    // stream from HBM: one step of MFMAs does not cover that latency under load)
    struct Raw { f32x4 v[4]; };
    auto aload = [&](Raw& w, int kt) {
        const unsigned ko = (unsigned)kt << 7;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w.v[i] = bload4(rsX, (rowoff != OOB && kt < ktiles) ? rowoff + ko + 64 * (i >> 1) + 16 * (i & 1) : OOB);
    };
    Raw r0, r1;
    aload(r0, 0);
    rp_dma_block<NPIECE, WAVES>(rsI, lds, 0u, wave, lane);
    aload(r1, 1);

    f32x16 acc[1][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[0][j][v] = 0.f;

    auto step = [&](Raw& cur, int kt) {               // consumes cur (k-tile kt), refills it with k-tile kt + 2
        const int b = kt & 1;
        // block kt of the image has landed; the 4 loads of k-tile kt + 1 (issued after it) may stay in flight
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bf16x8 a[2][3];
        rp_split8(cur.v[0], cur.v[1], a[0][0], a[0][1], a[0][2]);
        rp_split8(cur.v[2], cur.v[3], a[1][0], a[1][1], a[1][2]);
        rp_dma_block<NPIECE, WAVES>(rsI, lds + (b ^ 1) * BLK, (unsigned)(kt + 1) * BLK, wave, lane);
        aload(cur, kt + 2);
        const unsigned bb = rp_lds_addr(lds) + b * BLK + lane * 16;
        bf16x8 bf[3][3];                              // fragment f = 2 j + s of the block, two ahead of its MFMAs
        rp_frags<0>(bf[0], bb);
        rp_frags<1>(bf[1], bb);
        rp_static_for<0, 2 * NT>([&](auto F) {
            constexpr int f = decltype(F)::value;
            if constexpr (f + 2 < 2 * NT) { rp_frags<f + 2>(bf[(f + 2) % 3], bb); rp_frags_wait<6>(); }
            else if constexpr (f + 1 < 2 * NT) rp_frags_wait<3>();
            else rp_frags_wait<0>();
            acc[0][f >> 1] = rp_mfma6(a[f & 1], bf[f % 3], acc[0][f >> 1]);
        });
    };
    int kt = 0;
    for (; kt + 1 < ktiles; kt += 2) { step(r0, kt); step(r1, kt + 1); }
    if (kt < ktiles) step(r0, kt);
    gemm_epilogue<1, NT, EPI>(p.g, acc, m0, 0, r, h, 0);
}

// ------------------------------------------------------------------------------------------------ weight images
// One thread = one lane of one fragment (tile nt, step ks): B[k = 16 ks + 8 h + j][n = 32 nt + r], j = 0..7, from
//   transpose == 0:  B[k][n] = W[n][k]   (forward: W is [N_out, K_red] row-major)
//   transpose == 1:  B[k][n] = W[k][n]   (input gradient: W is [K_red, N_out] row-major)
// split into the three planes and stored at fragment index
//   panel order : (nt * KS + ks) * 3 + pl                      (KS = K_red / 16 steps)
//   stream order: ((kt * NT + nt) * 2 + (ks & 1)) * 3 + pl     (kt = ks / 2, NT = ceil(N_out / 32))
// Elements outside the matrix are zero.  table rows: {src_off (floats), dst_off (bytes), n_out, k_red, transpose, kind}.
__global__ __launch_bounds__(256) void weight_image_kernel(const float* __restrict__ base, char* __restrict__ out,
                                                           const long long* __restrict__ table) {
    const long long* e = table + (long)blockIdx.y * 6;
    const float* W = base + e[0];
    char* img = out + e[1];
    const int n_out = (int)e[2], k_red = (int)e[3], transpose = (int)e[4], kind = (int)e[5];
    const int NT = (n_out + 31) >> 5;
    const int KS = kind == 2 ? ((k_red + 31) >> 5) * 2 : (k_red + 15) >> 4;
    const int frag = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frag >= NT * KS) return;
    const int nt = frag / KS, ks = frag - nt * KS;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int n = 32 * nt + r, k0 = 16 * ks + 8 * h;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    if (n < n_out) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            const float v = k < k_red ? (transpose ? W[(long)k * n_out + n] : W[(long)n * k_red + k]) : 0.f;
            if (j < 4) lo[j] = v; else hi[j - 4] = v;
        }
    }
    uint2 a1, a2, a3, b1, b2, b3;
    x6_split(lo, a1, a2, a3);
    x6_split(hi, b1, b2, b3);
    const long f0 = kind == 2 ? ((long)((ks >> 1) * NT + nt) * 2 + (ks & 1)) * 3 : ((long)nt * KS + ks) * 3;
    uint4* dst = reinterpret_cast<uint4*>(img + f0 * RP_FRAG + lane * 16);
    dst[0] = uint4{a1.x, a1.y, b1.x, b1.y};
    dst[RP_FRAG / 16] = uint4{a2.x, a2.y, b2.x, b2.y};
    dst[2 * RP_FRAG / 16] = uint4{a3.x, a3.y, b3.x, b3.y};
}

// 1 = panel form, 2 = stream form, 0 = not taken by this engine
static int rp_kind(int n_out, int k_red) {
    if (n_out <= 0 || k_red <= 0) return 0;
    if (k_red == 192 || k_red == 96 || k_red == 48) return 1;
    if (k_red % 32 == 0 && (n_out == 192 || n_out == 96)) return 2;
    return 0;
}
static long rp_image_bytes(int n_out, int k_red) {
    const int kind = rp_kind(n_out, k_red);
    if (!kind) return 0;
    const long NT = (n_out + 31) / 32;
    const long KS = kind == 2 ? (long)((k_red + 31) / 32) * 2 : (k_red + 15) / 16;
    return NT * KS * 3 * RP_FRAG;
}

static int rp_env(const char* name, int dflt) {
    const char* e = getenv(name);                      // lab / measurement overrides
    return (e && atoi(e) > 0) ? atoi(e) : dflt;
}
static int rp_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t pr;
            if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
        }
    }
    return cus;
}
static int rp_panel_splits(int M, int rows, int nt_total, int ks, bool ln) {
    const int forced = rp_env("VSOM_RP_SPLITS", 0);
    if (forced) return forced > nt_total ? nt_total : forced;
    // workgroups resident per CU: 512 threads -> 1, 256 threads -> 2; rounds x (tiles per workgroup + prologue, in units of one
    // tile's MFMA time)
    const int slots = rp_cus() * (rows == 256 ? 1 : 2);
    const double pro = (ln ? 3.0 : 2.5) * 12.0 / ks;
    const int panels = cdiv(M, rows);
    double best = 1e30;
    int best_s = 1;
    for (int s = 1; s <= nt_total; ++s) {
        const int tps = cdiv(nt_total, s);
        if (cdiv(nt_total, tps) != s) continue;
        const double cost = (double)cdiv((long)panels * s, slots) * (tps + pro);
        if (cost < best - 1e-9) { best = cost; best_s = s; }
    }
    return best_s;
}

template <int KS, int EPI, int WAVES>
static int launch_panel(RpP& p, hipStream_t st) {
    const dim3 grid(cdiv(p.g.M, 32 * WAVES) * p.nsplit), block(64 * WAVES);
    const size_t lds = 2 * KS * 3 * RP_FRAG;
    if (p.gamma) hipLaunchKernelGGL((linear_rp_panel_kernel<KS, EPI, true, WAVES>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((linear_rp_panel_kernel<KS, EPI, false, WAVES>), grid, block, lds, st, p);
    VSOM_LAUNCH_CHECK("linear_rp_panel_kernel");
}
template <int EPI, int WAVES>
static int launch_panel_k(RpP& p, hipStream_t st) {
    switch (p.g.K) {
        case 192: return launch_panel<12, EPI, WAVES>(p, st);
        case 96: return launch_panel<6, EPI, WAVES>(p, st);
        default: return launch_panel<3, EPI, WAVES>(p, st);
    }
}
template <int NT, int EPI, int WAVES>
static int launch_stream(RpP& p, hipStream_t st) {
    hipLaunchKernelGGL((linear_rp_stream_kernel<NT, EPI, WAVES>), dim3(cdiv(p.g.M, 32 * WAVES)), dim3(64 * WAVES), 2 * NT * 6 * RP_FRAG, st, p);
    VSOM_LAUNCH_CHECK("linear_rp_stream_kernel");
}
static int rp_waves() { return rp_env("VSOM_RP_WAVES", 4) == 8 ? 8 : 4; }
template <int EPI>
static int launch_rp(RpP& p, int kind, hipStream_t st) {
    const bool w8 = rp_waves() == 8;
    if (kind == 1) return w8 ? launch_panel_k<EPI, 8>(p, st) : launch_panel_k<EPI, 4>(p, st);
    if (p.g.N == 192) return w8 ? launch_stream<6, EPI, 8>(p, st) : launch_stream<6, EPI, 4>(p, st);
    return w8 ? launch_stream<3, EPI, 8>(p, st) : launch_stream<3, EPI, 4>(p, st);
}

static void rp_stamp_report(unsigned long long* dev, int nwaves, hipStream_t st) {
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)nwaves * 8);
    (void)hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long r0 = ~0ull, r1 = 0;
    for (int w = 0; w < nwaves; ++w) { r0 = std::min(r0, h[w * 8]); r1 = std::max(r1, h[w * 8 + 5]); }
    auto med = [&](int k) { std::vector<unsigned long long> v; for (int w = 0; w < nwaves; ++w) v.push_back(h[w * 8 + k]); std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::vector<double> starts, ends;
    for (int w = 0; w < nwaves; ++w) { starts.push_back((h[w * 8] - r0) / 100.0); ends.push_back((h[w * 8 + 5] - r0) / 100.0); }
    std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
    fprintf(stderr, "[rp stamps] waves %d  kernel span %.1f us | per wave (median, cycles): prologue %llu  loop %llu (waits %llu, mfma-phase %llu) tiles %llu\n",
            nwaves, (r1 - r0) / 100.0, med(1), med(2), med(3), med(4), med(7));
    { std::vector<double> ghz; for (int w = 0; w < nwaves; ++w) ghz.push_back((double)(h[w * 8 + 1] + h[w * 8 + 2]) / ((h[w * 8 + 5] - h[w * 8]) * 10.0));
      std::sort(ghz.begin(), ghz.end());
      fprintf(stderr, "[rp stamps] in-kernel clock (s_memtime ticks per ns of s_memrealtime), GHz: p10 %.2f p50 %.2f p90 %.2f\n", ghz[nwaves / 10], ghz[nwaves / 2], ghz[nwaves * 9 / 10]); }
    fprintf(stderr, "[rp stamps] wave start us: p10 %.1f p50 %.1f p90 %.1f max %.1f | end us: p10 %.1f p50 %.1f p90 %.1f\n", starts[nwaves / 10], starts[nwaves / 2],
            starts[nwaves * 9 / 10], starts.back(), ends[nwaves / 10], ends[nwaves / 2], ends[nwaves * 9 / 10]);
}

}  // namespace vsom

using namespace vsom;

extern "C" {

int vsom_weight_image_kind(int n_out, int k_red) { return rp_kind(n_out, k_red); }
size_t vsom_weight_image_bytes(int n_out, int k_red) { return (size_t)rp_image_bytes(n_out, k_red); }

int vsom_weight_images_prepare(const float* params_base, void* images_base, const long long* table, int count, int max_fragments,
                               vsom_stream_t stream) {
    VSOM_REQUIRE(params_base && images_base && table && count > 0 && max_fragments > 0, VSOM_EINVAL, "weight_images_prepare: bad arguments");
    VSOM_REQUIRE(aligned16(images_base), VSOM_EALIGN, "weight_images_prepare: images must be 16-byte aligned");
    hipLaunchKernelGGL(weight_image_kernel, dim3(cdiv(max_fragments, 4), count), dim3(256), 0, stream, params_base,
                       static_cast<char*>(images_base), table);
    VSOM_LAUNCH_CHECK("weight_image_kernel");
}

int vsom_linear_planes(const vsom_linear_planes_args* a, vsom_stream_t stream) {
    VSOM_REQUIRE(a && a->X && a->Wimg && a->Y, VSOM_EINVAL, "linear_planes: null pointer");
    VSOM_REQUIRE(a->M > 0 && a->ldx >= a->K && a->ldy >= a->N, VSOM_EINVAL, "linear_planes: bad shape / leading dimension");
    const int kind = rp_kind(a->N, a->K);
    VSOM_REQUIRE(kind != 0, VSOM_EUNSUPPORTED, "linear_planes: shape N=%d K=%d is not taken by the row-panel engine", a->N, a->K);
    VSOM_REQUIRE(aligned16(a->X) && aligned16(a->Wimg) && a->ldx % 4 == 0, VSOM_EALIGN, "linear_planes: X / image / ldx must be 16-byte aligned");
    const long xb = ((long)(a->M - 1) * a->ldx + a->K) * 4;
    VSOM_REQUIRE(xb < 0xFFFF0000L, VSOM_EUNSUPPORTED, "linear_planes: X spans more than 4 GiB");
    const bool ln = a->ln_gamma != nullptr;
    VSOM_REQUIRE(!ln || (kind == 1 && a->ln_beta && a->ln_mean && a->ln_rstd && aligned16(a->ln_gamma) && aligned16(a->ln_beta)),
                 VSOM_EINVAL, "linear_planes: the LayerNorm prologue needs K in {48, 96, 192}, beta, mean and rstd (16-byte aligned)");
    VSOM_REQUIRE(!ln || !a->ln_out || (aligned16(a->ln_out) && a->ld_ln_out % 4 == 0 && a->ld_ln_out >= a->K), VSOM_EALIGN,
                 "linear_planes: ln_out must be 16-byte aligned");
    RpP p = {};
    GemmP& g = p.g;
    g.A = a->X; g.lda = a->ldx; g.C = a->Y; g.ldc = a->ldy; g.M = a->M; g.N = a->N; g.K = a->K;
    g.bias = a->bias; g.R = a->R; g.ldr = a->ldr; g.r_mod = a->r_mod > 0 ? a->r_mod : a->M; g.r_off = 0;
    g.C2 = a->Y2; g.ldc2 = a->ldy2; g.alpha = 1.f; g.accumulate = a->accumulate;
    g.a_bytes = (unsigned)xb;
    p.img = static_cast<const char*>(a->Wimg);
    p.img_bytes = (unsigned)rp_image_bytes(a->N, a->K);
    p.nt_total = cdiv(a->N, 32);
    p.gamma = a->ln_gamma; p.beta = a->ln_beta; p.eps = a->ln_eps; p.mean = a->ln_mean; p.rstd = a->ln_rstd;
    p.ln_out = a->ln_out; p.ld_ln = a->ld_ln_out;
    p.ablate = rp_env("VSOM_RP_ABLATE", 0);
    static unsigned long long* stamp_buf = nullptr;
    const bool stamps = rp_env("VSOM_RP_STAMPS", 0) != 0 && kind == 1;
    if (stamps && !stamp_buf) (void)hipMalloc(&stamp_buf, 1 << 22);
    p.stamps = stamps ? stamp_buf : nullptr;
    if (kind == 1) {
        p.nsplit = rp_panel_splits(a->M, 32 * rp_waves(), p.nt_total, a->K / 16, ln);
        p.tiles_per_split = cdiv(p.nt_total, p.nsplit);
        p.nsplit = cdiv(p.nt_total, p.tiles_per_split);
    } else {
        p.nsplit = 1; p.tiles_per_split = p.nt_total;
    }
    if (stamps) {                                         // lab: run, wait, print where the waves spent their time
        const int rc = launch_rp<EPI_BIAS>(p, kind, stream);
        rp_stamp_report(stamp_buf, cdiv(a->M, 32 * rp_waves()) * p.nsplit * rp_waves(), stream);
        return rc;
    }
    switch (a->epilogue) {
        case EPI_NONE: return launch_rp<EPI_NONE>(p, kind, stream);
        case EPI_BIAS: return launch_rp<EPI_BIAS>(p, kind, stream);
        case EPI_BIAS_GELU:
            VSOM_REQUIRE(a->Y2 && a->ldy2 >= a->N, VSOM_EINVAL, "linear_planes: the gelu epilogue needs Y2");
            return launch_rp<EPI_BIAS_GELU>(p, kind, stream);
        case EPI_BIAS_RES:
            VSOM_REQUIRE(a->R && a->ldr >= a->N, VSOM_EINVAL, "linear_planes: the residual epilogue needs R");
            return launch_rp<EPI_BIAS_RES>(p, kind, stream);
        case EPI_GELU_BWD:
            VSOM_REQUIRE(a->R && a->ldr >= a->N, VSOM_EINVAL, "linear_planes: the gelu-backward epilogue needs R");
            return launch_rp<EPI_GELU_BWD>(p, kind, stream);
        case EPI_BIAS_RELU:
            VSOM_REQUIRE(a->Y2 && a->ldy2 >= a->N, VSOM_EINVAL, "linear_planes: the relu epilogue needs Y2");
            return launch_rp<EPI_BIAS_RELU>(p, kind, stream);
    }
    set_error("linear_planes: unknown epilogue %d", a->epilogue);
    return VSOM_EINVAL;
}

}  // extern "C"
