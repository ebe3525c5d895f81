// Lab: split-bf16 weight-gradient GEMM  dW[n,k] = sum_t dY[t,n] X[t,k]  (both operands strided along the reduction).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/tn_lab tools/tn_lab.hip
// usage: tn_lab T N_out K_in
//
// The fp32 tiles are loaded as they lie in memory (rows = reduction index t, 16-byte loads along the columns), split
// into three bf16 planes and written to LDS in the SAME orientation ([t][col], 8-byte stores); the MFMA fragments
// (8 consecutive t for one column) come out of LDS through ds_read_b64_tr_b16 -- no register transposes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
__device__ __forceinline__ unsigned fb(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bf(unsigned x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ void split3(f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
    const unsigned HI = 0xffff0000u, SEL = 0x07060302u;
    float r[4], s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = v[e] - bf(fb(v[e]) & HI); s[e] = r[e] - bf(fb(r[e]) & HI); }
    p1.x = __builtin_amdgcn_perm(fb(v[1]), fb(v[0]), SEL); p1.y = __builtin_amdgcn_perm(fb(v[3]), fb(v[2]), SEL);
    p2.x = __builtin_amdgcn_perm(fb(r[1]), fb(r[0]), SEL); p2.y = __builtin_amdgcn_perm(fb(r[3]), fb(r[2]), SEL);
    p3.x = __builtin_amdgcn_perm(fb(s[1]), fb(s[0]), SEL); p3.y = __builtin_amdgcn_perm(fb(s[3]), fb(s[2]), SEL);
}
__device__ __forceinline__ s16x4 trread(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
// row stride (bytes) of a [32 t][C cols] bf16 plane: == 64 (mod 128) -> the four t-rows of a transposed read tile the
// 256-byte bank row
constexpr int tn_stride(int C) { return ((C * 2 + 63) / 128) * 128 + 64; }

template <int WM, int WN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void tn(const float* __restrict__ dY, long ldy, const float* __restrict__ X, long ldx,
                                                              float* __restrict__ slab, int T, int NO, int KI, int ktiles_per_split) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, NT = WAVES_M * WAVES_N * 64;
    constexpr int SA = tn_stride(BM), SB = tn_stride(BN), PA = 32 * SA, PB = 32 * SB;
    __shared__ __attribute__((aligned(16))) char lds[3 * (PA + PB)];
    char* As = lds; char* Bs = lds + 3 * PA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm0 = (wave / WAVES_N) * WM * 32, wn0 = (wave % WAVES_N) * WN * 32;
    const int tiles_n = KI / BN, tiles_m = NO / BM, ntiles = tiles_m * tiles_n;
    const int z = blockIdx.x / ntiles, rem = blockIdx.x - z * ntiles;
    const int bm0 = (rem / tiles_n) * BM, bn0 = (rem % tiles_n) * BN;
    const int ktiles = (T + 31) >> 5;
    const int kt_begin = z * ktiles_per_split;
    int kt_end = kt_begin + ktiles_per_split; if (kt_end > ktiles) kt_end = ktiles;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY), 0, (int)(((long)(T - 1) * ldy + NO) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)(((long)(T - 1) * ldx + KI) * 4), 0x00020000);
    // staging: float4 index idx = i*NT + t of the [32][C/4] tile
    constexpr int FA = (32 * BM / 4 + NT - 1) / NT, FB = (32 * BN / 4 + NT - 1) / NT;
    f32x4 sa[FA], sb[FB];
    unsigned ga[FA], gb[FB];      // global byte offsets (row part excluded: + kt*32*ld*4 per k-tile via soffset)
    int la[FA], lb[FB];           // LDS byte offsets inside plane 0
#pragma unroll
    for (int i = 0; i < FA; ++i) {
        const int idx = i * NT + t, row = idx / (BM / 4), c4 = idx % (BM / 4);
        const bool ok = idx < 32 * BM / 4;
        ga[i] = ok ? (unsigned)(((long)row * ldy + bm0 + c4 * 4) * 4) : 0xFFFFFFF0u;
        la[i] = ok ? row * SA + c4 * 8 : -1;
    }
#pragma unroll
    for (int i = 0; i < FB; ++i) {
        const int idx = i * NT + t, row = idx / (BN / 4), c4 = idx % (BN / 4);
        const bool ok = idx < 32 * BN / 4;
        gb[i] = ok ? (unsigned)(((long)row * ldx + bn0 + c4 * 4) * 4) : 0xFFFFFFF0u;
        lb[i] = ok ? row * SB + c4 * 8 : -1;
    }
    auto gload = [&](int kt) {
        // rows beyond T: offset beyond the buffer -> zeros (the descriptor ends at the last valid element)
        const unsigned sa_off = (unsigned)((long)kt * 32 * ldy * 4), sb_off = (unsigned)((long)kt * 32 * ldx * 4);
#pragma unroll
        for (int i = 0; i < FA; ++i) sa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ga[i], sa_off, 0));
#pragma unroll
        for (int i = 0; i < FB; ++i) sb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, gb[i], sb_off, 0));
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            if (la[i] < 0) continue;
            uint2 p1, p2, p3; split3(sa[i], p1, p2, p3);
            *(uint2*)(As + la[i]) = p1; *(uint2*)(As + PA + la[i]) = p2; *(uint2*)(As + 2 * PA + la[i]) = p3;
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            if (lb[i] < 0) continue;
            uint2 p1, p2, p3; split3(sb[i], p1, p2, p3);
            *(uint2*)(Bs + lb[i]) = p1; *(uint2*)(Bs + PB + lb[i]) = p2; *(uint2*)(Bs + 2 * PB + lb[i]) = p3;
        }
    };
    // transposed fragment reads: lane l -> group G = l >> 4, q = (l >> 2) & 3, pp = l & 3
    const int G = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int trow = 8 * (G >> 1) + q;                 // + 16*ks (+4 for the second half)
    const int tcol = 16 * (G & 1) + 4 * pp;            // + 32-column block
    int fa[WM], fbo[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) fa[i] = trow * SA + (wm0 + i * 32 + tcol) * 2;
#pragma unroll
    for (int j = 0; j < WN; ++j) fbo[j] = trow * SB + (wn0 + j * 32 + tcol) * 2;
    auto frag = [&](const char* base, int off, int stride, int ks) -> bf16x8 {
        const s16x4 lo = trread(base + off + (ks * 16) * stride);
        const s16x4 hi = trread(base + off + (ks * 16 + 4) * stride);
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto mfma_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[WM][3], b[WN][3];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][pl] = frag(As + pl * PA, fa[i], SA, ks);
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[j][pl] = frag(Bs + pl * PB, fbo[j], SB, ks);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
    };
    if (kt_begin < kt_end) { gload(kt_begin); lstore(); }
    __syncthreads();
    for (int kt = kt_begin; kt + 1 < kt_end; ++kt) {
        gload(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_tile();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        lstore();
        __syncthreads();
    }
    if (kt_begin < kt_end) mfma_tile();
    // slab[z][n*KI + k]; accumulator register v of tile (i, j): row (v&3) + 8*(v>>2) + 4*h, column r
    const int r = lane & 31, h = lane >> 5;
    float* sl = slab + (long)z * NO * KI;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float* d = sl + (long)(bm0 + wm0 + i * 32 + 4 * h) * KI + bn0 + wn0 + j * 32 + r;
#pragma unroll
            for (int v = 0; v < 16; ++v) { *d = acc[i][j][v]; d += (((v & 3) == 3) ? 5 : 1) * (long)KI; }
        }
}

__global__ void reduce_slabs(const float* slab, int nslab, long n, float* out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int z = 0; z < nslab; ++z) s += slab[(long)z * n + i];
    out[i] = s;
}

static std::vector<double> g_ref;
template <int WM, int WN, int WAVES_M, int WAVES_N>
void run(const float* dY, const float* X, float* slab, float* out, int T, int NO, int KI, int splits) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, NT = WAVES_M * WAVES_N * 64;
    if (NO % BM || KI % BN) { printf("  tile %3dx%-3d skipped\n", BM, BN); return; }
    const int tiles = (NO / BM) * (KI / BN), ktiles = (T + 31) / 32;
    if (splits <= 0) splits = (512 + tiles - 1) / tiles;
    int per = (ktiles + splits - 1) / splits; splits = (ktiles + per - 1) / per;
    dim3 grid(tiles * splits);
    auto go = [&]() { hipLaunchKernelGGL((tn<WM, WN, WAVES_M, WAVES_N>), grid, dim3(NT), 0, 0, dY, (long)NO, X, (long)KI, slab, T, NO, KI, per); };
    go();
    hipLaunchKernelGGL(reduce_slabs, dim3((NO * KI + 255) / 256), dim3(256), 0, 0, slab, splits, (long)NO * KI, out);
    hipDeviceSynchronize();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("  launch error %s\n", hipGetErrorString(e)); return; }
    std::vector<float> ho((size_t)NO * KI); hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    for (size_t i = 0; i < g_ref.size(); ++i) { md = fmax(md, fabs(ho[i] - g_ref[i])); mx = fmax(mx, fabs(g_ref[i])); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rr = 0; rr < 6; ++rr) { hipEventRecord(e0); for (int i = 0; i < 10; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 10 < best) best = ms / 10; }
    printf("  tile %3dx%-3d thr %3d splits %3d grid %5d: %7.1f us  %6.1f TF(f32-eq)  rel.err %.2e (first %zu outputs)\n", BM, BN, NT, splits, grid.x, best * 1e3,
           2.0 * T * NO * KI / best / 1e9, md / mx, g_ref.size());
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 33280, NO = argc > 2 ? atoi(argv[2]) : 576, KI = argc > 3 ? atoi(argv[3]) : 192;
    float *dY, *X, *slab, *out;
    hipMalloc(&dY, (size_t)T * NO * 4); hipMalloc(&X, (size_t)T * KI * 4); hipMalloc(&slab, (size_t)600 * NO * KI * 4); hipMalloc(&out, (size_t)NO * KI * 4);
    std::vector<float> hy((size_t)T * NO), hx((size_t)T * KI);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((float)((s >> 8) & 0xffff) / 65536.f - 0.5f) * (1.f + (float)(s >> 28)); };
    for (auto& v : hy) v = rnd();
    for (auto& v : hx) v = rnd();
    hipMemcpy(dY, hy.data(), hy.size() * 4, hipMemcpyHostToDevice); hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    // fp64 reference of the first 2 output rows + a few scattered
    const int NREF = 2 * KI;
    g_ref.assign(NREF, 0.0);
    for (int i = 0; i < NREF; ++i) { const int n = i / KI, k = i % KI; double a = 0; for (int tt = 0; tt < T; ++tt) a += (double)hy[(size_t)tt * NO + n] * hx[(size_t)tt * KI + k]; g_ref[i] = a; }
    printf("T=%d N_out=%d K_in=%d  (%.2f GFLOP f32-eq)\n", T, NO, KI, 2.0 * T * NO * KI / 1e9);
    run<1, 1, 2, 2>(dY, X, slab, out, T, NO, KI, 0);
    run<1, 3, 2, 2>(dY, X, slab, out, T, NO, KI, 0);
    run<1, 3, 2, 2>(dY, X, slab, out, T, NO, KI, 86);
    run<1, 3, 3, 2>(dY, X, slab, out, T, NO, KI, 0);
    run<2, 3, 3, 2>(dY, X, slab, out, T, NO, KI, 0);
    run<2, 3, 3, 2>(dY, X, slab, out, T, NO, KI, 86);
    run<3, 3, 2, 2>(dY, X, slab, out, T, NO, KI, 0);
    run<3, 1, 2, 2>(dY, X, slab, out, T, NO, KI, 0);
    run<3, 2, 2, 1>(dY, X, slab, out, T, NO, KI, 0);
    return 0;
}
