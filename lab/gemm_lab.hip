// Lab: minimal NT f32-MFMA GEMM (no bounds checks) with compile-time switches, to find what limits
// the production kernel at M=33280, N=576, K=192.  C[M,N] = A[M,K] * B[N,K]^T.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bits: 1 = global loads in loop, 2 = LDS store + barriers, 4 = LDS frag reads, 8 = epilogue stores
template <int WM, int WN, int WAVES_M, int WAVES_N, int MODE, int OCC>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, OCC) void lab(const float* __restrict__ A, const float* __restrict__ B,
                                                              float* __restrict__ C, int M, int N, int K) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr int RPP = NT / 8;          // rows per pass
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * 36];
    float* As = lds; float* Bs = lds + BM * 36;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM * 32, wn0 = (wave % WAVES_N) * WN * 32;
    const int tiles_n = N / BN;
    const int bm0 = (blockIdx.x / tiles_n) * BM, bn0 = (blockIdx.x % tiles_n) * BN;
    f32x16 acc[WM][WN];
    for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    f32x4 sa[BM / RPP], sb[BN / RPP];
    const float* ap = A + (long)(bm0 + (t >> 3)) * K + ((t & 7) << 2);
    const float* bp = B + (long)(bn0 + (t >> 3)) * K + ((t & 7) << 2);
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) sa[p] = *(const f32x4*)(ap + (long)p * RPP * K + k0);
#pragma unroll
        for (int p = 0; p < BN / RPP; ++p) sb[p] = *(const f32x4*)(bp + (long)p * RPP * K + k0);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) *(f32x4*)(As + (p * RPP + (t >> 3)) * 36 + ((t & 7) << 2)) = sa[p];
#pragma unroll
        for (int p = 0; p < BN / RPP; ++p) *(f32x4*)(Bs + (p * RPP + (t >> 3)) * 36 + ((t & 7) << 2)) = sb[p];
    };
    gload(0); lstore(); __syncthreads();
    const int nk = K / 32;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if ((MODE & 1) && more) gload((kt + 1) * 32);
#pragma unroll
        for (int kb = 0; kb < 32; kb += 8) {
            f32x4 a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                if (MODE & 4) a[i] = *(const f32x4*)(As + (wm0 + i * 32 + r) * 36 + kb + 4 * h);
                else a[i] = f32x4{1.f + kb, 2.f, 3.f, 4.f + lane};
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (MODE & 4) b[j] = *(const f32x4*)(Bs + (wn0 + j * 32 + r) * 36 + kb + 4 * h);
                else b[j] = f32x4{1.f, 2.f + kb, 3.f + lane, 4.f};
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        if (MODE & 2) {
            __syncthreads();
            if (more) { lstore(); __syncthreads(); }
        }
    }
    if (MODE & 8) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = bm0 + wm0 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h, n = bn0 + wn0 + j * 32 + r;
                    C[(long)m * N + n] = acc[i][j][v];
                }
    } else {
        float s = 0; for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int v = 0; v < 16; ++v) s += acc[i][j][v];
        if (s == 123.456f) C[t] = s;
    }
}
template <int WM, int WN, int WAVES_M, int WAVES_N, int MODE, int OCC>
void run(const char* tag, const float* A, const float* B, float* C, int M, int N, int K) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    dim3 grid((M / BM) * (N / BN));
    auto go = [&]() { hipLaunchKernelGGL((lab<WM, WN, WAVES_M, WAVES_N, MODE, OCC>), grid, dim3(WAVES_M * WAVES_N * 64), 0, 0, A, B, C, M, N, K); };
    go(); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); for (int i = 0; i < 5; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 5 < best) best = ms / 5; }
    printf("%-28s tile %dx%d waves %dx%d mode %2d occ %d: %7.1f us  %6.1f TF\n", tag, BM, BN, WAVES_M, WAVES_N, MODE, OCC, best * 1e3, 2.0 * M * N * K / best / 1e9);
}

// ---- variant: direct-to-LDS loads (global_load_lds dwordx4), double-buffered, XOR-swizzled unpadded image --------
// LDS image per operand tile: [rows][32 floats]; 16-B chunk position p of row r holds global chunk p ^ ((r>>1)&7).
template <int WM, int WN, int WAVES_M, int WAVES_N, int OCC>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, OCC) void lab_glds(const float* __restrict__ A, const float* __restrict__ B,
                                                                   float* __restrict__ C, int M, int N, int K) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr int TILE = (BM + BN) * 32;                  // floats per stage
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM * 32, wn0 = (wave % WAVES_N) * WN * 32;
    const int tiles_n = N / BN;
    const int bm0 = (blockIdx.x / tiles_n) * BM, bn0 = (blockIdx.x % tiles_n) * BN;
    f32x16 acc[WM][WN];
    for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    // each wave-instruction fills 8 rows (1 KiB): lane -> (row_in_8 = lane>>3, position p = lane&7)
    const int lrow = lane >> 3, lp = lane & 7;
    auto stage = [&](int buf, int k0) {
        float* As = lds + buf * TILE; float* Bs = As + BM * 32;
#pragma unroll
        for (int q = 0; q < BM / 8 / NW; ++q) {
            const int row = (q * NW + wave) * 8 + lrow;               // row within the A tile
            const int c = lp ^ ((row >> 1) & 7);
            const float* src = A + (long)(bm0 + row) * K + k0 + 4 * c;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(As + (q * NW + wave) * 256), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < BN / 8 / NW; ++q) {
            const int row = (q * NW + wave) * 8 + lrow;
            const int c = lp ^ ((row >> 1) & 7);
            const float* src = B + (long)(bn0 + row) * K + k0 + 4 * c;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Bs + (q * NW + wave) * 256), 16, 0, 0);
        }
    };
    stage(0, 0);
    __syncthreads();
    const int nk = K / 32;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * 32);
        const float* As = lds + (kt & 1) * TILE; const float* Bs = As + BM * 32;
#pragma unroll
        for (int kb = 0; kb < 32; kb += 8) {
            f32x4 a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) { const int row = wm0 + i * 32 + r; a[i] = *(const f32x4*)(As + row * 32 + 4 * ((kb / 4 + h) ^ ((row >> 1) & 7))); }
#pragma unroll
            for (int j = 0; j < WN; ++j) { const int row = wn0 + j * 32 + r; b[j] = *(const f32x4*)(Bs + row * 32 + 4 * ((kb / 4 + h) ^ ((row >> 1) & 7))); }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = bm0 + wm0 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h, n = bn0 + wn0 + j * 32 + r;
                C[(long)m * N + n] = acc[i][j][v];
            }
}
template <int WM, int WN, int WAVES_M, int WAVES_N, int OCC>
void run_glds(const char* tag, const float* A, const float* B, float* C, float* Cref, int M, int N, int K) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    dim3 grid((M / BM) * (N / BN));
    auto go = [&]() { hipLaunchKernelGGL((lab_glds<WM, WN, WAVES_M, WAVES_N, OCC>), grid, dim3(WAVES_M * WAVES_N * 64), 0, 0, A, B, C, M, N, K); };
    go(); hipDeviceSynchronize();
    // check against the register-staged kernel's output
    std::vector<float> x(1 << 16), y(1 << 16);
    hipMemcpy(x.data(), C + 12345, x.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(y.data(), Cref + 12345, y.size() * 4, hipMemcpyDeviceToHost);
    double md = 0; for (size_t i = 0; i < x.size(); ++i) md = fmax(md, fabs((double)x[i] - y[i]));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); for (int i = 0; i < 5; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 5 < best) best = ms / 5; }
    printf("%-28s tile %dx%d waves %dx%d occ %d: %7.1f us  %6.1f TF   (max |diff| vs reg-staged %.3g)\n", tag, BM, BN, WAVES_M, WAVES_N, OCC, best * 1e3, 2.0 * M * N * K / best / 1e9, md);
}

// ---- variant: persistent workgroups, natural loop nest, cross-tile prefetch -------------------------------------
template <int WM, int WN, int WAVES_M, int WAVES_N, int OCC>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, OCC) void lab_persist(const float* __restrict__ A, const float* __restrict__ B,
                                                                      float* __restrict__ C, int M, int N, int K) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr int RPP = NT / 8;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * 36];
    float* As = lds; float* Bs = lds + BM * 36;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM * 32, wn0 = (wave % WAVES_N) * WN * 32;
    const int tiles_n = N / BN, ntiles = (M / BM) * tiles_n;
    const int nk = K / 32;
    f32x4 sa[BM / RPP], sb[BN / RPP];
    auto gload = [&](int tile, int k0) {
        const int bm0 = (tile / tiles_n) * BM, bn0 = (tile % tiles_n) * BN;
        const float* ap = A + (long)(bm0 + (t >> 3)) * K + ((t & 7) << 2) + k0;
        const float* bp = B + (long)(bn0 + (t >> 3)) * K + ((t & 7) << 2) + k0;
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) sa[p] = *(const f32x4*)(ap + (long)p * RPP * K);
#pragma unroll
        for (int p = 0; p < BN / RPP; ++p) sb[p] = *(const f32x4*)(bp + (long)p * RPP * K);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) *(f32x4*)(As + (p * RPP + (t >> 3)) * 36 + ((t & 7) << 2)) = sa[p];
#pragma unroll
        for (int p = 0; p < BN / RPP; ++p) *(f32x4*)(Bs + (p * RPP + (t >> 3)) * 36 + ((t & 7) << 2)) = sb[p];
    };
    int tile = blockIdx.x;
    if (tile < ntiles) { gload(tile, 0); lstore(); }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const int bm0 = (tile / tiles_n) * BM, bn0 = (tile % tiles_n) * BN;
        f32x16 acc[WM][WN];
        for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
        const int ntile = tile + gridDim.x;
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1 < nk) || (ntile < ntiles);
            if (kt + 1 < nk) gload(tile, (kt + 1) * 32); else if (ntile < ntiles) gload(ntile, 0);
#pragma unroll
            for (int kb = 0; kb < 32; kb += 8) {
                f32x4 a[WM], b[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) a[i] = *(const f32x4*)(As + (wm0 + i * 32 + r) * 36 + kb + 4 * h);
#pragma unroll
                for (int j = 0; j < WN; ++j) b[j] = *(const f32x4*)(Bs + (wn0 + j * 32 + r) * 36 + kb + 4 * h);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) { lstore(); __syncthreads(); }
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = bm0 + wm0 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h, n = bn0 + wn0 + j * 32 + r;
                    C[(long)m * N + n] = acc[i][j][v];
                }
    }
}
template <int WM, int WN, int WAVES_M, int WAVES_N, int OCC>
void run_persist(const char* tag, const float* A, const float* B, float* C, float* Cref, int M, int N, int K, int grid) {
    auto go = [&]() { hipLaunchKernelGGL((lab_persist<WM, WN, WAVES_M, WAVES_N, OCC>), dim3(grid), dim3(WAVES_M * WAVES_N * 64), 0, 0, A, B, C, M, N, K); };
    go(); hipDeviceSynchronize();
    std::vector<float> x(1 << 16), y(1 << 16);
    hipMemcpy(x.data(), C + 12345, x.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(y.data(), Cref + 12345, y.size() * 4, hipMemcpyDeviceToHost);
    double md = 0; for (size_t i = 0; i < x.size(); ++i) md = fmax(md, fabs((double)x[i] - y[i]));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); for (int i = 0; i < 5; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 5 < best) best = ms / 5; }
    printf("%-28s grid %4d occ %d: %7.1f us  %6.1f TF   (max |diff| %.3g)\n", tag, grid, OCC, best * 1e3, 2.0 * M * N * K / best / 1e9, md);
}

// ---- variant: fp32-accurate GEMM on bf16 MFMA via 3-way operand split (6 products) -----------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float lo_f(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// v (4 floats) -> three planes of 4 bf16 (8 bytes each).
#ifndef SPLIT_MODE
#define SPLIT_MODE 1
#endif
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ void split3(f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
#if SPLIT_MODE == 0   // RNE pieces, scalar casts
    p1.x = pack2(v[0], v[1]); p1.y = pack2(v[2], v[3]);
    const float r0 = v[0] - lo_f(p1.x), r1 = v[1] - hi_f(p1.x), r2 = v[2] - lo_f(p1.y), r3 = v[3] - hi_f(p1.y);
    p2.x = pack2(r0, r1); p2.y = pack2(r2, r3);
    const float s0 = r0 - lo_f(p2.x), s1 = r1 - hi_f(p2.x), s2 = r2 - lo_f(p2.y), s3 = r3 - hi_f(p2.y);
    p3.x = pack2(s0, s1); p3.y = pack2(s2, s3);
#elif SPLIT_MODE == 1  // truncation pieces: a = a1 + a2 + a3 EXACTLY (8+8+8 significand bits)
    const unsigned MSK = 0xffff0000u;
    float r[4], s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = v[e] - bitsf(fbits(v[e]) & MSK); s[e] = r[e] - bitsf(fbits(r[e]) & MSK); }
    p1.x = __builtin_amdgcn_perm(fbits(v[1]), fbits(v[0]), 0x07060302u); p1.y = __builtin_amdgcn_perm(fbits(v[3]), fbits(v[2]), 0x07060302u);
    p2.x = __builtin_amdgcn_perm(fbits(r[1]), fbits(r[0]), 0x07060302u); p2.y = __builtin_amdgcn_perm(fbits(r[3]), fbits(r[2]), 0x07060302u);
    p3.x = __builtin_amdgcn_perm(fbits(s[1]), fbits(s[0]), 0x07060302u); p3.y = __builtin_amdgcn_perm(fbits(s[3]), fbits(s[2]), 0x07060302u);
#else                  // RNE pieces, packed converts
    auto cv = [](float a, float b) { f32x2 t = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2)); };
    p1.x = cv(v[0], v[1]); p1.y = cv(v[2], v[3]);
    const float r0 = v[0] - lo_f(p1.x), r1 = v[1] - hi_f(p1.x), r2 = v[2] - lo_f(p1.y), r3 = v[3] - hi_f(p1.y);
    p2.x = cv(r0, r1); p2.y = cv(r2, r3);
    const float s0 = r0 - lo_f(p2.x), s1 = r1 - hi_f(p2.x), s2 = r2 - lo_f(p2.y), s3 = r3 - hi_f(p2.y);
    p3.x = cv(s0, s1); p3.y = cv(s2, s3);
#endif
}
template <int WM, int WN, int WAVES_M, int WAVES_N, int OCC>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, OCC) void lab_x6(const float* __restrict__ A, const float* __restrict__ B,
                                                                 float* __restrict__ C, int M, int N, int K) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr int RPP = NT / 8;
    constexpr int RS = 80;                                   // bytes per row of a plane (32 bf16 + 16 B pad)
    constexpr int PA = BM * RS, PB = BN * RS;                // bytes per plane
    __shared__ __attribute__((aligned(16))) char lds[3 * (PA + PB)];
    char* As = lds; char* Bs = lds + 3 * PA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM * 32, wn0 = (wave % WAVES_N) * WN * 32;
    const int tiles_n = N / BN;
    const int bm0 = (blockIdx.x / tiles_n) * BM, bn0 = (blockIdx.x % tiles_n) * BN;
    f32x16 acc[WM][WN];
    for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    f32x4 sa[BM / RPP], sb[BN / RPP];
    const float* ap = A + (long)(bm0 + (t >> 3)) * K + ((t & 7) << 2);
    const float* bp = B + (long)(bn0 + (t >> 3)) * K + ((t & 7) << 2);
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) sa[p] = *(const f32x4*)(ap + (long)p * RPP * K + k0);
#pragma unroll
        for (int p = 0; p < BN / RPP; ++p) sb[p] = *(const f32x4*)(bp + (long)p * RPP * K + k0);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) {
            uint2 p1, p2, p3; split3(sa[p], p1, p2, p3);
            const int off = (p * RPP + (t >> 3)) * RS + ((t & 7) << 3);
            *(uint2*)(As + off) = p1; *(uint2*)(As + PA + off) = p2; *(uint2*)(As + 2 * PA + off) = p3;
        }
#pragma unroll
        for (int p = 0; p < BN / RPP; ++p) {
            uint2 p1, p2, p3; split3(sb[p], p1, p2, p3);
            const int off = (p * RPP + (t >> 3)) * RS + ((t & 7) << 3);
            *(uint2*)(Bs + off) = p1; *(uint2*)(Bs + PB + off) = p2; *(uint2*)(Bs + 2 * PB + off) = p3;
        }
    };
    gload(0); lstore(); __syncthreads();
    const int nk = K / 32;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload((kt + 1) * 32);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[WM][3], b[WN][3];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][pl] = *(const bf16x8*)(As + pl * PA + (wm0 + i * 32 + r) * RS + ks * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[j][pl] = *(const bf16x8*)(Bs + pl * PB + (wn0 + j * 32 + r) * RS + ks * 32 + h * 16);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) { lstore(); __syncthreads(); }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = bm0 + wm0 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h, n = bn0 + wn0 + j * 32 + r;
                C[(long)m * N + n] = acc[i][j][v];
            }
}
template <int WM, int WN, int WAVES_M, int WAVES_N, int OCC>
void run_x6(const char* tag, const float* A, const float* B, float* C, float* Cref, const std::vector<float>& hA, const std::vector<float>& hB, int M, int N, int K) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    dim3 grid((M / BM) * (N / BN));
    auto go = [&]() { hipLaunchKernelGGL((lab_x6<WM, WN, WAVES_M, WAVES_N, OCC>), grid, dim3(WAVES_M * WAVES_N * 64), 0, 0, A, B, C, M, N, K); };
    go(); hipDeviceSynchronize();
    // accuracy vs fp64 on 2 rows, for both this kernel and the exact-f32 kernel's output
    std::vector<float> x(N), y(N); double e6 = 0, e32 = 0, nrm = 0;
    for (int row : {0, 777, M - 1}) {
        hipMemcpy(x.data(), C + (long)row * N, N * 4, hipMemcpyDeviceToHost); hipMemcpy(y.data(), Cref + (long)row * N, N * 4, hipMemcpyDeviceToHost);
        for (int n = 0; n < N; ++n) { double ref = 0; for (int k = 0; k < K; ++k) ref += (double)hA[(long)row * K + k] * hB[(long)n * K + k];
            e6 += (x[n] - ref) * (x[n] - ref); e32 += (y[n] - ref) * (y[n] - ref); nrm += ref * ref; }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); for (int i = 0; i < 5; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 5 < best) best = ms / 5; }
    printf("%-22s tile %dx%d occ %d: %7.1f us  %6.1f TF(f32-equiv)  rel err vs fp64: x6 %.2e, exact-f32 %.2e\n", tag, BM, BN, OCC, best * 1e3, 2.0 * M * N * K / best / 1e9, sqrt(e6 / nrm), sqrt(e32 / nrm));
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 33280, N = argc > 2 ? atoi(argv[2]) : 576, K = argc > 3 ? atoi(argv[3]) : 192;
    const bool only_f32 = argc > 4;
    float *A, *B, *C; hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> h((size_t)(M > N ? M : N) * K); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(((i * 2654435761u) >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    run<1, 2, 4, 1, 0, 1>("mfma only", A, B, C, M, N, K);
    run<1, 2, 4, 1, 4, 1>("+lds reads", A, B, C, M, N, K);
    run<1, 2, 4, 1, 6, 1>("+lds reads+store/barrier", A, B, C, M, N, K);
    run<1, 2, 4, 1, 7, 1>("+global loads", A, B, C, M, N, K);
    run<1, 2, 4, 1, 15, 1>("full", A, B, C, M, N, K);
    run<1, 2, 4, 1, 8, 1>("mfma+epilogue", A, B, C, M, N, K);
    float* C2; hipMalloc(&C2, (size_t)M * N * 4);
    run<1, 2, 4, 1, 15, 1>("full (reference out)", A, B, C2, M, N, K);
    if (only_f32) return 0;
    { std::vector<float> hB(h.begin(), h.begin() + (size_t)N * K);
      run_x6<1, 2, 4, 1, 1>("bf16x6 split", A, B, C, C2, h, hB, M, N, K);
      run_x6<1, 2, 4, 1, 2>("bf16x6 split", A, B, C, C2, h, hB, M, N, K);
      run_x6<2, 2, 2, 2, 1>("bf16x6 split", A, B, C, C2, h, hB, M, N, K);
      run_x6<2, 2, 4, 1, 1>("bf16x6 split 256x64", A, B, C, C2, h, hB, M, N, K); }
    run_persist<1, 2, 4, 1, 1>("persist natural", A, B, C, C2, M, N, K, 768);
    run_persist<1, 2, 4, 1, 1>("persist natural", A, B, C, C2, M, N, K, 780);
    run_persist<1, 2, 4, 1, 1>("persist natural", A, B, C, C2, M, N, K, 1024);
    run_persist<1, 2, 4, 1, 1>("persist natural", A, B, C, C2, M, N, K, 1170);
    run_glds<1, 2, 4, 1, 1>("glds dbuf", A, B, C, C2, M, N, K);
    run_glds<1, 2, 4, 1, 3>("glds dbuf", A, B, C, C2, M, N, K);
    run_glds<1, 2, 4, 1, 4>("glds dbuf", A, B, C, C2, M, N, K);
    run_glds<2, 2, 2, 2, 2>("glds dbuf 128x128", A, B, C, C2, M, N, K);
    run<1, 2, 4, 1, 15, 2>("full occ2", A, B, C, M, N, K);
    run<1, 2, 4, 1, 15, 3>("full occ3", A, B, C, M, N, K);
    run<1, 2, 4, 1, 15, 4>("full occ4", A, B, C, M, N, K);
    run<2, 2, 2, 1, 15, 2>("full 2 waves of 64x64", A, B, C, M, N, K);
    run<2, 2, 2, 1, 15, 3>("full 2 waves of 64x64", A, B, C, M, N, K);
    run<2, 2, 2, 1, 0, 1>("mfma only 2 waves 64x64", A, B, C, M, N, K);
    return 0;
}
