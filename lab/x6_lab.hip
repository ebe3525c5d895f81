// Ablation bench for the split-bf16 GEMM main loop (NT).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/x6_lab tools/x6_lab.hip
// usage: x6_lab M N K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
enum { SPLIT = 1, LDSW = 2, LDSR = 4, MFMA6 = 8, EPI = 16, GLOAD = 32, ALL = 63, BPRE = 64 };   // BPRE: B comes pre-split (3 bf16 planes in HBM)
__device__ __forceinline__ unsigned fb(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bf(unsigned x) { return __builtin_bit_cast(float, x); }
template <int MODE>
__device__ __forceinline__ void split3(f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
    const unsigned HI = 0xffff0000u, SEL = 0x07060302u;
    float r[4], s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (MODE & SPLIT) { r[e] = v[e] - bf(fb(v[e]) & HI); s[e] = r[e] - bf(fb(r[e]) & HI); } else { r[e] = v[e]; s[e] = v[e]; }
    }
    p1.x = __builtin_amdgcn_perm(fb(v[1]), fb(v[0]), SEL); p1.y = __builtin_amdgcn_perm(fb(v[3]), fb(v[2]), SEL);
    p2.x = __builtin_amdgcn_perm(fb(r[1]), fb(r[0]), SEL); p2.y = __builtin_amdgcn_perm(fb(r[3]), fb(r[2]), SEL);
    p3.x = __builtin_amdgcn_perm(fb(s[1]), fb(s[0]), SEL); p3.y = __builtin_amdgcn_perm(fb(s[3]), fb(s[2]), SEL);
}
template <int WM, int WN, int WAVES_M, int WAVES_N, int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                         const unsigned short* __restrict__ Bp) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, RS = 80, PA = BM * RS, PB = BN * RS;
    __shared__ __attribute__((aligned(16))) char lds[3 * (PA + PB)];
    char* As = lds; char* Bs = lds + 3 * PA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM * 32, wn0 = (wave % WAVES_N) * WN * 32;
    const int tiles_n = N / BN;
    const int bm0 = (blockIdx.x / tiles_n) * BM, bn0 = (blockIdx.x % tiles_n) * BN;
    if (bm0 >= M) return;
    f32x16 acc[WM][WN];
    for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    f32x4 sa[BM / 32], sb[BN / 32];
    f32x4 pb[3];                                         // BPRE: 8 bf16 of each plane (tile BN = 64: one 16-byte chunk per thread)
    const unsigned short* bpp = Bp + (long)(bn0 + (t >> 2)) * K + ((t & 3) << 3);
    const float* ap = A + (long)(bm0 + (t >> 3)) * K + ((t & 7) << 2);
    const float* bp = B + (long)(bn0 + (t >> 3)) * K + ((t & 7) << 2);
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < BM / 32; ++p) sa[p] = *(const f32x4*)(ap + (long)p * 32 * K + k0);
        if (MODE & BPRE) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) pb[pl] = *(const f32x4*)(bpp + (long)pl * N * K + k0);
        } else {
#pragma unroll
            for (int p = 0; p < BN / 32; ++p) sb[p] = *(const f32x4*)(bp + (long)p * 32 * K + k0);
        }
    };
    auto lstore = [&](bool force) {
        if (!(MODE & LDSW) && !force) {     // keep the values live without writing
            float s = 0; for (int p = 0; p < BM / 32; ++p) s += sa[p][0]; if (!(MODE & BPRE)) for (int p = 0; p < BN / 32; ++p) s += sb[p][0];
            if (s == 12345.678f) As[0] = 1;
            return;
        }
#pragma unroll
        for (int p = 0; p < BM / 32; ++p) {
            uint2 p1, p2, p3; split3<MODE>(sa[p], p1, p2, p3);
            const int off = (p * 32 + (t >> 3)) * RS + ((t & 7) << 3);
            *(uint2*)(As + off) = p1; *(uint2*)(As + PA + off) = p2; *(uint2*)(As + 2 * PA + off) = p3;
        }
        if (MODE & BPRE) {
            const int off = (t >> 2) * RS + ((t & 3) << 4);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *(f32x4*)(Bs + pl * PB + off) = pb[pl];
        } else {
#pragma unroll
            for (int p = 0; p < BN / 32; ++p) {
                uint2 p1, p2, p3; split3<MODE>(sb[p], p1, p2, p3);
                const int off = (p * 32 + (t >> 3)) * RS + ((t & 7) << 3);
                *(uint2*)(Bs + off) = p1; *(uint2*)(Bs + PB + off) = p2; *(uint2*)(Bs + 2 * PB + off) = p3;
            }
        }
    };
    gload(0); lstore(true); __syncthreads();
    const int nk = K / 32;
    bf16x8 a[WM][3], b[WN][3];
    auto rd = [&](int ks) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[i][pl] = *(const bf16x8*)(As + pl * PA + (wm0 + i * 32 + r) * RS + ks * 32 + h * 16);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[j][pl] = *(const bf16x8*)(Bs + pl * PB + (wn0 + j * 32 + r) * RS + ks * 32 + h * 16);
    };
    if (!(MODE & LDSR)) rd(0);
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more && (MODE & GLOAD)) gload((kt + 1) * 32);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (MODE & LDSR) rd(ks);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    f32x16 c = acc[i][j];
                    if (MODE & MFMA6) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        __syncthreads();
        if (more) { lstore(false); __syncthreads(); }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (!(MODE & EPI)) { float s = 0; for (int v = 0; v < 16; ++v) s += acc[i][j][v]; if (s != 12345.678f) continue; }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = bm0 + wm0 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h, n = bn0 + wn0 + j * 32 + r;
                if (m < M) C[(long)m * N + n] = acc[i][j][v];
            }
        }
}
template <int WM, int WN, int WAVES_M, int WAVES_N, int MODE>
void run(const char* tag, const float* A, const float* B, float* C, int M, int N, int K, const unsigned short* Bp = nullptr) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    dim3 grid(((M + BM - 1) / BM) * (N / BN));
    auto go = [&]() { hipLaunchKernelGGL((k<WM, WN, WAVES_M, WAVES_N, MODE>), grid, dim3(256), 0, 0, A, B, C, M, N, K, Bp); };
    go(); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); for (int i = 0; i < 5; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms / 5 < best) best = ms / 5; }
    printf("  %-34s tile %3dx%-3d: %7.1f us  %6.1f TF\n", tag, BM, BN, best * 1e3, 2.0 * M * N * K / best / 1e9);
}
template <int WM, int WN, int WAVES_M, int WAVES_N>
void suite(const float* A, const float* B, float* C, int M, int N, int K, const unsigned short* Bp) {
    run<WM, WN, WAVES_M, WAVES_N, ALL>("full", A, B, C, M, N, K);
    if (WAVES_N * WN * 32 == 64) run<WM, WN, WAVES_M, WAVES_N, ALL | BPRE>("full, B pre-split in HBM", A, B, C, M, N, K, Bp);
    run<WM, WN, WAVES_M, WAVES_N, ALL & ~SPLIT>("no split math", A, B, C, M, N, K);
    run<WM, WN, WAVES_M, WAVES_N, ALL & ~MFMA6>("1 of 6 mfma", A, B, C, M, N, K);
    run<WM, WN, WAVES_M, WAVES_N, ALL & ~EPI>("no output stores", A, B, C, M, N, K);
    run<WM, WN, WAVES_M, WAVES_N, ALL & ~LDSR>("no lds reads", A, B, C, M, N, K);
    run<WM, WN, WAVES_M, WAVES_N, ALL & ~LDSW & ~SPLIT>("no lds writes/split", A, B, C, M, N, K);
    run<WM, WN, WAVES_M, WAVES_N, ALL & ~GLOAD>("no global loads in loop", A, B, C, M, N, K);
    run<WM, WN, WAVES_M, WAVES_N, MFMA6 | EPI>("mfma + epilogue only", A, B, C, M, N, K);
    run<WM, WN, WAVES_M, WAVES_N, MFMA6>("mfma only", A, B, C, M, N, K);
    run<WM, WN, WAVES_M, WAVES_N, EPI>("epilogue only (1 mfma)", A, B, C, M, N, K);
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 33280, N = argc > 2 ? atoi(argv[2]) : 576, K = argc > 3 ? atoi(argv[3]) : 192;
    const int Mp = (M + 127) / 128 * 128;
    float *A, *B, *C; hipMalloc(&A, (size_t)Mp * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)Mp * N * 4);
    std::vector<float> h((size_t)Mp * K); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(((i * 2654435761u) >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    printf("M=%d N=%d K=%d\n", M, N, K);
    // pre-split copy of B: 3 planes [N][K] bf16 (truncation pieces)
    std::vector<unsigned short> hp((size_t)3 * N * K);
    for (size_t i = 0; i < (size_t)N * K; ++i) {
        float v = h[i]; unsigned u; memcpy(&u, &v, 4); unsigned short a = u >> 16; unsigned ua = (unsigned)a << 16; float fa; memcpy(&fa, &ua, 4);
        float r1 = v - fa; memcpy(&u, &r1, 4); unsigned short b2 = u >> 16; unsigned ub = (unsigned)b2 << 16; float fb; memcpy(&fb, &ub, 4);
        float r2 = r1 - fb; memcpy(&u, &r2, 4); unsigned short c3 = u >> 16;
        hp[i] = a; hp[(size_t)N * K + i] = b2; hp[(size_t)2 * N * K + i] = c3;
    }
    unsigned short* Bp; hipMalloc(&Bp, hp.size() * 2); hipMemcpy(Bp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
    // correctness of the pre-split path against the in-kernel split
    { float* C2; hipMalloc(&C2, (size_t)Mp * N * 4);
      hipLaunchKernelGGL((k<1, 1, 2, 2, ALL>), dim3(((M + 63) / 64) * (N / 64)), dim3(256), 0, 0, A, B, C, M, N, K, Bp);
      hipLaunchKernelGGL((k<1, 1, 2, 2, ALL | BPRE>), dim3(((M + 63) / 64) * (N / 64)), dim3(256), 0, 0, A, B, C2, M, N, K, Bp);
      std::vector<float> c1((size_t)1024), c2((size_t)1024); hipMemcpy(c1.data(), C, 4096, hipMemcpyDeviceToHost); hipMemcpy(c2.data(), C2, 4096, hipMemcpyDeviceToHost);
      double md = 0; for (int i = 0; i < 1024; ++i) md = fmax(md, fabs(c1[i] - c2[i])); printf("pre-split vs in-kernel split: max |diff| %g\n", md); }
    suite<1, 1, 2, 2>(A, B, C, M, N, K, Bp);
    suite<1, 2, 4, 1>(A, B, C, M, N, K, Bp);
    return 0;
}
