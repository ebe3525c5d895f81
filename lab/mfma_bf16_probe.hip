// Issue rate of v_mfma_f32_32x32x16_bf16 for the accumulation patterns of the split-bf16 kernels: NACC accumulators,
// CHAIN back-to-back dependent MFMAs on one accumulator before moving to the next (the x3 / x6 kernels use 3 / 6).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_probe.hip -o tools/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int CHAIN>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) s += acc[i][v];
    if (s == 1.2345f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int CHAIN>
static void run(int waves_per_simd) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, CHAIN><<<256 * waves_per_simd, 256>>>(out, 10, cyc);
    hipEventRecord(e0);
    k<NACC, CHAIN><<<256 * waves_per_simd, 256>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC * CHAIN;
    const double tf = 256.0 * 4 * waves_per_simd * n * 32768.0 / (ms * 1e-3) / 1e12;
    printf("accumulators %2d, chain %d, waves/SIMD %d: %.1f memtime ticks per MFMA per wave, %.1f ns per MFMA per SIMD, %.0f TF aggregate\n", NACC, CHAIN,
           waves_per_simd, (double)c / n, ms * 1e6 / (n * waves_per_simd), tf);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1, 1>(1); run<2, 1>(1); run<4, 1>(1); run<12, 1>(1);
    run<12, 3>(1); run<12, 6>(1); run<6, 3>(1); run<4, 6>(1); run<3, 6>(1);
    run<12, 3>(2); run<6, 3>(2); run<3, 6>(2); run<4, 1>(2);
    return 0;
}
