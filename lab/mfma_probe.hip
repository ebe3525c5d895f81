// Microbenchmark: cycles per f32 MFMA as a function of independent accumulator chains / waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < NACC; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long*)out)[100000] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int v = 0; v < 4; ++v) acc[i][v] = 0.f;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < NACC; ++i) for (int v = 0; v < 4; ++v) s += acc[i][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long*)out)[100000] = t1 - t0;
}
template <typename F> void run(const char* name, F launch, int nacc, int blocks_per_cu, float* d) {
    int iters = 2000;
    launch(256 * blocks_per_cu, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); launch(256 * blocks_per_cu, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long cyc; hipMemcpy(&cyc, ((long*)d) + 100000, 8, hipMemcpyDeviceToHost);
    double n_per_wave = (double)iters * 8 * nacc;
    printf("%-10s nacc=%d waves/SIMD=%d : %.1f memtime-ticks per MFMA per wave; wall %.3f ms -> %.1f ns per MFMA per SIMD\n", name, nacc,
           blocks_per_cu, cyc / n_per_wave, ms, ms * 1e6 / (n_per_wave * blocks_per_cu));
}
int main() {
    float* d; hipMalloc(&d, 4 << 20);
#define R32(N, B) run("32x32x2", [&](int g, int it) { hipLaunchKernelGGL(k32<N>, dim3(g), dim3(256), 0, 0, d, it, 1.f, 2.f); }, N, B, d)
#define R16(N, B) run("16x16x4", [&](int g, int it) { hipLaunchKernelGGL(k16<N>, dim3(g), dim3(256), 0, 0, d, it, 1.f, 2.f); }, N, B, d)
    R32(1, 1); R32(2, 1); R32(3, 1); R32(4, 1); R32(2, 2); R32(2, 3); R32(4, 2);
    R16(1, 1); R16(2, 1); R16(4, 1); R16(8, 1); R16(8, 2); R16(8, 3);
    return 0;
}
