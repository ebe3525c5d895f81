// v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 in the regime of the split-bf16 GEMM loops: A planes in registers,
// B planes re-read from LDS with ds_read_b128 (random data), six products per fp32 product, two waves per SIMD on every CU.
// Equal FLOPs, equal LDS bytes; wall time and the in-kernel clock (s_memtime ticks per ns of s_memrealtime) for both.
//   hipcc --offload-arch=gfx950 -O3 lab/mfma_shape_probe.hip -o lab/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int STEPS = 12;          // 16-deep steps held in registers (K = 192)

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const uint4* __restrict__ src, float* out, int iters, unsigned long long* stamps) {
    __shared__ uint4 lds[STEPS * 3 * 64];            // one block of B fragments: 36 KB
    for (int i = threadIdx.x; i < STEPS * 3 * 64; i += 512) lds[i] = src[i];
    const int lane = threadIdx.x & 63;
    bf16x8 a[STEPS][3];
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[s][p] = __builtin_bit_cast(bf16x8, src[(s * 3 + p) * 64 + lane + 7]);
    __syncthreads();
    unsigned long long t0 = 0, r0 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    float total = 0.f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (SHAPE == 32) {
            f32x16 acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                bf16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, lds[(s * 3 + p) * 64 + lane]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][0], b[0], acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) total += acc[v];
        } else {
            // the same 32 x 32 x 192 block as 2 x 2 tiles of 16 x 16, 32 deep per instruction: registers a[2q], a[2q+1] hold the
            // two 16-row halves of a 32-deep A step, the LDS block the two 16-column halves of B
            f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < STEPS / 2; ++q) {
                bf16x8 b[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[j][p] = __builtin_bit_cast(bf16x8, lds[((2 * q + j) * 3 + p) * 64 + lane]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * q + i][2], b[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * q + i][0], b[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * q + i][1], b[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * q + i][1], b[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * q + i][0], b[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * q + i][0], b[j][0], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        }
    }
    unsigned long long t1 = 0, r1 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    if (total == 1.2345f) out[0] = total;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const uint4* src, int iters) {
    float* out; unsigned long long* st;
    hipMalloc(&out, 4); hipMalloc(&st, 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<SHAPE><<<256, 512>>>(src, out, iters, st);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) k<SHAPE><<<256, 512>>>(src, out, iters, st);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
    double ghz = 0; for (int b = 0; b < 256; ++b) ghz += (double)h[2 * b] / (h[2 * b + 1] * 10.0); ghz /= 256;
    const double flop = 256.0 * 8 * iters * STEPS * 6 * 32768.0;
    printf("%dx%d MFMA: %.3f ms per launch, %.0f TF bf16 (%.0f TF f32-equivalent at six products), in-kernel clock %.2f GHz\n", SHAPE, SHAPE, ms,
           flop / (ms * 1e-3) / 1e12, flop / 6 / (ms * 1e-3) / 1e12, ghz);
}
int main() {
    std::vector<unsigned> h(STEPS * 3 * 64 * 4 + 64);
    srand(1);
    for (auto& v : h) {                       // random bf16 pairs in [-1, 1): random mantissas and signs
        unsigned lo = ((rand() & 1) << 15) | ((120 + rand() % 7) << 7) | (rand() & 127), hi = ((rand() & 1) << 15) | ((120 + rand() % 7) << 7) | (rand() & 127);
        v = lo | (hi << 16);
    }
    uint4* src; hipMalloc(&src, h.size() * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) { run<32>(src, 4000); run<16>(src, 4000); }
    return 0;
}
