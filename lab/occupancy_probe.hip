// How many workgroups of a given LDS size / thread count does a gfx950 CU really hold at once?  A spin kernel
// records its start on the 100 MHz clock; workgroups that start in the first microseconds are the resident set.
//   hipcc --offload-arch=gfx950 -O3 tools/occupancy_probe.hip -o tools/occupancy_probe && tools/occupancy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int VG>
__global__ __launch_bounds__(576) void spin(unsigned long long* st, int spin_ticks, float* sink) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float keep[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) keep[i] = (float)(threadIdx.x + i);
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
        for (int i = 0; i < VG; ++i) keep[i] = keep[i] * 1.0001f + lds[(threadIdx.x + i) & 63];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += keep[i];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = t0;
}

template <int VG>
static void probe(int threads, int lds_bytes) {
    const int grid = 256 * 8;
    unsigned long long* st;
    float* sink;
    hipMalloc(&st, grid * 8);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)spin<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    spin<VG><<<grid, threads, lds_bytes>>>(st, 2000, sink);   // 20 us
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), st, grid * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (auto v : h) t0 = v < t0 ? v : t0;
    int first = 0;
    for (auto v : h) first += (v - t0) < 1000;              // started within 10 us
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)spin<VG>);
    printf("VGPRs %3d  threads %3d (%d waves)  LDS %6.1f KB : %4d of %d workgroups resident at once = %.2f per CU  (%s)\n", fa.numRegs,
           threads, threads / 64, lds_bytes / 1024.0, first, grid, first / 256.0, hipGetErrorString(e));
    hipFree(st);
    hipFree(sink);
}

int main() {
    for (int kb : {16, 32, 36, 40, 48, 53, 56, 64, 72, 76, 80, 96, 128, 160}) probe<16>(320, kb * 1024 - (kb == 160 ? 0 : 0));
    for (int kb : {36, 72}) probe<16>(256, kb * 1024);
    for (int kb : {36, 72}) probe<100>(320, kb * 1024);
    // the same register budget with 3-, 4-, 5-, 6- and 8-wave workgroups (LDS small): which SIMD fills first?
    for (int th : {192, 256, 320, 384, 512}) probe<60>(th, 4096);
    for (int th : {192, 256, 320, 384, 512}) probe<100>(th, 4096);
    for (int th : {192, 256, 320, 384, 512}) probe<150>(th, 4096);
    return 0;
}
