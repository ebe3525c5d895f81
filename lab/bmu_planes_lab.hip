// Lab: the BMU contraction fed from PRE-SPLIT bf16 plane images.
//
// Operands are split once, outside the contraction, into "fragment images": for every 16-deep k step and every block of
// 32 rows, the two bf16 planes (round-to-nearest two-piece split, gemm_x6.h x3_split) as 1 KB MFMA fragments in lane order
// (lane l = (r = l & 31, h = l >> 5) holds row 32 rb + r, k = 16 s + 8 h .. + 8).  The contraction kernel then is
//     buffer_load_dwordx4 ... lds  (linear 1 KB pieces, no VGPRs, no VALU)  ->  ds_read_b128 (lane-linear)  ->  MFMA
// with a ring of NST 16-deep stages in LDS (DMA three stages ahead, counted vmcnt), one barrier per stage, and the fragments
// of stage s + 1 read under the MFMAs of stage s (register double buffer).  Same products, same order, same reduction
// split as bmu_x3_kernel<2,3,4,2>: the slabs must be bit-identical to vsom_bmu_cosine_x3_dots'.
//
//   hipcc --offload-arch=gfx950 -O3 lab/bmu_planes_lab.hip -o lab/bmu_planes_lab -ldl
//   ./lab/bmu_planes_lab [B K L]           (run from the repo root: loads vit_som_amd/libvitsom_hip.so for the comparison)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = b & 7, i = b >> 3;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + i;
}
__device__ __forceinline__ float as_float(unsigned x) { return __builtin_bit_cast(float, x); }

// 4 floats -> two planes of 4 bf16 (round to nearest even)
__device__ __forceinline__ void x3_split(f32x4 v, uint2& p1, uint2& p2) {
    const bf16x2_t a01 = {(__bf16)v[0], (__bf16)v[1]}, a23 = {(__bf16)v[2], (__bf16)v[3]};
    const unsigned u01 = __builtin_bit_cast(unsigned, a01), u23 = __builtin_bit_cast(unsigned, a23);
    const float r0 = v[0] - as_float(u01 << 16), r1 = v[1] - as_float(u01 & 0xffff0000u);
    const float r2 = v[2] - as_float(u23 << 16), r3 = v[3] - as_float(u23 & 0xffff0000u);
    const bf16x2_t b01 = {(__bf16)r0, (__bf16)r1}, b23 = {(__bf16)r2, (__bf16)r3};
    p1.x = u01; p1.y = u23;
    p2.x = __builtin_bit_cast(unsigned, b01); p2.y = __builtin_bit_cast(unsigned, b23);
}

// ---------------------------------------------------------------------------------------------- image writer
// image[(s * nrb + rb) * 2 + plane][lane][16 B]; rows >= R and k >= L are zero.  One thread per (s, rb, lane).
__global__ __launch_bounds__(256) void split_image_kernel(const float* __restrict__ src, long ld, int R, int L, int nrb, int nst,
                                                          uint4* __restrict__ img) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)nst * nrb * 64) return;
    const int lane = (int)(idx & 63);
    const long f = idx >> 6;
    const int rb = (int)(f % nrb), s = (int)(f / nrb);
    const int row = rb * 32 + (lane & 31), k = s * 16 + 8 * (lane >> 5);
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (row < R) {
        const float* p = src + (long)row * ld + k;
        if (k + 3 < L) v0 = *reinterpret_cast<const f32x4*>(p);
        if (k + 7 < L) v1 = *reinterpret_cast<const f32x4*>(p + 4);
    }
    uint2 a1, a2, b1, b2;
    x3_split(v0, a1, a2);
    x3_split(v1, b1, b2);
    img[(f * 2 + 0) * 64 + lane] = uint4{a1.x, a1.y, b1.x, b1.y};
    img[(f * 2 + 1) * 64 + lane] = uint4{a2.x, a2.y, b2.x, b2.y};
}

// ---------------------------------------------------------------------------------------------- contraction
struct PlP {
    const void* ximg; const void* wimg;
    unsigned ximg_bytes, wimg_bytes;
    int nrb_x, nrb_w;                   // 32-row blocks of X (B / 32) and W (K / 32), rounded up
    int B, K, nst;                      // nst = 16-deep stages in total (ceil(L / 32) * 2)
    int stages_per_split;
    float* slab; long slab_stride;
    unsigned long long* stamps;         // [4]: memtime / memrealtime around the stage loop of workgroup 0, wave 0
    int variant;                        // ablations: 1 = no MFMA, 2 = no DMA after the prologue, 4 = no fragment reads
};

__device__ __forceinline__ i32x4 raw_srd(const void* base, unsigned bytes) {
    const unsigned long a = (unsigned long)base;
    return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// one 1 KB piece global -> LDS (lane l: 16 B from voff to lds_dst + 16 l).  In assembly so that hipcc does not order the
// fragment reads behind it with vmcnt(0); the waits are placed by hand.
__device__ __forceinline__ void dma16(i32x4 rs, unsigned lds_dst, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const char* p) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)p;
}
template <int OFF>
__device__ __forceinline__ void frag_read(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

template <int WM, int WN>
struct Frags { bf16x8 a[WM][2], b[WN][2]; };

template <int WM, int WN>
__device__ __forceinline__ void frags_issue(Frags<WM, WN>& f, unsigned aaddr, unsigned baddr) {
    if constexpr (WM == 2) {
        frag_read<0>(f.a[0][0], aaddr); frag_read<1024>(f.a[0][1], aaddr);
        frag_read<2048>(f.a[1][0], aaddr); frag_read<3072>(f.a[1][1], aaddr);
    }
    frag_read<0>(f.b[0][0], baddr); frag_read<1024>(f.b[0][1], baddr);
    frag_read<2048>(f.b[1][0], baddr); frag_read<3072>(f.b[1][1], baddr);
    if constexpr (WN >= 3) { frag_read<4096>(f.b[2][0], baddr); frag_read<5120>(f.b[2][1], baddr); }
    if constexpr (WN >= 4) { frag_read<6144>(f.b[3][0], baddr); frag_read<7168>(f.b[3][1], baddr); }
}
// all fragment reads of this wave have returned; the registers pass through the statement so that no MFMA can move above it
template <int WM, int WN>
__device__ __forceinline__ void frags_wait(Frags<WM, WN>& f) {
    if constexpr (WN == 3)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]),
                     "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.b[2][0]), "+v"(f.b[2][1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]),
                     "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.b[2][0]), "+v"(f.b[2][1]), "+v"(f.b[3][0]), "+v"(f.b[3][1]));
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int NST, bool PAIR>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (WAVES_M * WAVES_N == 4 ? 2 : 1)) void bmu_pl_kernel(const PlP g) {
    constexpr int NW = WAVES_M * WAVES_N, RA = WAVES_M * WM, RB = WAVES_N * WN;      // 32-row blocks per tile
    constexpr int BM = RA * 32, BN = RB * 32;
    constexpr int STAGE = (RA + RB) * 2048, PIECES = (RA + RB) * 2, PPW = (PIECES + NW - 1) / NW;
    constexpr unsigned NOWHERE = 0x80000000u;
    constexpr int AHEAD = NST - 1;                  // stages in flight towards LDS
    extern __shared__ __attribute__((aligned(1024))) char lds[];            // NST stages + 1 KB that absorbs the filler pieces
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wmi = wave / WAVES_N, wni = wave % WAVES_N;
    const int tiles_n = (g.K + BN - 1) / BN, tiles_m = (g.B + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int z = lid / ntiles, rem = lid - z * ntiles;
    const int tm = rem % tiles_m, tn = rem / tiles_m;
    const int s_begin = z * g.stages_per_split;
    int s_end = s_begin + g.stages_per_split;
    if (s_end > g.nst) s_end = g.nst;
    const int S = s_end - s_begin;

    const i32x4 rsX = raw_srd(g.ximg, g.ximg_bytes), rsW = raw_srd(g.wimg, g.wimg_bytes);
    const unsigned lds0 = lds_addr(lds);
    // this wave's pieces: q = i NW + wave; q < 2 RA: X piece q, else W piece q - 2 RA; beyond PIECES: filler
    unsigned pv[PPW], pd[PPW], pstep[PPW];
    bool px[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = i * NW + wave;
        px[i] = q < 2 * RA;
        const int qq = px[i] ? q : q - 2 * RA;
        const int rb0 = px[i] ? tm * RA : tn * RB, nrb = px[i] ? g.nrb_x : g.nrb_w;
        const bool ok = q < PIECES && rb0 + (qq >> 1) < nrb;
        // image offset of the piece at stage 0 and its step per stage
        pv[i] = ok ? (unsigned)(((long)rb0 * 2 + qq) * 1024 + lane * 16) : NOWHERE;
        pstep[i] = ok ? (unsigned)nrb * 2048u : 0u;
        pd[i] = q < PIECES ? (unsigned)(q * 1024) : (unsigned)(NST * STAGE);
    }
    auto dma_stage = [&](int s_rel) {               // stage s_begin + s_rel -> ring slot s_rel % NST
        const int s_abs = s_begin + s_rel;
        const bool live = s_rel < S;
        const unsigned slot = lds0 + (unsigned)(s_rel % NST) * STAGE;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const unsigned off = (live && pv[i] != NOWHERE) ? pv[i] + (unsigned)s_abs * pstep[i] : NOWHERE;
            dma16(px[i] ? rsX : rsW, (pd[i] == (unsigned)(NST * STAGE) ? lds0 : slot) + pd[i], off);
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    const unsigned a_in_stage = (unsigned)(wmi * WM * 2048 + lane * 16), b_in_stage = (unsigned)(RA * 2048 + wni * WN * 2048 + lane * 16);
    Frags<WM, WN> f0, f1;
    // product P of the three (a2 b1, a1 b2, a1 b1 -- smallest first) for all tiles of the wave: independent MFMAs
    auto mfma_group = [&](const Frags<WM, WN>& f, auto P) {
        constexpr int PA_ = P.value == 0 ? 1 : 0, PB_ = P.value == 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][PA_], f.b[j][PB_], acc[i][j], 0, 0, 0);
    };
    // stage s: its fragments are on their way into `cur`; stages s + 1 .. s + AHEAD - 1 are in flight towards LDS.
    // Everything that is not an MFMA is issued BETWEEN the three MFMA groups, in their shadow.
    auto step = [&](int s, Frags<WM, WN>& cur, Frags<WM, WN>& nxt) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((AHEAD - 2) * PPW) : "memory");   // this wave's pieces of stage s + 1 have landed
        __builtin_amdgcn_s_barrier();                                          // everybody's have; nobody reads slot (s - 1) % NST any more
        frags_wait<WM, WN>(cur);
        __builtin_amdgcn_sched_barrier(0);
        if (!(g.variant & 1)) mfma_group(cur, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        const unsigned slot = lds0 + (unsigned)((s + 1) % NST) * STAGE;
        if (!(g.variant & 4)) frags_issue<WM, WN>(nxt, slot + a_in_stage, slot + b_in_stage);
        __builtin_amdgcn_sched_barrier(0);
        if (!(g.variant & 1)) mfma_group(cur, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if (!(g.variant & 2)) dma_stage(s + AHEAD);
        __builtin_amdgcn_sched_barrier(0);
        if (!(g.variant & 1)) mfma_group(cur, std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
    };

    unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    if constexpr (!PAIR) {
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) dma_stage(a);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((AHEAD - 1) * PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        frags_issue<WM, WN>(f0, lds0 + a_in_stage, lds0 + b_in_stage);
        if (g.variant & 4) frags_issue<WM, WN>(f1, lds0 + a_in_stage, lds0 + b_in_stage);
        for (int s = 0; s < S; s += 2) {
            step(s, f0, f1);
            if (s + 1 < S) step(s + 1, f1, f0);
        }
    } else {
        // one barrier per TWO stages (S is even).  At the barrier of iteration b: stages <= 2 b + 2 have landed, every fragment
        // read of slots <= 2 b has returned; then stages 2 b + NST - 1 and 2 b + NST are issued into the slots of 2 b - 1, 2 b.
        auto half = [&](int s, Frags<WM, WN>& cur, Frags<WM, WN>& nxt, int dma_s) {
            __builtin_amdgcn_sched_barrier(0);
            if (!(g.variant & 1)) mfma_group(cur, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            const unsigned slot = lds0 + (unsigned)((s + 1) % NST) * STAGE;
            if (!(g.variant & 4)) frags_issue<WM, WN>(nxt, slot + a_in_stage, slot + b_in_stage);
            __builtin_amdgcn_sched_barrier(0);
            if (!(g.variant & 1)) mfma_group(cur, std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            if (!(g.variant & 2)) dma_stage(dma_s);
            __builtin_amdgcn_sched_barrier(0);
            if (!(g.variant & 1)) mfma_group(cur, std::integral_constant<int, 2>{});
            __builtin_amdgcn_sched_barrier(0);
            frags_wait<WM, WN>(nxt);
        };
#pragma unroll
        for (int a = 0; a < NST - 1; ++a) dma_stage(a);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 4) * PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        frags_issue<WM, WN>(f0, lds0 + a_in_stage, lds0 + b_in_stage);
        if (g.variant & 4) frags_issue<WM, WN>(f1, lds0 + a_in_stage, lds0 + b_in_stage);
        frags_wait<WM, WN>(f0);
        for (int s = 0; s < S; s += 2) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 4) * PPW) : "memory");
            __builtin_amdgcn_s_barrier();
            half(s, f0, f1, s + NST - 1);
            half(s + 1, f1, f0, s + NST);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    if (g.stamps && blockIdx.x == 0 && t == 0) { g.stamps[0] = t1 - t0; g.stamps[1] = r1 - r0; g.stamps[2] = (unsigned long long)S; }
    if (g.variant & 8) return;

    float* sl = g.slab + (long)z * g.slab_stride;
    const int bm0 = tm * BM, bn0 = tn * BN, wm0 = wmi * WM * 32, wn0 = wni * WN * 32;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = bn0 + wn0 + j * 32 + r;
            if (n >= g.K) continue;
            const int mb = bm0 + wm0 + i * 32 + 4 * h;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = mb + (v & 3) + 8 * (v >> 2);
                if (m < g.B) sl[(long)m * g.K + n] = acc[i][j][v];
            }
        }
}

// ---------------------------------------------------------------------------------------------- host
static int cdiv(int a, int b) { return (a + b - 1) / b; }

template <int WM, int WN, int WAVES_M, int WAVES_N, int NST, bool PAIR = false>
static float run(const char* name, PlP g, int splits, const float* ref, size_t nref, int reps) {
    constexpr int RA = WAVES_M * WM, RB = WAVES_N * WN;
    constexpr size_t LDS = (size_t)NST * (RA + RB) * 2048 + (((RA + RB) * 2) % (WAVES_M * WAVES_N) ? 1024 : 0);
    auto kern = bmu_pl_kernel<WM, WN, WAVES_M, WAVES_N, NST, PAIR>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    const int tiles = cdiv(g.B, RA * 32) * cdiv(g.K, RB * 32);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(g.slab, 0xff, (size_t)splits * g.slab_stride * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(WAVES_M * WAVES_N * 64), LDS, 0, g);
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(WAVES_M * WAVES_N * 64), LDS, 0, g);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / 10); sum += ms / 10;
    }
    CK(hipGetLastError());
    double maxdiff = -1;
    if (ref && g.variant == 0) {
        std::vector<float> got(nref);
        CK(hipMemcpy(got.data(), g.slab, nref * 4, hipMemcpyDeviceToHost));
        maxdiff = 0;
        for (size_t i = 0; i < nref; ++i) { const double d = fabs((double)got[i] - ref[i]); if (!(d <= maxdiff)) maxdiff = d; }
    }
    const double gflop = 2.0 * g.B * g.K * (double)g.nst * 16 * 1e-9;
    printf("%-44s tiles %3d x %2d splits  best %7.1f us  avg %7.1f us  %6.1f TF f32-equivalent", name, tiles, splits, best * 1e3, sum / reps * 1e3,
           gflop / best);
    if (maxdiff >= 0) printf("   maxdiff vs production slabs %.3g", maxdiff);
    if (g.stamps) {
        unsigned long long st[3];
        CK(hipMemcpy(st, g.stamps, sizeof st, hipMemcpyDeviceToHost));
        printf("\n      loop of workgroup 0: %llu stages, %.0f shader cycles per stage, %.2f us in all, clock %.2f GHz", st[2], (double)st[0] / st[2],
               st[1] * 0.01, (double)st[0] / (st[1] * 10.0));
    }
    printf("\n");
    return best;
}

int main(int argc, char** argv) {
    const int B = argc > 3 ? atoi(argv[1]) : 512, K = argc > 3 ? atoi(argv[2]) : 1600, L = argc > 3 ? atoi(argv[3]) : 12288;
    void* lib = dlopen("vit_som_amd/libvitsom_hip.so", RTLD_NOW);
    if (!lib) { printf("cannot load vit_som_amd/libvitsom_hip.so: %s\n", dlerror()); return 1; }
    auto ws_bytes = (size_t (*)(int, int, int))dlsym(lib, "vsom_bmu_cosine_x3_workspace_bytes");
    auto dots = (int (*)(const float*, long, const float*, int, int, int, void*, size_t, void*))dlsym(lib, "vsom_bmu_cosine_x3_dots");
    std::vector<float> hx((size_t)B * L), hw((size_t)K * L);
    srand(1);
    for (auto& v : hx) v = (float)(rand() & 0xffffff) / 16777216.f - 0.5f;
    for (auto& v : hw) v = ((float)(rand() & 0xffffff) / 16777216.f - 0.5f) * 0.1f;
    float *X, *W;
    CK(hipMalloc(&X, hx.size() * 4)); CK(hipMalloc(&W, hw.size() * 4));
    CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));

    // production kernel: time and slabs
    const size_t wsb = ws_bytes(B, K, L);
    void* ws; CK(hipMalloc(&ws, wsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) dots(X, L, W, B, K, L, ws, wsb, nullptr);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) dots(X, L, W, B, K, L, ws, wsb, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 10);
    }
    // the production split rule (bmu_x3.hip: bmu_x3_splits, 256 x 192 tiles)
    const int ktiles = cdiv(L, 32), tiles = cdiv(B, 256) * cdiv(K, 192);
    int s = 256 / tiles; if (s > ktiles) s = ktiles; if (s > 64) s = 64; if (s < 1) s = 1;
    const int per = cdiv(ktiles, s), splits = cdiv(ktiles, per);
    const size_t nref = (size_t)splits * B * K;
    std::vector<float> ref(nref);
    CK(hipMemcpy(ref.data(), ws, nref * 4, hipMemcpyDeviceToHost));
    printf("B %d K %d L %d: production bmu_x3_kernel<2,3,4,2> %d splits of %d k-tiles  best %.1f us\n", B, K, L, splits, per, best * 1e3);

    // images
    const int nst = ktiles * 2, nrb_x = cdiv(B, 32), nrb_w = cdiv(K, 32);
    const size_t xb = (size_t)nst * nrb_x * 2048, wb = (size_t)nst * nrb_w * 2048;
    void *ximg, *wimg;
    CK(hipMalloc(&ximg, xb)); CK(hipMalloc(&wimg, wb));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(split_image_kernel, dim3((unsigned)(((long)nst * nrb_x * 64 + 255) / 256)), dim3(256), 0, 0, X, (long)L, B, L, nrb_x, nst, (uint4*)ximg);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float m1; CK(hipEventElapsedTime(&m1, e0, e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(split_image_kernel, dim3((unsigned)(((long)nst * nrb_w * 64 + 255) / 256)), dim3(256), 0, 0, W, (long)L, K, L, nrb_w, nst, (uint4*)wimg);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float m2; CK(hipEventElapsedTime(&m2, e0, e1));
        if (rep) printf("image writers (separate kernels): X %.1f us, W %.1f us\n", m1 * 1e3, m2 * 1e3);
    }
    float* slab; CK(hipMalloc(&slab, nref * 4 * 2));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 64));
    PlP g = {};
    g.ximg = ximg; g.wimg = wimg; g.ximg_bytes = (unsigned)xb; g.wimg_bytes = (unsigned)wb;
    g.nrb_x = nrb_x; g.nrb_w = nrb_w; g.B = B; g.K = K; g.nst = nst; g.stages_per_split = per * 2;
    g.slab = slab; g.slab_stride = (long)B * K; g.stamps = stamps;
    run<2, 3, 4, 2, 4>("planes 256x192 8 waves, 4-stage ring", g, splits, ref.data(), nref, 6);
    g.variant = 8; run<2, 3, 4, 2, 4>("  no slab stores", g, splits, nullptr, 0, 3);
    g.variant = 14; run<2, 3, 4, 2, 4>("  MFMA only, no slab stores", g, splits, nullptr, 0, 3);
    g.variant = 0;
    run<2, 3, 4, 2, 4, true>("256x192 8 waves, barrier per 2 stages, ring 4", g, splits, ref.data(), nref, 6);
    run<2, 3, 4, 2, 5, true>("256x192 8 waves, barrier per 2 stages, ring 5", g, splits, ref.data(), nref, 6);
    g.variant = 8; run<2, 3, 4, 2, 5, true>("  no slab stores", g, splits, nullptr, 0, 3);
    g.variant = 6; run<2, 3, 4, 2, 5, true>("  MFMA only", g, splits, nullptr, 0, 3);
    g.variant = 14; run<2, 3, 4, 2, 5, true>("  MFMA only, no slab stores", g, splits, nullptr, 0, 3);
    g.variant = 2; run<2, 3, 4, 2, 5, true>("  no DMA in the loop", g, splits, nullptr, 0, 3);
    g.variant = 4; run<2, 3, 4, 2, 5, true>("  no fragment reads", g, splits, nullptr, 0, 3);
    g.variant = 0;
    run<2, 3, 2, 2, 4, true>("128x192 4 waves x2, barrier per 2 stages, ring 4", g, splits, ref.data(), nref, 6);
    g.variant = 0;
    {   // 256 x 256 tiles: other split -> not comparable bit for bit; compared after summing the slabs on the host instead
        const int tiles2 = cdiv(B, 256) * cdiv(K, 256);
        int s2 = 256 / tiles2; if (s2 > ktiles) s2 = ktiles; if (s2 > 64) s2 = 64;
        const int per2 = cdiv(ktiles, s2), splits2 = cdiv(ktiles, per2);
        g.stages_per_split = per2 * 2;
        if ((size_t)splits2 * B * K <= nref * 2) run<2, 4, 4, 2, 4>("planes 256x256, 4-stage ring", g, splits2, nullptr, 0, 6);
    }
    return 0;
}
