// Wave-specialised split-bf16 GEMM for the Linear forward / input-gradient shapes (both operands k-contiguous, "NT").
//
// Round-2 lab result (profiles/r02_gemm_lab_findings.txt): in the two-barrier loop of gemm_x6.h the same waves load,
// split, store to LDS and issue the MFMAs, in phases separated by barriers -- the matrix pipe idles while its wave
// stages operands, and a wave stuck issuing memory instructions drains its own MFMA queue.  Here the roles are split:
//   * NL = 8 LOADER waves load both fp32 tiles (128 bytes per row and 32-deep stage, loads two stages ahead in two
//     register sets), split them into the three bf16 planes (x6_split) and store them into a TWO-SLOT LDS ring (the
//     swizzled 64-byte-row image of gemm_x6.h, 2 x 60 KB);
//   * 4 CONSUMER waves (one per SIMD: each owns its SIMD's matrix pipe) only read fragments and issue MFMAs, 72 per
//     wave between two barriers, and run the shared epilogue (gemm_epilogue: same accumulator layout).
// One barrier per 32-deep stage: barrier g publishes stage g (slot g & 1) to the consumers and releases slot (g + 1) & 1
// -- every consumer has finished reading stage g - 1 before it arrives -- to the loaders.  Workgroups are persistent
// (grid = min(tiles, CUs)) and walk their tiles in order; the loaders run ahead into the next tile while the consumers
// store.  Tile 128 x 192 (2 x 2 consumer waves of 64 x 96) or 128 x 96 (4 x 1 waves of 32 x 96, the decoder's widths);
// K % 32 == 0.  Measured alone at the c3 shapes: qkv forward 47 us (gemm_x6_kernel: 55), the fc1 / dX-fc2 shape 57 (82).
#pragma once
#include "gemm_x6.h"

namespace vsom {

template <int WM, int WN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__((WAVES_M * WAVES_N + 8) * 64) void gemm_x6_ws_kernel(const GemmP g) {
    constexpr int NL = 8;
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, NC = WAVES_M * WAVES_N, LT = NL * 64;
    constexpr int PA = BM * X6_RS, PB = BN * X6_RS, SLOT = 3 * (PA + PB);
    __shared__ __attribute__((aligned(16))) char lds[2 * SLOT];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_n = g.N / BN, tiles_m = (g.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int ns = g.K >> 5;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * ns;
    if (total <= 0) return;
    // tile j of this workgroup: row-tile major, so the column tiles of one row tile go to neighbouring workgroups
    auto tile_origin = [&](int j, int& bm0, int& bn0) {
        const int tile = blockIdx.x + j * gridDim.x;
        bm0 = (tile / tiles_n) * BM;
        bn0 = (tile % tiles_n) * BN;
    };

    if (wave >= NC) {
        // ------------------------------------------------------------------ loaders
        const int lt = t - NC * 64;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (int)g.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (int)g.b_bytes, 0x00020000);
        // float4 number f = i * LT + lt of a [rows][8] tile: row = f >> 3, 8-byte piece kq = f & 7 (k = 4 kq .. 4 kq + 3)
        constexpr int FA = (BM * 8 + LT - 1) / LT, FB = (BN * 8 + LT - 1) / LT;
        unsigned oa[FA], ob[FB];
        int la[FA], lb[FB];
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const int f = i * LT + lt, row = f >> 3, kq = f & 7;
            const bool ok = f < BM * 8;
            oa[i] = ok ? (unsigned)(((long)row * g.lda + 4 * kq) * 4) : OOB;
            la[i] = ok ? x6_piece_off(row, kq) : -1;
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
            const int f = i * LT + lt, row = f >> 3, kq = f & 7;
            const bool ok = f < BN * 8;
            ob[i] = ok ? (unsigned)(((long)row * g.ldb + 4 * kq) * 4) : OOB;
            lb[i] = ok ? 3 * PA + x6_piece_off(row, kq) : -1;
        }
        struct RS { f32x4 a[FA], b[FB]; };
        RS R0, R1;
        // rows past M / N: the tile's row offset runs past the descriptor -> the loads return zeros
        auto gload = [&](RS& R, int st) {
            st = st < total ? st : total - 1;       // unconditional (countable) loads; a stage past the end re-reads the last one
            const int j = st / ns, s2 = st - j * ns;
            int bm0, bn0;
            tile_origin(j, bm0, bn0);
            const unsigned sa = (unsigned)(((long)bm0 * g.lda + s2 * 32) * 4), sb = (unsigned)(((long)bn0 * g.ldb + s2 * 32) * 4);
#pragma unroll
            for (int i = 0; i < FA; ++i) R.a[i] = bload4s(rsA, oa[i], sa);
#pragma unroll
            for (int i = 0; i < FB; ++i) R.b[i] = bload4s(rsB, ob[i], sb);
        };
        auto lstore = [&](const RS& R, int st) {
            char* slot = lds + (st & 1) * SLOT;
#pragma unroll
            for (int i = 0; i < FA; ++i) {
                if (la[i] < 0) continue;
                uint2 p1, p2, p3;
                x6_split(R.a[i], p1, p2, p3);
                *reinterpret_cast<uint2*>(slot + la[i]) = p1;
                *reinterpret_cast<uint2*>(slot + PA + la[i]) = p2;
                *reinterpret_cast<uint2*>(slot + 2 * PA + la[i]) = p3;
            }
#pragma unroll
            for (int i = 0; i < FB; ++i) {
                if (lb[i] < 0) continue;
                uint2 p1, p2, p3;
                x6_split(R.b[i], p1, p2, p3);
                *reinterpret_cast<uint2*>(slot + lb[i]) = p1;
                *reinterpret_cast<uint2*>(slot + PB + lb[i]) = p2;
                *reinterpret_cast<uint2*>(slot + 2 * PB + lb[i]) = p3;
            }
        };
        // stage st lives in slot st & 1 and register set st & 1; the consumers read it between barrier st and barrier st + 1
        gload(R0, 0);
        gload(R1, 1);
        lstore(R0, 0);
        gload(R0, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // barrier 0
        for (int st = 1; st < total; st += 2) {
            lstore(R1, st);                                     // slot st & 1 was released by barrier st - 1
            gload(R1, st + 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // barrier st
            if (st + 1 >= total) break;
            lstore(R0, st + 1);
            gload(R0, st + 3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // barrier st + 1
        }
        return;
    }
    // ---------------------------------------------------------------------- consumers
    const int r = lane & 31, h = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WM * 32, wn0 = (wave % WAVES_N) * WN * 32;
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    int stage_in_tile = 0, jtile = 0;
    for (int st = 0; st < total; ++st) {
        __builtin_amdgcn_s_barrier();                           // barrier st: stage st is in slot st & 1
        const char* slot = lds + (st & 1) * SLOT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[WM][3], b[WN][3];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[i][pl] = *reinterpret_cast<const bf16x8*>(slot + pl * PA + x6_chunk_off(wm0 + i * 32 + r, 2 * ks + h));
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    b[j][pl] = *reinterpret_cast<const bf16x8*>(slot + 3 * PA + pl * PB + x6_chunk_off(wn0 + j * 32 + r, 2 * ks + h));
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);   // 2^-16 terms
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);   // 2^-8 terms
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);   // leading term
                    acc[i][j] = c;
                }
        }
        if (++stage_in_tile == ns) {                            // wave-uniform: the tile is complete
            int bm0, bn0;
            tile_origin(jtile, bm0, bn0);
            gemm_epilogue<WM, WN, EPI>(g, acc, bm0 + wm0, bn0 + wn0, r, h, 0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
            stage_in_tile = 0;
            ++jtile;
        }
    }
}

}  // namespace vsom
