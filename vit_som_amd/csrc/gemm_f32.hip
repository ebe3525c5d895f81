// GEMM launcher + the nn.Linear-shaped C-ABI entries built on it.
#include "gemm_f32.h"
#include "gemm_x6.h"
#include "gemm_x6_tn.h"

#include <atomic>

#include <stdarg.h>
#include <stdlib.h>

namespace vsom {

static thread_local char g_err[512] = "no error";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

// Arithmetic of the nn.Linear-shaped GEMMs (include/vitsom_hip.h): VSOM_GEMM_F32, VSOM_GEMM_SPLIT_BF16 (exact three-piece
// split, six products everywhere) or VSOM_GEMM_SPLIT_BF16_GRAD3 (default: the same forward; the weight- and input-gradient
// GEMMs of the Linear layers on the two-piece split, three products).
static std::atomic<int> g_gemm_mode{VSOM_GEMM_SPLIT_BF16_GRAD3};
int gemm_mode() { return g_gemm_mode.load(std::memory_order_relaxed); }
static bool split_engine() { return gemm_mode() != VSOM_GEMM_F32; }
int gemm_grad_products() { return gemm_mode() == VSOM_GEMM_SPLIT_BF16_GRAD3 ? 3 : 6; }

// One tile configuration: 128 x 64 (4 waves, each 32 x 64 = two 32x32 accumulators).  Measured
// against 128 x 128 on every GEMM shape of the step (and 4096^3): faster everywhere -- three
// workgroups per CU instead of two and half the epilogue per workgroup.
static long operand_bytes(long rows, long ld, long cols) { return ((rows - 1) * ld + cols) * 4; }

// Tile height for a GEMM: 128 rows normally; the k-strided ("TN") weight-gradient GEMMs also have a
// 64 x 64 tile, used when their output height (192 = proj / fc2 rows, 96, ...) would otherwise pad
// 128-row tiles by >= 10 %.
int gemm_tile_m(bool a_kc, bool b_kc, int M) {
    if (a_kc || b_kc) return 128;
    const double w128 = (double)cdiv(M, 128) * 128, w64 = (double)cdiv(M, 64) * 64;
    return (w128 > 1.10 * w64) ? 64 : 128;
}

template <bool A_KC, bool B_KC, int EPI>
static int launch_t(GemmP& g, int splits, hipStream_t stream) {
    const int BM = gemm_tile_m(A_KC, B_KC, g.M), BN = 64;
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    dim3 grid(tiles * splits, 1, 1), block(256);
    // extent of each operand in bytes; rows of a k-strided A may be remapped (a_seg)
    const long a_rows = A_KC ? g.M : (g.a_seg ? (long)((g.K - 1) / g.a_seg) * g.a_stride + g.a_off + (g.K - 1) % g.a_seg + 1 : g.K);
    const long a_cols = A_KC ? g.K : g.M;
    const long b_rows = B_KC ? g.N : g.K, b_cols = B_KC ? g.K : g.N;
    const long ab = operand_bytes(a_rows, g.lda, a_cols), bb = operand_bytes(b_rows, g.ldb, b_cols);
    const bool fast = g.a_vec && g.b_vec && (a_cols % 4 == 0) && (b_cols % 4 == 0) && ab < 0xFFFF0000L && bb < 0xFFFF0000L;
    g.a_bytes = (unsigned)ab; g.b_bytes = (unsigned)bb;
    g.n_major = bb > ab;        // share the larger operand's panel between neighbouring workgroups
    if constexpr (!A_KC && !B_KC && (EPI == EPI_SLAB || EPI == EPI_ROWAXPY)) {
        if (fast && split_engine()) {
            if (BM == 64) VSOM_LAUNCH((gemm_x6_kernel<false, false, 1, 1, 2, 2, EPI>), grid, block, 0, stream, g);
            else if (EPI == EPI_ROWAXPY && g.products == 3) VSOM_LAUNCH((gemm_x6_kernel<false, false, 1, 2, 4, 1, EPI, 2>), grid, block, 0, stream, g);
            else VSOM_LAUNCH((gemm_x6_kernel<false, false, 1, 2, 4, 1, EPI>), grid, block, 0, stream, g);
            VSOM_LAUNCH_CHECK("gemm_x6_kernel");
        }
    }
    if constexpr (!A_KC && !B_KC) {
        if (BM == 64) {
            if (fast) VSOM_LAUNCH((gemm_f32_kernel<A_KC, B_KC, 1, 1, 2, 2, EPI, true>), grid, block, 0, stream, g);
            else VSOM_LAUNCH((gemm_f32_kernel<A_KC, B_KC, 1, 1, 2, 2, EPI, false>), grid, block, 0, stream, g);
            VSOM_LAUNCH_CHECK("gemm_f32_kernel");
        }
    }
    if constexpr (A_KC && !B_KC && EPI == EPI_ROWAXPY) {
        if (fast && split_engine()) {
            if (g.products == 3) VSOM_LAUNCH((gemm_x6_kernel<true, false, 1, 2, 4, 1, EPI, 2>), grid, block, 0, stream, g);
            else VSOM_LAUNCH((gemm_x6_kernel<true, false, 1, 2, 4, 1, EPI>), grid, block, 0, stream, g);
            VSOM_LAUNCH_CHECK("gemm_x6_kernel");
        }
    }
    if constexpr (A_KC && B_KC && EPI != EPI_SLAB) {
        if (fast && split_engine()) {
            // 128 x 64 tiles; 64 x 64 only for problems of at most 64 rows.  (Rounds 1-2 chose between the two with a
            // "rounds of 256 workgroups" model fitted to each GEMM running ALONE, which sent about half of the step's
            // GEMMs to 64 x 64.  Inside the step two kernels share the chip nearly all the time (tools/timeline.py), and
            // there the larger tile wins: every GEMM of this family on 128 x 64 is 0.3 ms per step faster, A/B on one box
            // 11.07-11.14 -> 10.77-10.83 ms; restricting 64 x 64 to launches of < 512 or < 256 large tiles: 10.91 / 10.85.)
            if (g.M <= 64) {
                dim3 grid64(cdiv(g.M, 64) * cdiv(g.N, 64) * splits, 1, 1);
                VSOM_LAUNCH((gemm_x6_kernel<true, true, 1, 1, 2, 2, EPI>), grid64, block, 0, stream, g);
                VSOM_LAUNCH_CHECK("gemm_x6_kernel");
            }
            if constexpr (EPI == EPI_NONE || EPI == EPI_GELU_BWD) {          // the input-gradient GEMMs: three products on request
                if (g.products == 3) {
                    VSOM_LAUNCH((gemm_x6_kernel<true, true, 1, 2, 4, 1, EPI, 2>), grid, block, 0, stream, g);
                    VSOM_LAUNCH_CHECK("gemm_x6_kernel");
                }
            }
            VSOM_LAUNCH((gemm_x6_kernel<true, true, 1, 2, 4, 1, EPI>), grid, block, 0, stream, g);
            VSOM_LAUNCH_CHECK("gemm_x6_kernel");
        }
    }
    if (fast)
        VSOM_LAUNCH((gemm_f32_kernel<A_KC, B_KC, 1, 2, 4, 1, EPI, true>), grid, block, 0, stream, g);
    else
        VSOM_LAUNCH((gemm_f32_kernel<A_KC, B_KC, 1, 2, 4, 1, EPI, false>), grid, block, 0, stream, g);
    VSOM_LAUNCH_CHECK("gemm_f32_kernel");
}

int launch_gemm(bool a_kc, bool b_kc, int epi, GemmP g, int splits, hipStream_t stream) {
    VSOM_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, VSOM_EINVAL, "gemm: non-positive shape M=%d N=%d K=%d", g.M, g.N, g.K);
    VSOM_REQUIRE(g.A && g.B, VSOM_EINVAL, "gemm: null operand");
    const int ktiles = cdiv(g.K, 32);
    if (splits < 1) splits = 1;
    if (splits > ktiles) splits = ktiles;
    g.ktiles_per_split = cdiv(ktiles, splits);
    splits = cdiv(ktiles, g.ktiles_per_split);
    // 16-byte vector loads need an aligned base and row stride; the vector runs along k for
    // k-contiguous operands and along the tile's columns for k-strided ones.
    g.a_vec = aligned16(g.A) && (g.lda % 4 == 0);
    g.b_vec = aligned16(g.B) && (g.ldb % 4 == 0);
    if (a_kc && b_kc) {
        switch (epi) {
            case EPI_BIAS: return launch_t<true, true, EPI_BIAS>(g, splits, stream);
            case EPI_BIAS_GELU: return launch_t<true, true, EPI_BIAS_GELU>(g, splits, stream);
            case EPI_BIAS_RELU: return launch_t<true, true, EPI_BIAS_RELU>(g, splits, stream);
            case EPI_BIAS_RES: return launch_t<true, true, EPI_BIAS_RES>(g, splits, stream);
            case EPI_SLAB: return launch_t<true, true, EPI_SLAB>(g, splits, stream);
            case EPI_NONE: return launch_t<true, true, EPI_NONE>(g, splits, stream);
            case EPI_GELU_BWD: return launch_t<true, true, EPI_GELU_BWD>(g, splits, stream);
        }
    } else if (a_kc && !b_kc) {
        switch (epi) {
            case EPI_NONE: return launch_t<true, false, EPI_NONE>(g, splits, stream);
            case EPI_GELU_BWD: return launch_t<true, false, EPI_GELU_BWD>(g, splits, stream);
            case EPI_ROWAXPY: return launch_t<true, false, EPI_ROWAXPY>(g, splits, stream);
        }
    } else if (!a_kc && !b_kc) {
        switch (epi) {
            case EPI_SLAB: return launch_t<false, false, EPI_SLAB>(g, splits, stream);
            case EPI_ROWAXPY: return launch_t<false, false, EPI_ROWAXPY>(g, splits, stream);
        }
    }
    set_error("gemm: layout/epilogue combination (%d,%d,%d) not instantiated", (int)a_kc, (int)b_kc, epi);
    return VSOM_EUNSUPPORTED;
}

// ------------------------------------------------------------------ slab reduction
// (the reducer's body lives in gemm_f32.h: the LayerNorm backward batches many of these reductions into one launch)
template <int WAVES, int VEC>
__global__ __launch_bounds__(WAVES * 64) void reduce_slabs_kernel(const float* __restrict__ slabs, long stride,
                                                                  int nslabs, float* __restrict__ out1, long n1,
                                                                  float* __restrict__ out2, long off2, long n2,
                                                                  int nb1, int vec) {
    __shared__ __attribute__((aligned(16))) f32x4 sh[WAVES][64];
    reduce_slabs_body<WAVES, VEC>(sh, (int)blockIdx.x, slabs, stride, nslabs, out1, n1, out2, off2, n2, nb1, vec);
}

int reduce_slabs2_internal(const float* slabs, long stride, int nslabs, float* out1, long n1, float* out2, long off2,
                           long n2, hipStream_t stream) {
    if (n1 <= 0 && n2 <= 0) return VSOM_OK;
    if (!out2) n2 = 0;
    const int vec = aligned16(slabs) && (stride % 4 == 0) && (off2 % 4 == 0);
    if (n1 + n2 <= 4096 && nslabs >= 32) {
        const int nb1 = cdiv(n1, 64), nb2 = n2 > 0 ? cdiv(n2, 64) : 0;
        VSOM_LAUNCH((reduce_slabs_kernel<16, 1>), dim3(nb1 + nb2), dim3(1024), 0, stream, slabs, stride, nslabs, out1,
                           n1, out2, off2, n2, nb1, vec);
    } else {
        const int nb1 = cdiv(n1, 256), nb2 = n2 > 0 ? cdiv(n2, 256) : 0;
        VSOM_LAUNCH((reduce_slabs_kernel<4, 4>), dim3(nb1 + nb2), dim3(256), 0, stream, slabs, stride, nslabs, out1, n1,
                           out2, off2, n2, nb1, vec);
    }
    VSOM_LAUNCH_CHECK("reduce_slabs_kernel");
}

int reduce_slabs_internal(const float* slabs, long stride, int nslabs, float* out, long n, hipStream_t stream) {
    return reduce_slabs2_internal(slabs, stride, nslabs, out, n, nullptr, 0, 0, stream);
}

// Split count for a reduction-split GEMM (weight gradients over the token rows, the BMU pass over
// L).  Workgroups are resident 3 per CU (register budget of the 128x64 tile), so a launch runs in
// ceil(tiles*s / slots) rounds of ceil(ktiles/s) k-tiles each; pick the s that minimises
// rounds x (k-tiles per workgroup + fixed per-workgroup cost) + the slab-reduction cost.
int choose_splits(int tiles, int ktiles, int max_splits, bool prefer_xcd_multiple) {
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
        }
        slots = 3 * cus;
    }
    if (max_splits > ktiles) max_splits = ktiles;
    if (max_splits < 1) max_splits = 1;
    // measured relative MFMA efficiency with 1 / 2 / 3 co-resident workgroups per CU
    static const double eff[4] = {1.0, 0.64, 0.80, 1.0};
    const int per_cu = slots / 3;
    double best = 1e30;
    int best_s = 1;
    for (int s = 1; s <= max_splits; ++s) {
        const int per = cdiv(ktiles, s);
        if (cdiv(ktiles, per) != s) continue;            // only canonical split counts
        const long blocks = (long)tiles * s;
        const long full = blocks / slots;                // rounds with every slot taken
        const long rest = blocks - full * slots;
        const int share = (int)((rest + per_cu - 1) / per_cu);          // 0..3 workgroups per CU in the last round
        const double passes = 3.0 * full + (share ? share / eff[share] : 0.0);
        double cost = passes * (per + 3.5) + 0.003 * tiles * s;         // + slab reduction
        if (prefer_xcd_multiple && s % 8 == 0) cost *= 0.93;             // one reduction slice per XCD: operands fetched once
        if (cost < best) { best = cost; best_s = s; }
    }
    return best_s;
}
// Weight-gradient plan: tile configuration of gemm_x6_tn_kernel (0 = none fits: the generic k-strided
// kernel) and the number of reduction splits -- a function of the shape only, so that the workspace
// query and the launch agree.
struct TnPlan { int cfg; int splits; };
static TnPlan bwd_weight_plan(int M, int N, int K) {
    TnPlan p;
    p.cfg = (N % 192 == 0 && K % 64 == 0) ? 1 : (N % 96 == 0 && K % 96 == 0) ? 2 : 0;
    if (p.cfg == 0) {
        p.splits = choose_splits(cdiv(N, gemm_tile_m(false, false, N)) * cdiv(K, 64), cdiv(M, 32), 128);
        return p;
    }
    const int tiles = p.cfg == 1 ? (N / 192) * (K / 64) : (N / 96) * (K / 96);
    const int ktiles = cdiv(M, 32);
    // 384 workgroups.  Alone the kernel is fastest with two resident workgroups per CU (512), but inside the step, where it
    // shares the chip with the backward chain, fewer and longer reduction ranges win -- and more so since the gradient GEMMs
    // run three products (round 3, lab builds A/B on one box: 768 / 512 / 384 / 320 / 256 / 192 workgroups ->
    // +0.17 / 0 / -0.08...-0.10 / 0 / +0.15 / +0.45 ms per step; rounding the count to a multiple of 8: no difference)
    int s = (384 + tiles / 2) / tiles;
    if (s > ktiles) s = ktiles;
    if (s < 1) s = 1;
    const int per = cdiv(ktiles, s);
    p.splits = cdiv(ktiles, per);
    return p;
}
static long pad4(long n) { return (n + 3) & ~3L; }


// dW[N,K] = sum_m dY[row(m), n] X[m, k] (+ db = column sums of dY rows); row(m) = optional map
int linear_bwd_weight_impl(const float* dY, long lddy, const float* X, long ldx, float* dW, float* db, int M, int N,
                           int K, int a_seg, int a_stride, int a_off, void* ws, size_t ws_bytes,
                           hipStream_t stream) {
    VSOM_REQUIRE(dY && X && dW, VSOM_EINVAL, "linear_bwd_weight: null pointer");
    VSOM_REQUIRE(lddy >= N && ldx >= K, VSOM_EINVAL, "linear_bwd_weight: leading dimension too small");
    VSOM_REQUIRE(ws && ws_bytes >= vsom_linear_bwd_weight_workspace_bytes(M, N, K), VSOM_EWORKSPACE,
                 "linear_bwd_weight: workspace too small (%zu < %zu)", ws_bytes,
                 vsom_linear_bwd_weight_workspace_bytes(M, N, K));
    VSOM_REQUIRE(aligned16(ws), VSOM_EALIGN, "linear_bwd_weight: workspace must be 16-byte aligned");
    const TnPlan plan = bwd_weight_plan(M, N, K);
    const int splits = plan.splits;
    const long wlen = pad4((long)N * K), blen = pad4(N);
    float* slab = static_cast<float*>(ws);
    const long a_last = a_seg ? (long)((M - 1) / a_seg) * a_stride + a_off + (M - 1) % a_seg : M - 1;
    const long ab = (a_last * lddy + N) * 4, bb = ((long)(M - 1) * ldx + K) * 4;
    const bool tn_ok = plan.cfg != 0 && split_engine() && aligned16(dY) && aligned16(X) &&
                       lddy % 4 == 0 && ldx % 4 == 0 && ab < 0xFFFF0000L && bb < 0xFFFF0000L &&
                       (a_seg == 0 || (a_seg % 32 == 0 && M % a_seg == 0));
    if (tn_ok) {
        TnP t = {};
        t.dY = dY; t.X = X; t.ldy = lddy; t.ldx = ldx; t.T = M; t.NO = N; t.KI = K;
        t.ktiles_per_split = cdiv(cdiv(M, 32), splits);
        t.a_seg = a_seg; t.a_stride = a_stride; t.a_off = a_off;
        t.slab = slab; t.slab_stride = wlen + blen;
        t.slab_bias = db ? slab + wlen : nullptr; t.slab_bias_stride = wlen + blen;
        t.a_bytes = (unsigned)ab; t.b_bytes = (unsigned)bb;
        const bool x3 = gemm_grad_products() == 3;
        if (plan.cfg == 1) {
            if (x3) VSOM_LAUNCH((gemm_x6_tn_kernel<3, 1, 2, 2, 2>), dim3((N / 192) * (K / 64) * splits), dim3(256), 0, stream, t);
            else VSOM_LAUNCH((gemm_x6_tn_kernel<3, 1, 2, 2, 3>), dim3((N / 192) * (K / 64) * splits), dim3(256), 0, stream, t);
        } else {
            if (x3) VSOM_LAUNCH((gemm_x6_tn_kernel<3, 1, 1, 3, 2>), dim3((N / 96) * (K / 96) * splits), dim3(192), 0, stream, t);
            else VSOM_LAUNCH((gemm_x6_tn_kernel<3, 1, 1, 3, 3>), dim3((N / 96) * (K / 96) * splits), dim3(192), 0, stream, t);
        }
        const int rc = hip_status(hipGetLastError(), "gemm_x6_tn_kernel");
        if (rc) return rc;
        return reduce_slabs2_internal(slab, wlen + blen, splits, dW, (long)N * K, db, wlen, db ? N : 0, stream);
    }
    // GEMM rows = n, cols = k, reduction = m; both operands k-strided
    GemmP g = {};
    g.A = dY; g.lda = lddy; g.B = X; g.ldb = ldx;
    g.M = N; g.N = K; g.K = M;
    g.a_seg = a_seg; g.a_stride = a_stride; g.a_off = a_off;
    g.slab = slab; g.slab_stride = wlen + blen;
    g.slab_bias = db ? slab + wlen : nullptr; g.slab_bias_stride = wlen + blen;
    int rc = launch_gemm(false, false, EPI_SLAB, g, splits, stream);
    if (rc) return rc;
    // launch_gemm may have reduced the split count (canonical form): unused slabs were never written
    const int used = cdiv(cdiv(M, 32), cdiv(cdiv(M, 32), splits));
    return reduce_slabs2_internal(slab, wlen + blen, used, dW, (long)N * K, db, wlen, db ? N : 0, stream);
}

}  // namespace vsom

using namespace vsom;

extern "C" {

int vsom_version(void) { return VSOM_VERSION; }
const char* vsom_last_error_string(void) { return vsom::last_error(); }

int vsom_reduce_slabs(const float* slabs, long stride, int nslabs, float* out, long n, vsom_stream_t stream) {
    VSOM_REQUIRE(slabs && out && nslabs > 0 && n >= 0, VSOM_EINVAL, "reduce_slabs: bad arguments");
    return reduce_slabs_internal(slabs, stride, nslabs, out, n, stream);
}

int vsom_linear_fwd(const float* X, long ldx, const float* W, const float* bias, float* Y, long ldy, int M,
                    int N, int K, vsom_stream_t stream) {
    VSOM_REQUIRE(X && W && Y, VSOM_EINVAL, "linear_fwd: null pointer");
    VSOM_REQUIRE(ldx >= K && ldy >= N, VSOM_EINVAL, "linear_fwd: leading dimension too small");
    GemmP g = {};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = K; g.C = Y; g.ldc = ldy;
    g.M = M; g.N = N; g.K = K; g.bias = bias;
    return launch_gemm(true, true, EPI_BIAS, g, 1, stream);
}

int vsom_linear_gelu_fwd(const float* X, long ldx, const float* W, const float* bias, float* Ygrad, float* Yact,
                         int M, int N, int K, vsom_stream_t stream) {
    float* Ypre = Ygrad;
    VSOM_REQUIRE(X && W && Ypre && Yact, VSOM_EINVAL, "linear_gelu_fwd: null pointer");
    VSOM_REQUIRE(ldx >= K, VSOM_EINVAL, "linear_gelu_fwd: leading dimension too small");
    GemmP g = {};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = K; g.C = Ypre; g.ldc = N; g.C2 = Yact; g.ldc2 = N;
    g.M = M; g.N = N; g.K = K; g.bias = bias;
    return launch_gemm(true, true, EPI_BIAS_GELU, g, 1, stream);
}

int vsom_linear_relu_fwd(const float* X, long ldx, const float* W, const float* bias, float* Ygrad, float* Yact,
                         int M, int N, int K, vsom_stream_t stream) {
    VSOM_REQUIRE(X && W && Ygrad && Yact, VSOM_EINVAL, "linear_relu_fwd: null pointer");
    VSOM_REQUIRE(ldx >= K, VSOM_EINVAL, "linear_relu_fwd: leading dimension too small");
    GemmP g = {};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = K; g.C = Ygrad; g.ldc = N; g.C2 = Yact; g.ldc2 = N;
    g.M = M; g.N = N; g.K = K; g.bias = bias;
    return launch_gemm(true, true, EPI_BIAS_RELU, g, 1, stream);
}

int vsom_linear_residual_fwd(const float* X, long ldx, const float* W, const float* bias, const float* R,
                             long ldr, int r_mod, float* Y, long ldy, int M, int N, int K,
                             vsom_stream_t stream) {
    VSOM_REQUIRE(X && W && R && Y, VSOM_EINVAL, "linear_residual_fwd: null pointer");
    VSOM_REQUIRE(ldx >= K && ldy >= N && ldr >= N && r_mod > 0, VSOM_EINVAL, "linear_residual_fwd: bad leading dimension / r_mod");
    GemmP g = {};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = K; g.C = Y; g.ldc = ldy;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.R = R; g.ldr = ldr; g.r_mod = r_mod; g.r_off = 0;
    return launch_gemm(true, true, EPI_BIAS_RES, g, 1, stream);
}

int vsom_linear_bwd_input(const float* dY, long lddy, const float* W, float* dX, long lddx, int M, int N, int K,
                          int accumulate, const float* gelu_grad, vsom_stream_t stream) {
    const float* gelu_pre = gelu_grad;
    VSOM_REQUIRE(dY && W && dX, VSOM_EINVAL, "linear_bwd_input: null pointer");
    VSOM_REQUIRE(lddy >= N && lddx >= K, VSOM_EINVAL, "linear_bwd_input: leading dimension too small");
    // dX[M,K] = dY[M,N] * W[N,K]: reduction over N; W is "k-strided" (rows are reduction indices)
    GemmP g = {};
    g.A = dY; g.lda = lddy; g.B = W; g.ldb = K; g.C = dX; g.ldc = lddx;
    g.M = M; g.N = K; g.K = N; g.alpha = 1.f; g.accumulate = accumulate;
    if (gelu_pre) {
        g.R = gelu_pre; g.ldr = K;
        return launch_gemm(true, false, EPI_GELU_BWD, g, 1, stream);
    }
    return launch_gemm(true, false, EPI_NONE, g, 1, stream);
}

int vsom_linear_bwd_input_t(const float* dY, long lddy, const float* Wt, float* dX, long lddx, int M, int N, int K,
                            int accumulate, const float* gelu_grad, vsom_stream_t stream) {
    VSOM_REQUIRE(dY && Wt && dX, VSOM_EINVAL, "linear_bwd_input_t: null pointer");
    VSOM_REQUIRE(lddy >= N && lddx >= K, VSOM_EINVAL, "linear_bwd_input_t: leading dimension too small");
    // dX[M,K] = dY[M,N] * Wt[K,N]^T: both operands contiguous along the reduction (N)
    GemmP g = {};
    g.A = dY; g.lda = lddy; g.B = Wt; g.ldb = N; g.C = dX; g.ldc = lddx;
    g.M = M; g.N = K; g.K = N; g.alpha = 1.f; g.accumulate = accumulate;
    g.products = gemm_grad_products();
    if (gelu_grad) {
        g.R = gelu_grad; g.ldr = K;
        return launch_gemm(true, true, EPI_GELU_BWD, g, 1, stream);
    }
    return launch_gemm(true, true, EPI_NONE, g, 1, stream);
}

int vsom_set_gemm_mode(int mode) {
    VSOM_REQUIRE(mode == VSOM_GEMM_F32 || mode == VSOM_GEMM_SPLIT_BF16 || mode == VSOM_GEMM_SPLIT_BF16_GRAD3, VSOM_EINVAL,
                 "set_gemm_mode: unknown mode %d", mode);
    g_gemm_mode.store(mode, std::memory_order_relaxed);
    return VSOM_OK;
}
int vsom_get_gemm_mode(void) { return gemm_mode(); }

size_t vsom_linear_bwd_weight_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s = bwd_weight_plan(M, N, K).splits;
    return (size_t)s * (size_t)(pad4((long)N * K) + pad4(N)) * sizeof(float);
}

int vsom_linear_bwd_weight(const float* dY, long lddy, const float* X, long ldx, float* dW, float* db, int M,
                           int N, int K, void* ws, size_t ws_bytes, vsom_stream_t stream) {
    return linear_bwd_weight_impl(dY, lddy, X, ldx, dW, db, M, N, K, 0, 0, 0, ws, ws_bytes, stream);
}

}  // extern "C"
