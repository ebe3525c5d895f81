// GEMM launcher + the nn.Linear-shaped C-ABI entries built on it.
#include "gemm_f32.h"

#include <stdarg.h>

namespace vsom {

static thread_local char g_err[512] = "no error";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

// tile configuration: 0 = 128x128 (2x2 waves of 64x64), 1 = 128x64 (4x1 waves of 32x64)
static int pick_cfg(int N) {
    const double wasteA = (double)cdiv(N, 128) * 128 / N;
    const double wasteB = (double)cdiv(N, 64) * 64 / N;
    return (wasteA > 1.08 * wasteB) ? 1 : 0;
}

template <bool A_KC, bool B_KC, int EPI>
static int launch_t(const GemmP& g, int splits, hipStream_t stream) {
    const int cfg = pick_cfg(g.N);
    const int BM = 128, BN = cfg == 0 ? 128 : 64;
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    dim3 grid(tiles, 1, splits), block(256);
    if (cfg == 0)
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, 2, 2, 2, 2, EPI>), grid, block, 0, stream, g);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, 1, 2, 4, 1, EPI>), grid, block, 0, stream, g);
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
            case EPI_BIAS_RES: return launch_t<true, true, EPI_BIAS_RES>(g, splits, stream);
            case EPI_SLAB: return launch_t<true, true, EPI_SLAB>(g, splits, stream);
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
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, long stride,
                                                           int nslabs, float* __restrict__ out, long n, int vec) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (vec && i4 + 3 < n) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nslabs; ++k) s += *reinterpret_cast<const f32x4*>(slabs + (long)k * stride + i4);
        if ((reinterpret_cast<uintptr_t>(out + i4) & 15u) == 0) {
            *reinterpret_cast<f32x4*>(out + i4) = s;
        } else {
            out[i4] = s[0]; out[i4 + 1] = s[1]; out[i4 + 2] = s[2]; out[i4 + 3] = s[3];
        }
    } else {
        for (long i = i4; i < n && i < i4 + 4; ++i) {
            float s = 0.f;
            for (int k = 0; k < nslabs; ++k) s += slabs[(long)k * stride + i];
            out[i] = s;
        }
    }
}

int reduce_slabs_internal(const float* slabs, long stride, int nslabs, float* out, long n, hipStream_t stream) {
    if (n <= 0) return VSOM_OK;
    const int vec = aligned16(slabs) && (stride % 4 == 0);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(n, 1024)), dim3(256), 0, stream, slabs, stride, nslabs, out, n, vec);
    VSOM_LAUNCH_CHECK("reduce_slabs_kernel");
}

// split count for the weight-gradient reduction over M token rows
static int bwd_weight_splits(int M, int N, int K) {
    const int BN = pick_cfg(K) == 0 ? 128 : 64;
    const int tiles = cdiv(N, 128) * cdiv(K, BN);
    const int ktiles = cdiv(M, 32);
    int s = cdiv(1024, tiles);              // aim for ~4 workgroups per CU
    if (s > ktiles) s = ktiles;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    const int per = cdiv(ktiles, s);
    return cdiv(ktiles, per);
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
    const int splits = bwd_weight_splits(M, N, K);
    const long wlen = pad4((long)N * K), blen = pad4(N);
    float* slab = static_cast<float*>(ws);
    // GEMM rows = n, cols = k, reduction = m; both operands k-strided
    GemmP g = {};
    g.A = dY; g.lda = lddy; g.B = X; g.ldb = ldx;
    g.M = N; g.N = K; g.K = M;
    g.a_seg = a_seg; g.a_stride = a_stride; g.a_off = a_off;
    g.slab = slab; g.slab_stride = wlen + blen;
    g.slab_bias = db ? slab + wlen : nullptr; g.slab_bias_stride = wlen + blen;
    int rc = launch_gemm(false, false, EPI_SLAB, g, splits, stream);
    if (rc) return rc;
    rc = reduce_slabs_internal(slab, wlen + blen, splits, dW, (long)N * K, stream);
    if (rc) return rc;
    if (db) rc = reduce_slabs_internal(slab + wlen, wlen + blen, splits, db, N, stream);
    return rc;
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

int vsom_linear_gelu_fwd(const float* X, long ldx, const float* W, const float* bias, float* Ypre, float* Yact,
                         int M, int N, int K, vsom_stream_t stream) {
    VSOM_REQUIRE(X && W && Ypre && Yact, VSOM_EINVAL, "linear_gelu_fwd: null pointer");
    VSOM_REQUIRE(ldx >= K, VSOM_EINVAL, "linear_gelu_fwd: leading dimension too small");
    GemmP g = {};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = K; g.C = Ypre; g.ldc = N; g.C2 = Yact; g.ldc2 = N;
    g.M = M; g.N = N; g.K = K; g.bias = bias;
    return launch_gemm(true, true, EPI_BIAS_GELU, g, 1, stream);
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
                          int accumulate, const float* gelu_pre, vsom_stream_t stream) {
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

size_t vsom_linear_bwd_weight_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s = bwd_weight_splits(M, N, K);
    return (size_t)s * (size_t)(pad4((long)N * K) + pad4(N)) * sizeof(float);
}

int vsom_linear_bwd_weight(const float* dY, long lddy, const float* X, long ldx, float* dW, float* db, int M,
                           int N, int K, void* ws, size_t ws_bytes, vsom_stream_t stream) {
    return linear_bwd_weight_impl(dY, lddy, X, ldx, dW, db, M, N, K, 0, 0, 0, ws, ws_bytes, stream);
}

}  // extern "C"
