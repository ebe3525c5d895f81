// Shared host/device helpers for libvitsom_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <functional>
#include <utility>

#include "../../include/vitsom_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace vsom {

// thread-local last-error text (vsom_last_error_string)
void set_error(const char* fmt, ...);

inline int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return VSOM_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

#define VSOM_REQUIRE(cond, code, ...)      \
    do {                                   \
        if (!(cond)) {                     \
            ::vsom::set_error(__VA_ARGS__); \
            return (code);                 \
        }                                  \
    } while (0)

#define VSOM_LAUNCH_CHECK(name) return ::vsom::hip_status(hipGetLastError(), name)

// ---- launch tape (tape.hip): every kernel launch of the library goes through VSOM_LAUNCH.  Normally that IS
// hipLaunchKernelGGL; while the calling thread records a tape (vsom_tape_begin) the launch is also kept -- kernel, grid, block,
// LDS size, stream and the by-value arguments, all inside one closure -- so that vsom_tape_replay can re-issue a whole
// training step's launches from C in one call (the host mirror's Python + ctypes path costs ~9 us per launch, 4 ms per
// step: host-bound below ~256 images per GPU).
struct TapeRec;
extern thread_local TapeRec* g_tape_rec;                    // non-null while this thread records (and is not paused)
void tape_push(std::function<void()>&& op);

template <class F>
inline void launch_or_record(F&& f) {
    f();
    if (g_tape_rec) tape_push(std::function<void()>(std::forward<F>(f)));
}
#define VSOM_LAUNCH(kernel, grid, block, lds, stream, ...)                                       \
    ::vsom::launch_or_record([=]() { hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); })

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// nn.GELU() (exact-erf form) and its derivative from ONE exponential:
//   gelu(x) = x Phi(x),  gelu'(x) = Phi(x) + x phi(x),  Phi(x) = (1 + erf(x/sqrt2))/2,
//   phi(x) = exp(-x^2/2)/sqrt(2 pi)  -- and erf(x/sqrt2) = 1 - poly(t) exp(-x^2/2) with the SAME
// exponential (Abramowitz-Stegun 7.1.26, |erf error| <= 1.5e-7, i.e. <= 7.5e-8 on Phi: below
// fp32 rounding of the surrounding GEMMs and far inside the 1e-4 parity bar).
__device__ __forceinline__ void gelu_erf_both(float x, float& act, float& grad) {
    const float ax = fabsf(x);
    const float e = __expf(-0.5f * x * x);
    const float t = __frcp_rn(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc_abs = p * t * e;                    // 1 - erf(|x|/sqrt2)
    const float cdf = (x >= 0.f) ? 1.0f - 0.5f * erfc_abs : 0.5f * erfc_abs;
    act = x * cdf;
    grad = fmaf(x * 0.39894228040143267794f, e, cdf);
}

// XCD-aware bijective block remap (MI355X: 8 XCDs, blocks dealt round-robin).  Blocks that share
// an XCD (equal b % 8) get a contiguous range of logical ids, so tiles that share an operand
// panel hit the same per-XCD L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = b & 7, i = b >> 3;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + i;
}

}  // namespace vsom
