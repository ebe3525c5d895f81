// The AdamW / Adam element update (torch/optim/adam.py single-tensor path), shared by the flat-arena kernel (misc.hip) and
// the prototype-slice kernel that also emits the BMU plane images (bmu_x3.hip): one definition, bitwise the same result.
#pragma once
#include "gemm_f32.h"

#include <math.h>

namespace vsom {

struct AdamwC {
    float lr, b1, b2, eps, step_size, inv_bc2_sqrt, gscale;
    int adamw;
};
inline AdamwC adamw_constants(float lr, float beta1, float beta2, float eps, int step, float grad_scale, int adamw) {
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    return AdamwC{lr, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, adamw};
}
__device__ __forceinline__ void adamw_update(f32x4& pp, const f32x4& gg, f32x4& mm, f32x4& vv, float wd, const AdamwC& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float gr = gg[e] * c.gscale;
        float pe = pp[e];
        if (c.adamw) pe *= (1.f - c.lr * wd); else gr = fmaf(wd, pe, gr);
        const float me = mm[e] + (gr - mm[e]) * (1.f - c.b1);          // lerp_, torch/optim/adam.py
        const float ve = fmaf(vv[e], c.b2, (1.f - c.b2) * gr * gr);
        const float denom = sqrtf(ve) * c.inv_bc2_sqrt + c.eps;
        pp[e] = pe - c.step_size * (me / denom);
        mm[e] = me; vv[e] = ve;
    }
}

}  // namespace vsom
