// Exact-erf GELU pieces for the bf16 paths (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7), shared by the
// elementwise kernels (ffn_act.hip).
#pragma once
#include "cwlt_common.h"

namespace cwlt {

// Two elements at a time on the packed-f32 VALU ops (v_pk_mul_f32 / v_pk_fma_f32: one instruction, two lanes of
// arithmetic); only |x|, v_rcp, v_exp and the sign transfer stay per element.  Same A&S 7.1.26 formula.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2& cdf, f32x2& pdf) {
    f32x2 u;
    u[0] = fabsf(x[0]);
    u[1] = fabsf(x[1]);
    u = u * 0.70710678118654752440f;
    const f32x2 den = u * 0.3275911f + 1.0f;
    f32x2 t;
    t[0] = __builtin_amdgcn_rcpf(den[0]);
    t[1] = __builtin_amdgcn_rcpf(den[1]);
    const f32x2 arg = (u * -1.44269504088896340736f) * u;          // -u^2 * log2(e)
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(arg[0]);
    e[1] = __builtin_amdgcn_exp2f(arg[1]);
    f32x2 poly = t * 1.061405429f + -1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t + -0.284496736f;
    poly = poly * t + 0.254829592f;
    poly = poly * t;
    const f32x2 erfa = 1.0f - poly * e;
    f32x2 sg;
    sg[0] = copysignf(erfa[0], x[0]);
    sg[1] = copysignf(erfa[1], x[1]);
    cdf = sg * 0.5f + 0.5f;
    pdf = e * 0.39894228040143267794f;
}

}  // namespace cwlt
