// erf-GELU for the bf16 paths: value and derivative from ONE v_exp_f32 and no reciprocal, shared by the elementwise
// kernels (ffn_act.hip) and the GEMM epilogues (gemm_nt.hip, gemm_ws.hip) so that fused and unfused paths agree bit
// for bit.  The f32 parity path keeps libm's erff / expf (ffn_act.hip).
//
//   e = exp(-x^2 / 2)                      (also the density: pdf = e / sqrt(2 pi), exact)
//   Phi(-|x|) = e * Q(|x|)                 Q: degree-5 minimax fit of the Mills-type ratio Phi(-a) / exp(-a^2 / 2)
//                                          on a >= 0 under the weight (1 + a) e, Q(0) = 1/2 exactly
//   Phi(x) = 1/2 + sign(x) (1/2 - e Q(|x|))
// Errors against the exact-erf forms in f32 arithmetic (tools/gelu_fit.py, all x): |gelu| and |gelu'| <= 1.5e-4 -- a
// bf16 result carries 2^-9 relative (2e-3 at |y| = 1), and where the output is small the error is too (it scales
// with x e(x)).  Round 2's Abramowitz-Stegun 7.1.26 form (1.5e-7) needed v_rcp_f32 besides the v_exp_f32 and 23
// instruction slots per element where this needs 17.
#pragma once
#include "cwlt_common.h"

namespace cwlt {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float GELU_Q1 = -3.959011294e-01f, GELU_Q2 = 2.307531891e-01f, GELU_Q3 = -9.244854863e-02f,
                GELU_Q4 = 2.139916809e-02f, GELU_Q5 = -2.076792892e-03f;
constexpr float GELU_NEG_HALF_LOG2E = -0.72134752044448170368f;     // exp(-x^2 / 2) = exp2(x * (x * this))
constexpr float GELU_INV_SQRT_2PI = 0.39894228040143267794f;

// The fit's constants times an output scale s (the dropout keep scale): results then come out already scaled.
struct GeluK {
    float c0, c1, c2, c3, c4, c5, pdfc;
};
__device__ __forceinline__ GeluK gelu_consts(float s) {
    GeluK k;
    k.c0 = 0.5f * s;
    k.c1 = GELU_Q1 * s;
    k.c2 = GELU_Q2 * s;
    k.c3 = GELU_Q3 * s;
    k.c4 = GELU_Q4 * s;
    k.c5 = GELU_Q5 * s;
    k.pdfc = GELU_INV_SQRT_2PI * s;
    return k;
}
// first half: the exponential (kept apart so that a kernel can split the work of one element over two phases)
// (x x) c, in this order: two elements' worth is two packed multiplies (gelu_scaled2), where (x c) x compiled to four
// scalar ones
__device__ __forceinline__ float gelu_expterm(float x) {
    return __builtin_amdgcn_exp2f((x * x) * GELU_NEG_HALF_LOG2E);
}
// s * Phi(x) and s * (1/2 - e Q(|x|)) >= 0
__device__ __forceinline__ void gelu_cdf(float x, float e, const GeluK& k, float& cdf, float& us) {
    const float a = fabsf(x);
    float q = fmaf(a, k.c5, k.c4);
    q = fmaf(a, q, k.c3);
    q = fmaf(a, q, k.c2);
    q = fmaf(a, q, k.c1);
    q = fmaf(a, q, k.c0);
    us = fmaf(-e, q, k.c0);
    cdf = k.c0 + copysignf(us, x);
}
// y = s * gelu(x), dy = s * gelu'(x)
__device__ __forceinline__ void gelu_scaled(float x, const GeluK& k, float& y, float& dy) {
    const float e = gelu_expterm(x);
    float cdf, us;
    gelu_cdf(x, e, k, cdf, us);
    y = fmaf(fabsf(x), us, k.c0 * x);
    dy = fmaf(e * x, k.pdfc, cdf);
}

// two elements at once, written on 2-vectors so that hipcc emits v_pk_mul_f32 / v_pk_fma_f32 throughout; the same IEEE
// operations in the same order as gelu_scaled, element by element (the elementwise kernels and the GEMM epilogues must
// agree bit for bit)
__device__ __forceinline__ void gelu_scaled2(f32x2 x, const GeluK& k, f32x2& y, f32x2& dy) {
    const f32x2 u = (x * x) * GELU_NEG_HALF_LOG2E;
    f32x2 e, a;
    e[0] = __builtin_amdgcn_exp2f(u[0]);
    e[1] = __builtin_amdgcn_exp2f(u[1]);
    a[0] = fabsf(x[0]);
    a[1] = fabsf(x[1]);
    f32x2 q = __builtin_elementwise_fma(a, f32x2{k.c5, k.c5}, f32x2{k.c4, k.c4});
    q = __builtin_elementwise_fma(a, q, f32x2{k.c3, k.c3});
    q = __builtin_elementwise_fma(a, q, f32x2{k.c2, k.c2});
    q = __builtin_elementwise_fma(a, q, f32x2{k.c1, k.c1});
    q = __builtin_elementwise_fma(a, q, f32x2{k.c0, k.c0});
    const f32x2 us = __builtin_elementwise_fma(-e, q, f32x2{k.c0, k.c0});
    f32x2 cdf;
    cdf[0] = k.c0 + copysignf(us[0], x[0]);
    cdf[1] = k.c0 + copysignf(us[1], x[1]);
    y = __builtin_elementwise_fma(a, us, x * k.c0);
    dy = __builtin_elementwise_fma(e * x, f32x2{k.pdfc, k.pdfc}, cdf);
}

}  // namespace cwlt
