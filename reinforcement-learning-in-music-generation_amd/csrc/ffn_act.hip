// Fused  g = dropout(gelu(h + bias))  for the FFN inner activation, fwd + bwd (+ bias gradient).
//
// Replaces `self.dropout(self.activation(self.linear1(y)))` of fast_transformers'
// TransformerEncoderLayer (activation='gelu' -> F.gelu, exact erf; dropout 0.1) reached from
// /root/reference/dqn_policy/model.py:128-137.  The Linear runs without bias (plain GEMM); the bias
// add, the activation and the dropout are one pass over the (rows, d_ff) tensor, and in backward the
// same pass accumulates the bias gradient (column sums) in registers.  HBM-bound.
//
// Mapping: a thread owns 4 adjacent columns (16 B f32 / 8 B bf16 per access, wave-wide coalesced)
// and walks a contiguous slab of rows; grid = (column tiles of 1024, row slabs).
#include "cwlt_common.h"
#include "cwlt_gelu.h"

namespace cwlt {

// exact-erf GELU (F.gelu default).  f32 storage (parity path): libm-accurate erff / expf.
// bf16 storage: cwlt_gelu.h (one v_exp_f32 per element, |err| <= 1.5e-4 against 2e-3 of bf16 resolution at |y| = 1).
__device__ __forceinline__ void gelu_parts_exact(float x, float& cdf, float& pdf) {
    cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
}

// WANT_GD: also write gd = mask * keep_scale * gelu'(h + bias), the factor the backward multiplies the upstream gradient
// with (consumed by the epilogue of cwlt_gemm_nt_mul).  gd may alias h (each thread reads its chunk of h before it
// writes the same chunk of gd), which is how the encoder uses it: h itself is not needed again.
template <typename T, bool WANT_GD>
__global__ __launch_bounds__(256) void bias_gelu_dropout_fwd_kernel(const T* h, const float* __restrict__ bias,
                                                                    T* __restrict__ g, T* gd, long rows, int F,
                                                                    uint32_t thresh, float keep_scale, uint64_t seed,
        const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += *seed_base;   // device-resident offset: lets a captured hipGraph draw fresh masks per replay
    constexpr int V = VecIO<T>::N;
    constexpr bool FAST = sizeof(T) == 2;
    const GeluK gk = gelu_consts(keep_scale);
    const int ci = blockIdx.x * 256 + threadIdx.x;
    if (ci * V >= F) return;
    float b[V];
#pragma unroll
    for (int j = 0; j < V; ++j) b[j] = 0.f;
    if (bias) loadf<V>(bias + ci * V, b);
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
#pragma unroll 4
    for (long r = r0; r < r1; ++r) {
        const long off = r * F + ci * V;
        float t[V], d[V];
        VecIO<T>::load(h + off, t);
        const uint32_t km = thresh ? dropout_mask<V>(seed, off, thresh) : 0xffffffffu;
        if (FAST) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float y, dy;
                gelu_scaled(t[j] + b[j], gk, y, dy);
                t[j] = ((km >> j) & 1u) ? y : 0.f;
                if (WANT_GD) d[j] = ((km >> j) & 1u) ? dy : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float x = t[j] + b[j];
                float cdf, pdf;
                gelu_parts_exact(x, cdf, pdf);
                t[j] = ((km >> j) & 1u) ? x * cdf * keep_scale : 0.f;
                if (WANT_GD) d[j] = ((km >> j) & 1u) ? (cdf + x * pdf) * keep_scale : 0.f;
            }
        }
        VecIO<T>::store(g + off, t);
        if (WANT_GD) VecIO<T>::store(gd + off, d);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bias_gelu_dropout_bwd_kernel(const T* __restrict__ dg, const T* __restrict__ h,
                                                                    const float* __restrict__ bias, T* __restrict__ dh,
                                                                    float* __restrict__ part, long rows, int F,
                                                                    uint32_t thresh, float keep_scale, uint64_t seed,
        const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += *seed_base;   // device-resident offset: lets a captured hipGraph draw fresh masks per replay
    constexpr int V = VecIO<T>::N;
    constexpr bool FAST = sizeof(T) == 2;
    const GeluK gk = gelu_consts(keep_scale);
    const int ci = blockIdx.x * 256 + threadIdx.x;
    if (ci * V >= F) return;
    float b[V], acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) b[j] = acc[j] = 0.f;
    if (bias) loadf<V>(bias + ci * V, b);
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
#pragma unroll 2
    for (long r = r0; r < r1; ++r) {
        const long off = r * F + ci * V;
        float d[V], t[V];
        VecIO<T>::load(dg + off, d);
        VecIO<T>::load(h + off, t);
        const uint32_t km = thresh ? dropout_mask<V>(seed, off, thresh) : 0xffffffffu;
        if (FAST) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float y, dy;
                gelu_scaled(t[j] + b[j], gk, y, dy);
                d[j] = ((km >> j) & 1u) ? d[j] * dy : 0.f;
                acc[j] += d[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float x = t[j] + b[j];
                float cdf, pdf;
                gelu_parts_exact(x, cdf, pdf);
                d[j] = ((km >> j) & 1u) ? d[j] * keep_scale * (cdf + x * pdf) : 0.f;
                acc[j] += d[j];
            }
        }
        VecIO<T>::store(dh + off, d);
    }
    if (part) {
#pragma unroll
        for (int j = 0; j < V; ++j) part[(long)blockIdx.y * F + ci * V + j] = acc[j];
    }
}

// y = dropout(x + pe[t]) with pe (max_len, D) f32, row r of x is position r % T   (model.py:90-92)
template <typename T>
__global__ __launch_bounds__(256) void posenc_dropout_kernel(const T* __restrict__ x, const float* __restrict__ pe,
                                                             T* __restrict__ y, long rows, int Tlen, int D,
                                                             uint32_t thresh, float keep_scale, uint64_t seed,
        const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += *seed_base;   // device-resident offset: lets a captured hipGraph draw fresh masks per replay
    constexpr int V = VecIO<T>::N;
    const int nd = D / V;
    const long nv = rows * (long)nd;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        const long r = i / nd;
        const int ci = (int)(i - r * nd);
        const long off = r * D + ci * V;
        float t[V];
        VecIO<T>::load(x + off, t);
        if (pe) {
            float p[V];
            loadf<V>(pe + (r % Tlen) * (long)D + ci * V, p);
#pragma unroll
            for (int j = 0; j < V; ++j) t[j] += p[j];
        }
        if (thresh) {
            const uint32_t km = dropout_mask<V>(seed, off, thresh);
#pragma unroll
            for (int j = 0; j < V; ++j) t[j] = ((km >> j) & 1u) ? t[j] * keep_scale : 0.f;
        }
        VecIO<T>::store(y + off, t);
    }
}

}  // namespace cwlt
extern "C" {

int cwlt_rowslab_blocks(int64_t rows) {
    int64_t b = (rows + 15) / 16;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

int cwlt_bias_gelu_dropout_fwd(const void* h, const float* bias, void* g, void* gd, int64_t rows, int F, float p,
                               uint64_t seed, const uint64_t* seed_base, int dtype, void* stream) {
    using namespace cwlt;
    if (!h || !g || g == h || gd == g || rows < 0 || F <= 0 || (F & 7) || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    const int vec = dtype == CWLT_BF16 ? 8 : 4;
    // forward has no per-slab partial sums to reduce afterwards: use finer slabs (more loads in flight)
    int64_t nslab = (rows + 7) / 8;
    if (nslab > 4096) nslab = 4096;
    const dim3 grid((F / vec + 255) / 256, (unsigned)nslab), block(256);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t th = drop_thresh(p);
    const float ks = drop_scale(p);
#define CWLT_GELU_FWD(T, GD)                                                                                      \
    hipLaunchKernelGGL((bias_gelu_dropout_fwd_kernel<T, GD>), grid, block, 0, st, (const T*)h, bias, (T*)g, (T*)gd,  \
                       (long)rows, F, th, ks, seed, seed_base)
    if (dtype == CWLT_F32) {
        if (gd) CWLT_GELU_FWD(float, true); else CWLT_GELU_FWD(float, false);
    } else if (dtype == CWLT_BF16) {
        if (gd) CWLT_GELU_FWD(bf16_t, true); else CWLT_GELU_FWD(bf16_t, false);
    } else {
        return CWLT_ERR_DTYPE;
    }
#undef CWLT_GELU_FWD
    return (int)hipGetLastError();
}

/* part: cwlt_rowslab_blocks(rows) * F floats (only if dbias != NULL); dbias (F) f32 = column sums of dh. */
int cwlt_bias_gelu_dropout_bwd(const void* dg, const void* h, const float* bias, void* dh, float* part, float* dbias,
                               int64_t rows, int F, float p, uint64_t seed, const uint64_t* seed_base, int dtype, void* stream) {
    using namespace cwlt;
    if (!dg || !h || !dh || rows < 0 || F <= 0 || (F & 7) || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (dbias && !part) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    const int nb = cwlt_rowslab_blocks(rows);
    const int vec = dtype == CWLT_BF16 ? 8 : 4;
    const dim3 grid((F / vec + 255) / 256, nb), block(256);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t th = drop_thresh(p);
    const float ks = drop_scale(p);
    float* pp = dbias ? part : nullptr;
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((bias_gelu_dropout_bwd_kernel<float>), grid, block, 0, st, (const float*)dg, (const float*)h,
                           bias, (float*)dh, pp, (long)rows, F, th, ks, seed, seed_base);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((bias_gelu_dropout_bwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dg,
                           (const bf16_t*)h, bias, (bf16_t*)dh, pp, (long)rows, F, th, ks, seed, seed_base);
    else
        return CWLT_ERR_DTYPE;
    int e = (int)hipGetLastError();
    if (e || !dbias) return e;
    return launch_colsum_finalize(part, dbias, nb, (long)F, F, 1.0f, 0, st);
}

/* y = dropout(x + pe[r % T]); pe may be NULL (plain dropout; also the backward of this op with dy as x). */
int cwlt_posenc_dropout(const void* x, const float* pe, void* y, int64_t rows, int T, int D, float p, uint64_t seed,
                        const uint64_t* seed_base, int dtype, void* stream) {
    using namespace cwlt;
    if (!x || !y || rows < 0 || D <= 0 || (D & 7) || T <= 0 || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    const long n4 = rows * (long)(D / (dtype == CWLT_BF16 ? 8 : 4));
    long nb = (n4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t th = drop_thresh(p);
    const float ks = drop_scale(p);
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((posenc_dropout_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)x, pe, (float*)y,
                           (long)rows, T, D, th, ks, seed, seed_base);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((posenc_dropout_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, pe,
                           (bf16_t*)y, (long)rows, T, D, th, ks, seed, seed_base);
    else
        return CWLT_ERR_DTYPE;
    return (int)hipGetLastError();
}

}  // extern "C"
