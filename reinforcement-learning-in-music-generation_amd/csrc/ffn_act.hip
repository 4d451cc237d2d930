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

namespace cwlt {

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

template <typename T>
__global__ __launch_bounds__(256) void bias_gelu_dropout_fwd_kernel(const T* __restrict__ h,
                                                                    const float* __restrict__ bias, T* __restrict__ g,
                                                                    long rows, int F, uint32_t thresh, float keep_scale,
                                                                    uint64_t seed) {
    const int c4 = blockIdx.x * 256 + threadIdx.x;
    if (c4 * 4 >= F) return;
    const float4 b = bias ? load4(bias + c4 * 4) : make_float4(0, 0, 0, 0);
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
#pragma unroll 4
    for (long r = r0; r < r1; ++r) {
        const long off = r * F + c4 * 4;
        const float4 t = load4(h + off);
        float4 o;
        o.x = gelu_f(t.x + b.x); o.y = gelu_f(t.y + b.y); o.z = gelu_f(t.z + b.z); o.w = gelu_f(t.w + b.w);
        if (thresh) {
            o.x = dropout_keep(seed, off + 0, thresh) ? o.x * keep_scale : 0.f;
            o.y = dropout_keep(seed, off + 1, thresh) ? o.y * keep_scale : 0.f;
            o.z = dropout_keep(seed, off + 2, thresh) ? o.z * keep_scale : 0.f;
            o.w = dropout_keep(seed, off + 3, thresh) ? o.w * keep_scale : 0.f;
        }
        store4(g + off, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bias_gelu_dropout_bwd_kernel(const T* __restrict__ dg, const T* __restrict__ h,
                                                                    const float* __restrict__ bias, T* __restrict__ dh,
                                                                    float* __restrict__ part, long rows, int F,
                                                                    uint32_t thresh, float keep_scale, uint64_t seed) {
    const int c4 = blockIdx.x * 256 + threadIdx.x;
    if (c4 * 4 >= F) return;
    const float4 b = bias ? load4(bias + c4 * 4) : make_float4(0, 0, 0, 0);
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll 4
    for (long r = r0; r < r1; ++r) {
        const long off = r * F + c4 * 4;
        float4 d = load4(dg + off);
        const float4 t = load4(h + off);
        if (thresh) {
            d.x = dropout_keep(seed, off + 0, thresh) ? d.x * keep_scale : 0.f;
            d.y = dropout_keep(seed, off + 1, thresh) ? d.y * keep_scale : 0.f;
            d.z = dropout_keep(seed, off + 2, thresh) ? d.z * keep_scale : 0.f;
            d.w = dropout_keep(seed, off + 3, thresh) ? d.w * keep_scale : 0.f;
        }
        float4 o;
        o.x = d.x * gelu_grad_f(t.x + b.x); o.y = d.y * gelu_grad_f(t.y + b.y);
        o.z = d.z * gelu_grad_f(t.z + b.z); o.w = d.w * gelu_grad_f(t.w + b.w);
        store4(dh + off, o);
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    if (part) store4(part + (long)blockIdx.y * F + c4 * 4, acc);
}

// y = dropout(x + pe[t]) with pe (max_len, D) f32, row r of x is position r % T   (model.py:90-92)
template <typename T>
__global__ __launch_bounds__(256) void posenc_dropout_kernel(const T* __restrict__ x, const float* __restrict__ pe,
                                                             T* __restrict__ y, long rows, int Tlen, int D,
                                                             uint32_t thresh, float keep_scale, uint64_t seed) {
    const long n4 = rows * (long)(D >> 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long r = i / (D >> 2);
        const int c4 = (int)(i - r * (D >> 2));
        const long off = r * D + c4 * 4;
        float4 t = load4(x + off);
        if (pe) {
            const float4 p = load4(pe + (r % Tlen) * (long)D + c4 * 4);
            t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
        }
        if (thresh) {
            t.x = dropout_keep(seed, off + 0, thresh) ? t.x * keep_scale : 0.f;
            t.y = dropout_keep(seed, off + 1, thresh) ? t.y * keep_scale : 0.f;
            t.z = dropout_keep(seed, off + 2, thresh) ? t.z * keep_scale : 0.f;
            t.w = dropout_keep(seed, off + 3, thresh) ? t.w * keep_scale : 0.f;
        }
        store4(y + off, t);
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

int cwlt_bias_gelu_dropout_fwd(const void* h, const float* bias, void* g, int64_t rows, int F, float p, uint64_t seed,
                               int dtype, void* stream) {
    using namespace cwlt;
    if (!h || !g || rows < 0 || F <= 0 || (F & 3) || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    const dim3 grid((F / 4 + 255) / 256, cwlt_rowslab_blocks(rows)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t th = drop_thresh(p);
    const float ks = 1.0f / (1.0f - p);
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((bias_gelu_dropout_fwd_kernel<float>), grid, block, 0, st, (const float*)h, bias, (float*)g,
                           (long)rows, F, th, ks, seed);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((bias_gelu_dropout_fwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)h, bias,
                           (bf16_t*)g, (long)rows, F, th, ks, seed);
    else
        return CWLT_ERR_DTYPE;
    return (int)hipGetLastError();
}

/* part: cwlt_rowslab_blocks(rows) * F floats (only if dbias != NULL); dbias (F) f32 = column sums of dh. */
int cwlt_bias_gelu_dropout_bwd(const void* dg, const void* h, const float* bias, void* dh, float* part, float* dbias,
                               int64_t rows, int F, float p, uint64_t seed, int dtype, void* stream) {
    using namespace cwlt;
    if (!dg || !h || !dh || rows < 0 || F <= 0 || (F & 3) || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (dbias && !part) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    const int nb = cwlt_rowslab_blocks(rows);
    const dim3 grid((F / 4 + 255) / 256, nb), block(256);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t th = drop_thresh(p);
    const float ks = 1.0f / (1.0f - p);
    float* pp = dbias ? part : nullptr;
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((bias_gelu_dropout_bwd_kernel<float>), grid, block, 0, st, (const float*)dg, (const float*)h,
                           bias, (float*)dh, pp, (long)rows, F, th, ks, seed);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((bias_gelu_dropout_bwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)dg,
                           (const bf16_t*)h, bias, (bf16_t*)dh, pp, (long)rows, F, th, ks, seed);
    else
        return CWLT_ERR_DTYPE;
    int e = (int)hipGetLastError();
    if (e || !dbias) return e;
    return launch_colsum_finalize(part, dbias, nb, (long)F, F, 1.0f, 0, st);
}

/* y = dropout(x + pe[r % T]); pe may be NULL (plain dropout; also the backward of this op with dy as x). */
int cwlt_posenc_dropout(const void* x, const float* pe, void* y, int64_t rows, int T, int D, float p, uint64_t seed,
                        int dtype, void* stream) {
    using namespace cwlt;
    if (!x || !y || rows < 0 || D <= 0 || (D & 3) || T <= 0 || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    const long n4 = rows * (long)(D / 4);
    long nb = (n4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t th = drop_thresh(p);
    const float ks = 1.0f / (1.0f - p);
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((posenc_dropout_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)x, pe, (float*)y,
                           (long)rows, T, D, th, ks, seed);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((posenc_dropout_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)x, pe,
                           (bf16_t*)y, (long)rows, T, D, th, ks, seed);
    else
        return CWLT_ERR_DTYPE;
    return (int)hipGetLastError();
}

}  // extern "C"
