// Fused  s = x + dropout(a) ;  y = LayerNorm(s) * gamma + beta   (post-LN residual block), fwd + bwd.
//
// Replaces the PyTorch op chain inside fast_transformers' TransformerEncoderLayer.forward as the
// reference reaches it (dqn_policy/model.py:128-137,232):  x = x + dropout(attn) ; x = norm1(x) ;
// ... ; norm2(x + dropout(ffn)) ; and the encoder's final LayerNorm.  One pass over HBM instead of
// three (dropout, add, layer_norm), and in backward the column sums that make dgamma / dbeta / the
// preceding Linear's dbias are accumulated in registers while the rows stream through.
//
// Mapping: one wave per row; a lane keeps its 4-element chunks (lane + 64*c) of the row in
// registers, so every load/store is a fully coalesced wave-wide access; row statistics by wave
// shuffles; f32 arithmetic whatever the storage dtype.  HBM-bound.
#include "cwlt_common.h"

namespace cwlt {

template <typename T, int NC>
__global__ __launch_bounds__(256) void add_dropout_ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ a,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, T* __restrict__ s_out,
                                                                 T* __restrict__ y, float* __restrict__ mean,
                                                                 float* __restrict__ rstd, long rows, int D, float eps,
                                                                 uint32_t thresh, float keep_scale, uint64_t seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nd4 = D >> 2;
    float4 gm[NC], bt[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int c4 = lane + 64 * c;
        gm[c] = c4 < nd4 ? load4(gamma + c4 * 4) : make_float4(0, 0, 0, 0);
        bt[c] = c4 < nd4 ? load4(beta + c4 * 4) : make_float4(0, 0, 0, 0);
    }
    const float invD = 1.0f / (float)D;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        float4 v[NC];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int c4 = lane + 64 * c;
            float4 t = make_float4(0, 0, 0, 0);
            if (c4 < nd4) {
                const long off = row * D + c4 * 4;
                t = load4(a + off);
                if (thresh) {
                    t.x = dropout_keep(seed, off + 0, thresh) ? t.x * keep_scale : 0.f;
                    t.y = dropout_keep(seed, off + 1, thresh) ? t.y * keep_scale : 0.f;
                    t.z = dropout_keep(seed, off + 2, thresh) ? t.z * keep_scale : 0.f;
                    t.w = dropout_keep(seed, off + 3, thresh) ? t.w * keep_scale : 0.f;
                }
                if (x) {
                    const float4 r = load4(x + off);
                    t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
                }
                if (s_out) store4(s_out + off, t);
            }
            v[c] = t;
            sum += (t.x + t.y) + (t.z + t.w);
        }
        const float mu = wave_sum(sum) * invD;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int c4 = lane + 64 * c;
            if (c4 < nd4) {
                const float dx = v[c].x - mu, dy = v[c].y - mu, dz = v[c].z - mu, dw = v[c].w - mu;
                sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        const float rs = rsqrtf(wave_sum(sq) * invD + eps);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int c4 = lane + 64 * c;
            if (c4 < nd4) {
                float4 o;
                o.x = (v[c].x - mu) * rs * gm[c].x + bt[c].x;
                o.y = (v[c].y - mu) * rs * gm[c].y + bt[c].y;
                o.z = (v[c].z - mu) * rs * gm[c].z + bt[c].z;
                o.w = (v[c].w - mu) * rs * gm[c].w + bt[c].w;
                store4(y + row * D + c4 * 4, o);
            }
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// part layout: [gridDim.x][3][D] = per-block column sums of (g*xhat, g, da)
template <typename T, int NC>
__global__ __launch_bounds__(256) void add_dropout_ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                                 const T* __restrict__ s,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, T* __restrict__ ds,
                                                                 T* __restrict__ da, float* __restrict__ part,
                                                                 long rows, int D, uint32_t thresh, float keep_scale,
                                                                 uint64_t seed) {
    __shared__ float red[4][NC * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nd4 = D >> 2;
    float4 gm[NC], ag[NC], ab[NC], ac[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int c4 = lane + 64 * c;
        gm[c] = c4 < nd4 ? load4(gamma + c4 * 4) : make_float4(0, 0, 0, 0);
        ag[c] = ab[c] = ac[c] = make_float4(0, 0, 0, 0);
    }
    const float invD = 1.0f / (float)D;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float4 g[NC], xh[NC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int c4 = lane + 64 * c;
            g[c] = xh[c] = make_float4(0, 0, 0, 0);
            if (c4 < nd4) {
                const long off = row * D + c4 * 4;
                float4 t = load4(dy + off);
                if (dy2) {
                    const float4 u = load4(dy2 + off);
                    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                }
                const float4 sv = load4(s + off);
                float4 h;
                h.x = (sv.x - mu) * rs; h.y = (sv.y - mu) * rs; h.z = (sv.z - mu) * rs; h.w = (sv.w - mu) * rs;
                g[c] = t;
                xh[c] = h;
                const float dx = t.x * gm[c].x, dyv = t.y * gm[c].y, dz = t.z * gm[c].z, dw = t.w * gm[c].w;
                s1 += (dx + dyv) + (dz + dw);
                s2 += (dx * h.x + dyv * h.y) + (dz * h.z + dw * h.w);
            }
        }
        const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int c4 = lane + 64 * c;
            if (c4 < nd4) {
                const long off = row * D + c4 * 4;
                float4 o;
                o.x = rs * (g[c].x * gm[c].x - m1 - xh[c].x * m2);
                o.y = rs * (g[c].y * gm[c].y - m1 - xh[c].y * m2);
                o.z = rs * (g[c].z * gm[c].z - m1 - xh[c].z * m2);
                o.w = rs * (g[c].w * gm[c].w - m1 - xh[c].w * m2);
                if (ds) store4(ds + off, o);
                float4 m = o;
                if (thresh) {
                    m.x = dropout_keep(seed, off + 0, thresh) ? o.x * keep_scale : 0.f;
                    m.y = dropout_keep(seed, off + 1, thresh) ? o.y * keep_scale : 0.f;
                    m.z = dropout_keep(seed, off + 2, thresh) ? o.z * keep_scale : 0.f;
                    m.w = dropout_keep(seed, off + 3, thresh) ? o.w * keep_scale : 0.f;
                }
                if (da) store4(da + off, m);
                ag[c].x += g[c].x * xh[c].x; ag[c].y += g[c].y * xh[c].y;
                ag[c].z += g[c].z * xh[c].z; ag[c].w += g[c].w * xh[c].w;
                ab[c].x += g[c].x; ab[c].y += g[c].y; ab[c].z += g[c].z; ab[c].w += g[c].w;
                ac[c].x += m.x; ac[c].y += m.y; ac[c].z += m.z; ac[c].w += m.w;
            }
        }
    }
    // block-level column sums: 4 waves -> one partial row per quantity
    float* pb = part + (long)blockIdx.x * 3 * D;
#pragma unroll
    for (int qn = 0; qn < 3; ++qn) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 t = qn == 0 ? ag[c] : (qn == 1 ? ab[c] : ac[c]);
            float* r = &red[wave][(lane + 64 * c) * 4];
            r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
        }
        __syncthreads();
        for (int col = threadIdx.x; col < D; col += 256)
            pb[qn * D + col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    }
}

// out[c] = scale * sum_b part[b*stride + c]   (deterministic fixed-order tree)
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int nblocks, long stride, int ncols, float scale,
                                                              int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (col < ncols) {
        const int per = (nblocks + 3) / 4;
        const int b0 = wave * per, b1 = min(nblocks, b0 + per);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int b = b0;
        for (; b + 3 < b1; b += 4) {
            a0 += part[(long)(b + 0) * stride + col];
            a1 += part[(long)(b + 1) * stride + col];
            a2 += part[(long)(b + 2) * stride + col];
            a3 += part[(long)(b + 3) * stride + col];
        }
        for (; b < b1; ++b) a0 += part[(long)b * stride + col];
        acc = (a0 + a1) + (a2 + a3);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && col < ncols) {
        const float v = scale * ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
        out[col] = accumulate ? out[col] + v : v;
    }
}

// per-block column sums of a (rows, ncols) matrix with row stride ld: part[blockIdx.y][ncols]
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ part,
                                                             long rows, int ncols, long ld) {
    const int c4 = blockIdx.x * 256 + threadIdx.x;
    if (c4 * 4 >= ncols) return;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
    float4 a0 = make_float4(0, 0, 0, 0), a1 = a0;
    long r = r0;
    for (; r + 1 < r1; r += 2) {
        const float4 t0 = load4(x + r * ld + c4 * 4);
        const float4 t1 = load4(x + (r + 1) * ld + c4 * 4);
        a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w;
        a1.x += t1.x; a1.y += t1.y; a1.z += t1.z; a1.w += t1.w;
    }
    if (r < r1) {
        const float4 t0 = load4(x + r * ld + c4 * 4);
        a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w;
    }
    store4(part + (long)blockIdx.y * ncols + c4 * 4, make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w));
}

int launch_colsum_finalize(const float* part, float* out, int nblocks, long stride, int ncols, float scale,
                           int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 63) / 64), dim3(256), 0, st, part, out, nblocks, stride,
                       ncols, scale, accumulate);
    return (int)hipGetLastError();
}

}  // namespace cwlt

extern "C" {

int cwlt_ln_blocks(int64_t rows) {
    int64_t b = (rows + 3) / 4;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

int cwlt_add_dropout_layernorm_fwd(const void* x, const void* a, const float* gamma, const float* beta, void* s_out,
                                   void* y, float* mean, float* rstd, int64_t rows, int D, float eps, float p,
                                   uint64_t seed, int dtype, void* stream) {
    using namespace cwlt;
    if (!a || !gamma || !beta || !y || !mean || !rstd) return CWLT_ERR_ARG;
    if (rows < 0 || D <= 0 || (D & 3) || D > 1024 || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    const uint32_t th = drop_thresh(p);
    const float ks = 1.0f / (1.0f - p);
    const dim3 grid(cwlt_ln_blocks(rows)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define CWLT_LN_FWD(T, NC)                                                                                       \
    hipLaunchKernelGGL((add_dropout_ln_fwd_kernel<T, NC>), grid, block, 0, st, (const T*)x, (const T*)a, gamma, \
                       beta, (T*)s_out, (T*)y, mean, rstd, (long)rows, D, eps, th, ks, seed)
    const int nc = (D + 255) / 256;
    if (dtype == CWLT_F32) {
        if (nc == 1) CWLT_LN_FWD(float, 1); else if (nc == 2) CWLT_LN_FWD(float, 2); else CWLT_LN_FWD(float, 4);
    } else if (dtype == CWLT_BF16) {
        if (nc == 1) CWLT_LN_FWD(bf16_t, 1); else if (nc == 2) CWLT_LN_FWD(bf16_t, 2); else CWLT_LN_FWD(bf16_t, 4);
    } else {
        return CWLT_ERR_DTYPE;
    }
#undef CWLT_LN_FWD
    return (int)hipGetLastError();
}

/* part: f32 workspace of cwlt_ln_blocks(rows) * 3 * D floats; dgamma/dbeta/dbias: (D) f32 outputs
 * (dbias = column sum of da, i.e. the bias gradient of the Linear that produced `a`; may be NULL). */
int cwlt_add_dropout_layernorm_bwd(const void* dy, const void* dy2, const void* s, const float* gamma,
                                   const float* mean, const float* rstd, void* ds, void* da, float* part,
                                   float* dgamma, float* dbeta, float* dbias, int64_t rows, int D, float p,
                                   uint64_t seed, int dtype, void* stream) {
    using namespace cwlt;
    if (!dy || !s || !gamma || !mean || !rstd || !part || !dgamma || !dbeta) return CWLT_ERR_ARG;
    if (rows < 0 || D <= 0 || (D & 3) || D > 1024 || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    const uint32_t th = drop_thresh(p);
    const float ks = 1.0f / (1.0f - p);
    const int nb = cwlt_ln_blocks(rows);
    const dim3 grid(nb), block(256);
    hipStream_t st = (hipStream_t)stream;
#define CWLT_LN_BWD(T, NC)                                                                                        \
    hipLaunchKernelGGL((add_dropout_ln_bwd_kernel<T, NC>), grid, block, 0, st, (const T*)dy, (const T*)dy2,      \
                       (const T*)s, gamma, mean, rstd, (T*)ds, (T*)da, part, (long)rows, D, th, ks, seed)
    const int nc = (D + 255) / 256;
    if (dtype == CWLT_F32) {
        if (nc == 1) CWLT_LN_BWD(float, 1); else if (nc == 2) CWLT_LN_BWD(float, 2); else CWLT_LN_BWD(float, 4);
    } else if (dtype == CWLT_BF16) {
        if (nc == 1) CWLT_LN_BWD(bf16_t, 1); else if (nc == 2) CWLT_LN_BWD(bf16_t, 2); else CWLT_LN_BWD(bf16_t, 4);
    } else {
        return CWLT_ERR_DTYPE;
    }
#undef CWLT_LN_BWD
    int e = (int)hipGetLastError();
    if (e) return e;
    const dim3 fg((D + 63) / 64);
    hipLaunchKernelGGL(colsum_finalize_kernel, fg, block, 0, st, part, dgamma, nb, (long)3 * D, D, 1.0f, 0);
    hipLaunchKernelGGL(colsum_finalize_kernel, fg, block, 0, st, part + D, dbeta, nb, (long)3 * D, D, 1.0f, 0);
    if (dbias)
        hipLaunchKernelGGL(colsum_finalize_kernel, fg, block, 0, st, part + 2 * D, dbias, nb, (long)3 * D, D, 1.0f, 0);
    return (int)hipGetLastError();
}

int cwlt_colsum_blocks(int64_t rows) {
    int64_t b = (rows + 63) / 64;
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return (int)b;
}

/* out[c] = sum_r x[r*ld + c], deterministic; part: cwlt_colsum_blocks(rows) * ncols floats. */
int cwlt_colsum(const void* x, float* part, float* out, int64_t rows, int ncols, int64_t ld, int dtype,
                void* stream) {
    using namespace cwlt;
    if (!x || !part || !out || rows < 0 || ncols <= 0 || (ncols & 3) || (ld & 3) || ld < ncols) return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return (int)hipMemsetAsync(out, 0, sizeof(float) * ncols, st);
    const int nb = cwlt_colsum_blocks(rows);
    const dim3 grid((ncols / 4 + 255) / 256, nb), block(256);
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, block, 0, st, (const float*)x, part, (long)rows, ncols,
                           (long)ld);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x, part, (long)rows,
                           ncols, (long)ld);
    else
        return CWLT_ERR_DTYPE;
    int e = (int)hipGetLastError();
    if (e) return e;
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 63) / 64), block, 0, st, part, out, nb, (long)ncols, ncols,
                       1.0f, 0);
    return (int)hipGetLastError();
}

}  // extern "C"
