// Fused  s = x + dropout(a) ;  y = LayerNorm(s) * gamma + beta   (post-LN residual block), fwd + bwd.
//
// Replaces the PyTorch op chain inside fast_transformers' TransformerEncoderLayer.forward as the
// reference reaches it (dqn_policy/model.py:128-137,232):  x = x + dropout(attn) ; x = norm1(x) ;
// ... ; norm2(x + dropout(ffn)) ; and the encoder's final LayerNorm.  One pass over HBM instead of
// three (dropout, add, layer_norm), and in backward the column sums that make dgamma / dbeta / the
// preceding Linear's dbias are accumulated in registers while the rows stream through.
//
// Mapping: one wave per row; a lane keeps its 16-byte chunks (lane + 64*c) of the row in registers,
// so every load/store is a fully coalesced 1 KiB wave-wide access; row statistics by wave shuffles;
// f32 arithmetic whatever the storage dtype.  HBM-bound.
#include "cwlt_common.h"

namespace cwlt {

template <typename T, int NC>
__global__ __launch_bounds__(256) void add_dropout_ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ a,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, T* __restrict__ s_out,
                                                                 T* __restrict__ y, float* __restrict__ mean,
                                                                 float* __restrict__ rstd, long rows, int D, float eps,
                                                                 uint32_t thresh, float keep_scale, uint64_t seed,
        const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += *seed_base;   // device-resident offset: lets a captured hipGraph draw fresh masks per replay
    constexpr int V = VecIO<T>::N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = D / V;
    float gm[NC][V], bt[NC][V];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ci = lane + 64 * c;
        if (ci < nchunk) {
            loadf<V>(gamma + ci * V, gm[c]);
            loadf<V>(beta + ci * V, bt[c]);
        }
    }
    const float invD = 1.0f / (float)D;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        float v[NC][V];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ci = lane + 64 * c;
            if (ci < nchunk) {
                const long off = row * D + ci * V;
                VecIO<T>::load(a + off, v[c]);
                if (thresh) {
                    const uint32_t km = dropout_mask<V>(seed, off, thresh);
#pragma unroll
                    for (int j = 0; j < V; ++j) v[c][j] = ((km >> j) & 1u) ? v[c][j] * keep_scale : 0.f;
                }
                if (x) {
                    float r[V];
                    VecIO<T>::load(x + off, r);
#pragma unroll
                    for (int j = 0; j < V; ++j) v[c][j] += r[j];
                }
                if (s_out) VecIO<T>::store(s_out + off, v[c]);
#pragma unroll
                for (int j = 0; j < V; ++j) sum += v[c][j];
            }
        }
        const float mu = wave_sum(sum) * invD;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (lane + 64 * c < nchunk) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float d = v[c][j] - mu;
                    sq = fmaf(d, d, sq);
                }
            }
        }
        const float rs = rsqrtf(wave_sum(sq) * invD + eps);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ci = lane + 64 * c;
            if (ci < nchunk) {
                float o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) o[j] = (v[c][j] - mu) * rs * gm[c][j] + bt[c][j];
                VecIO<T>::store(y + row * D + ci * V, o);
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
                                                                 uint64_t seed,
        const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += *seed_base;   // device-resident offset: lets a captured hipGraph draw fresh masks per replay
    constexpr int V = VecIO<T>::N;
    __shared__ float red[4][NC * 64 * V];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = D / V;
    float gm[NC][V], ag[NC][V], ab[NC][V], ac[NC][V];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ci = lane + 64 * c;
#pragma unroll
        for (int j = 0; j < V; ++j) gm[c][j] = ag[c][j] = ab[c][j] = ac[c][j] = 0.f;
        if (ci < nchunk) loadf<V>(gamma + ci * V, gm[c]);
    }
    const float invD = 1.0f / (float)D;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float g[NC][V], xh[NC][V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ci = lane + 64 * c;
            if (ci < nchunk) {
                const long off = row * D + ci * V;
                VecIO<T>::load(dy + off, g[c]);
                if (dy2) {
                    float u[V];
                    VecIO<T>::load(dy2 + off, u);
#pragma unroll
                    for (int j = 0; j < V; ++j) g[c][j] += u[j];
                }
                VecIO<T>::load(s + off, xh[c]);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    xh[c][j] = (xh[c][j] - mu) * rs;
                    const float dx = g[c][j] * gm[c][j];
                    s1 += dx;
                    s2 = fmaf(dx, xh[c][j], s2);
                }
            }
        }
        const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ci = lane + 64 * c;
            if (ci < nchunk) {
                const long off = row * D + ci * V;
                float o[V], m[V];
                const uint32_t km = thresh ? dropout_mask<V>(seed, off, thresh) : 0xffffffffu;
                const float ksc = thresh ? keep_scale : 1.f;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    o[j] = rs * (g[c][j] * gm[c][j] - m1 - xh[c][j] * m2);
                    m[j] = ((km >> j) & 1u) ? o[j] * ksc : 0.f;
                    ag[c][j] = fmaf(g[c][j], xh[c][j], ag[c][j]);
                    ab[c][j] += g[c][j];
                    ac[c][j] += m[j];
                }
                if (ds) VecIO<T>::store(ds + off, o);
                if (da) VecIO<T>::store(da + off, m);
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
            float* r = &red[wave][(lane + 64 * c) * V];
#pragma unroll
            for (int j = 0; j < V; ++j) r[j] = qn == 0 ? ag[c][j] : (qn == 1 ? ab[c][j] : ac[c][j]);
        }
        __syncthreads();
        for (int col = threadIdx.x; col < D; col += 256)
            pb[qn * D + col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    }
}

// out[qi*out_stride + c] (+)= scale * sum_b part[b*stride + qi*ncols + c]; 1024 threads = 16 waves per
// 64 columns, each wave sums a slice of the partial rows, fixed-order combine (deterministic)
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               long out_stride, int nblocks, long stride, int ncols,
                                                               float scale, int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const float* p = part + (long)blockIdx.y * ncols;
    float acc = 0.f;
    if (col < ncols) {
        const int per = (nblocks + 15) / 16;
        const int b0 = wave * per, b1 = min(nblocks, b0 + per);
        // 16 independent loads in flight per thread: the walk is latency-bound (a few KB per column block)
        float a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = 0.f;
        int b = b0;
        for (; b + 15 < b1; b += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] += p[(long)(b + u) * stride + col];
        }
        for (; b < b1; ++b) a[0] += p[(long)b * stride + col];
        acc = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
              (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && col < ncols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][lane];
        float* o = out + (long)blockIdx.y * out_stride + col;
        *o = accumulate ? *o + scale * t : scale * t;
    }
}

// per-block column sums of a (rows, ncols) matrix with row stride ld: part[blockIdx.y][ncols]
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ part,
                                                             long rows, int ncols, long ld) {
    constexpr int V = VecIO<T>::N;
    const int ci = blockIdx.x * 256 + threadIdx.x;
    if (ci * V >= ncols) return;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
    float a0[V], a1[V];
#pragma unroll
    for (int j = 0; j < V; ++j) a0[j] = a1[j] = 0.f;
    long r = r0;
    for (; r + 1 < r1; r += 2) {
        float t0[V], t1[V];
        VecIO<T>::load(x + r * ld + ci * V, t0);
        VecIO<T>::load(x + (r + 1) * ld + ci * V, t1);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            a0[j] += t0[j];
            a1[j] += t1[j];
        }
    }
    if (r < r1) {
        float t0[V];
        VecIO<T>::load(x + r * ld + ci * V, t0);
#pragma unroll
        for (int j = 0; j < V; ++j) a0[j] += t0[j];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) part[(long)blockIdx.y * ncols + ci * V + j] = a0[j] + a1[j];
}

int launch_colsum_finalize(const float* part, float* out, int nblocks, long stride, int ncols, float scale,
                           int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 63) / 64, 1), dim3(1024), 0, st, part, out, 0L, nblocks,
                       stride, ncols, scale, accumulate);
    return (int)hipGetLastError();
}

int launch_colsum_finalize_multi(const float* part, float* out, long out_stride, int nq, int nblocks, long stride,
                                 int ncols, hipStream_t st) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((ncols + 63) / 64, nq), dim3(1024), 0, st, part, out, out_stride,
                       nblocks, stride, ncols, 1.0f, 0);
    return (int)hipGetLastError();
}

template <typename T>
static int ln_nc(int D) { return (D / VecIO<T>::N + 63) / 64; }

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
                                   uint64_t seed, const uint64_t* seed_base, int dtype, void* stream) {
    using namespace cwlt;
    if (!a || !gamma || !beta || !y || !mean || !rstd) return CWLT_ERR_ARG;
    if (rows < 0 || D <= 0 || (D & 7) || D > 1024 || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    const uint32_t th = drop_thresh(p);
    const float ks = drop_scale(p);
    const dim3 grid(cwlt_ln_blocks(rows)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define CWLT_LN_FWD(T, NC)                                                                                       \
    hipLaunchKernelGGL((add_dropout_ln_fwd_kernel<T, NC>), grid, block, 0, st, (const T*)x, (const T*)a, gamma, \
                       beta, (T*)s_out, (T*)y, mean, rstd, (long)rows, D, eps, th, ks, seed, seed_base)
    if (dtype == CWLT_F32) {
        const int nc = ln_nc<float>(D);
        if (nc == 1) CWLT_LN_FWD(float, 1); else if (nc == 2) CWLT_LN_FWD(float, 2); else CWLT_LN_FWD(float, 4);
    } else if (dtype == CWLT_BF16) {
        const int nc = ln_nc<bf16_t>(D);
        if (nc == 1) CWLT_LN_FWD(bf16_t, 1); else CWLT_LN_FWD(bf16_t, 2);
    } else {
        return CWLT_ERR_DTYPE;
    }
#undef CWLT_LN_FWD
    return (int)hipGetLastError();
}

/* part: f32 workspace of cwlt_ln_blocks(rows) * 3 * D floats; stats: (3, D) f32 output =
 * dgamma | dbeta | dbias (dbias = column sum of da = bias gradient of the Linear that produced `a`). */
int cwlt_add_dropout_layernorm_bwd(const void* dy, const void* dy2, const void* s, const float* gamma,
                                   const float* mean, const float* rstd, void* ds, void* da, float* part,
                                   float* stats, int64_t rows, int D, float p, uint64_t seed, const uint64_t* seed_base, int dtype,
                                   void* stream) {
    using namespace cwlt;
    if (!dy || !s || !gamma || !mean || !rstd || !part || !stats) return CWLT_ERR_ARG;
    if (rows < 0 || D <= 0 || (D & 7) || D > 1024 || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return (int)hipMemsetAsync(stats, 0, sizeof(float) * 3 * D, st);
    const uint32_t th = drop_thresh(p);
    const float ks = drop_scale(p);
    const int nb = cwlt_ln_blocks(rows);
    const dim3 grid(nb), block(256);
#define CWLT_LN_BWD(T, NC)                                                                                        \
    hipLaunchKernelGGL((add_dropout_ln_bwd_kernel<T, NC>), grid, block, 0, st, (const T*)dy, (const T*)dy2,      \
                       (const T*)s, gamma, mean, rstd, (T*)ds, (T*)da, part, (long)rows, D, th, ks, seed, seed_base)
    if (dtype == CWLT_F32) {
        const int nc = ln_nc<float>(D);
        if (nc == 1) CWLT_LN_BWD(float, 1); else if (nc == 2) CWLT_LN_BWD(float, 2); else CWLT_LN_BWD(float, 4);
    } else if (dtype == CWLT_BF16) {
        const int nc = ln_nc<bf16_t>(D);
        if (nc == 1) CWLT_LN_BWD(bf16_t, 1); else CWLT_LN_BWD(bf16_t, 2);
    } else {
        return CWLT_ERR_DTYPE;
    }
#undef CWLT_LN_BWD
    int e = (int)hipGetLastError();
    if (e) return e;
    return launch_colsum_finalize_multi(part, stats, (long)D, 3, nb, (long)3 * D, D, st);
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
    if (!x || !part || !out || rows < 0 || ncols <= 0 || (ncols & 7) || (ld & 7) || ld < ncols) return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return (int)hipMemsetAsync(out, 0, sizeof(float) * ncols, st);
    const int nb = cwlt_colsum_blocks(rows);
    const dim3 block(256);
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((colsum_partial_kernel<float>), dim3((ncols / 4 + 255) / 256, nb), block, 0, st,
                           (const float*)x, part, (long)rows, ncols, (long)ld);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), dim3((ncols / 8 + 255) / 256, nb), block, 0, st,
                           (const bf16_t*)x, part, (long)rows, ncols, (long)ld);
    else
        return CWLT_ERR_DTYPE;
    int e = (int)hipGetLastError();
    if (e) return e;
    return launch_colsum_finalize(part, out, nb, (long)ncols, ncols, 1.0f, 0, st);
}

}  // extern "C"
