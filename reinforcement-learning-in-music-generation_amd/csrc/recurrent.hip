// Recurrent (one token per call) causal linear attention step -- the generation-time form.
//
// Replaces fast_transformers RecurrentLinearAttention.forward (recurrent/attention/self_attention/
// linear_attention.py of pytorch-fast-transformers 0.4.0) reached through RecurrentEncoderBuilder at
// /root/reference/dqn_policy/model.py:141-150,236-238 and dqn_policy/testing-no-type-cp.py:126-179:
//     Q = elu(q)+1 ; K = elu(k)+1 ; Zi += K ; Si += K (x) v
//     out = (Q . Si) / (Q . Zi + eps)
// q, k, v: (N, H, 64) rows of the per-token projections (row stride ld*); S: (N, H, 64, 64) f32 and
// Z: (N, H, 64) f32 are updated IN PLACE.  One workgroup (256 threads) per (n, h): thread (d-quarter, m)
// owns 16 rows of one state column, so the 16 KiB state streams through registers exactly once
// (read + write = 32 KiB per head per token: the step is pure HBM/L2 traffic, launch-latency bound).
#include "cwlt_common.h"

namespace cwlt {

template <typename T>
__global__ __launch_bounds__(256) void recurrent_cla_step_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                 const T* __restrict__ v, float* __restrict__ S,
                                                                 float* __restrict__ Z, T* __restrict__ out, int H,
                                                                 long ldq, long ldk, long ldv, long ldo, float eps) {
    __shared__ float qf[64], kf[64], vv[64], part[4][64], den_s;
    const int tid = threadIdx.x;
    const int n = blockIdx.x / H, h = blockIdx.x % H;
    if (tid < 64) {
        const float x = load1(q + (long)n * ldq + h * 64 + tid);
        qf[tid] = x > 0.f ? x + 1.f : (expf(x) - 1.f) + 1.f;
    } else if (tid < 128) {
        const int d = tid - 64;
        const float x = load1(k + (long)n * ldk + h * 64 + d);
        kf[d] = x > 0.f ? x + 1.f : (expf(x) - 1.f) + 1.f;
    } else if (tid < 192) {
        vv[tid - 128] = load1(v + (long)n * ldv + h * 64 + (tid - 128));
    }
    __syncthreads();
    float* Zb = Z + ((long)n * H + h) * 64;
    if (tid < 64) {
        const float z = Zb[tid] + kf[tid];
        Zb[tid] = z;
        float d = qf[tid] * z;
        d = wave_sum(d);
        if (tid == 0) den_s = d;
    }
    // state column m, rows d0 .. d0+15
    const int m = tid & 63, d0 = (tid >> 6) * 16;
    float* Sb = S + ((long)n * H + h) * 4096;
    float acc = 0.f;
    const float vm = vv[m];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int d = d0 + i;
        const float s = fmaf(kf[d], vm, Sb[d * 64 + m]);
        Sb[d * 64 + m] = s;
        acc = fmaf(qf[d], s, acc);
    }
    part[tid >> 6][m] = acc;
    __syncthreads();
    if (tid < 64) {
        const float num = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        store1(out + (long)n * ldo + h * 64 + tid, num * (1.0f / (den_s + eps)));
    }
}

}  // namespace cwlt

extern "C" {

int cwlt_recurrent_cla_step(const void* q, const void* k, const void* v, float* S, float* Z, void* out, int N, int H,
                            int head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float eps, int dtype,
                            void* stream) {
    using namespace cwlt;
    if (!q || !k || !v || !S || !Z || !out || N < 0 || H <= 0 || head_dim != 64) return CWLT_ERR_ARG;
    if (N == 0) return CWLT_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((recurrent_cla_step_kernel<float>), dim3(N * H), dim3(256), 0, st, (const float*)q,
                           (const float*)k, (const float*)v, S, Z, (float*)out, H, (long)ldq, (long)ldk, (long)ldv,
                           (long)ldo, eps);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((recurrent_cla_step_kernel<bf16_t>), dim3(N * H), dim3(256), 0, st, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, S, Z, (bf16_t*)out, H, (long)ldq, (long)ldk, (long)ldv,
                           (long)ldo, eps);
    else
        return CWLT_ERR_DTYPE;
    return (int)hipGetLastError();
}

}  // extern "C"
