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
// (read + write = 32 KiB per head per token: the step is pure HBM/L2 traffic, launch-latency bound --
// hence one memory round trip and one barrier, see the kernel).
#include "cwlt_common.h"

namespace cwlt {

__device__ __forceinline__ float phi(float x) { return x > 0.f ? x + 1.f : (expf(x) - 1.f) + 1.f; }

template <typename T>
__global__ __launch_bounds__(256) void recurrent_cla_step_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                 const T* __restrict__ v, float* __restrict__ S,
                                                                 float* __restrict__ Z, T* __restrict__ out, int H,
                                                                 long ldq, long ldk, long ldv, long ldo, float eps) {
    // Wave w owns state rows 16w .. 16w+15, lane m owns column m.  Every load a thread needs (q, k, v, Z and its 16
    // state values) is independent of every other and issued up front: one memory round trip, then registers.
    // phi(q)[d], phi(k)[d] for the wave's 16 rows come out of the wave's own lane-indexed copy with v_readlane
    // (d is wave-uniform) -- no LDS staging, ONE barrier (to add the four row-quarter partial sums).
    __shared__ float part[4][64];
    __shared__ float den_s;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x / H, h = blockIdx.x % H;
    float* Sb = S + ((long)n * H + h) * 4096 + (w * 16) * 64 + lane;
    float* Zb = Z + ((long)n * H + h) * 64;
    float s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = Sb[i * 64];
    const float qf = phi(load1(q + (long)n * ldq + h * 64 + lane));
    const float kf = phi(load1(k + (long)n * ldk + h * 64 + lane));
    const float vm = load1(v + (long)n * ldv + h * 64 + lane);
    if (w == 0) {                                   // normaliser: only wave 0 touches Z
        const float z = Zb[lane] + kf;
        Zb[lane] = z;
        const float d = wave_sum(qf * z);
        if (lane == 0) den_s = d;
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float kd = lane_value(kf, w * 16 + i), qd = lane_value(qf, w * 16 + i);
        s[i] = fmaf(kd, vm, s[i]);
        Sb[i * 64] = s[i];
        acc = fmaf(qd, s[i], acc);
    }
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        const float num = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        store1(out + (long)n * ldo + h * 64 + lane, num * (1.0f / (den_s + eps)));
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
