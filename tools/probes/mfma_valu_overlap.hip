// Does VALU work overlap MFMA work on gfx950, and in which arrangement?  (hipcc --offload-arch=gfx950 -O3, run on a GPU box)
// Every workgroup = 8 waves (2 per SIMD), `iters` steps; per step and wave: NM MFMAs (32x32x16 bf16, independent
// accumulators) and NV packed-f32 FMAs on private registers.  Modes:
//   0  MFMA only          1  VALU only          2  same wave: MFMAs then VALU            3  waves 0-3 MFMA-first, 4-7 VALU-first
//   4  waves 0-3 only MFMA, waves 4-7 only VALU (2x the per-wave amounts, same totals)   5..7 = 2..4 with s_barrier per step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NM, int NV>
__device__ __forceinline__ void mfmas(f32x16 (&acc)[4], bf16x8 a, bf16x8 b) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
}
template <int NV>
__device__ __forceinline__ void valus(f32x2 (&v)[8]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * 1.0000001f + 1e-9f;      // v_pk_fma_f32
}

template <int NM, int NV>
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters, int mode) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[4];
    f32x2 v[8];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) { v[i][0] = threadIdx.x * 1e-3f + i; v[i][1] = i; }
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x & 15) + j); b[j] = (__bf16)(0.002f * j); }
    const bool bar = mode >= 5;
    const int m = bar ? mode - 3 : mode;
    for (int it = 0; it < iters; ++it) {
        if (m == 0) mfmas<NM, NV>(acc, a, b);
        else if (m == 1) valus<NV>(v);
        else if (m == 2) { mfmas<NM, NV>(acc, a, b); __builtin_amdgcn_sched_barrier(0); valus<NV>(v); }
        else if (m == 3) {
            if (w < 4) { mfmas<NM, NV>(acc, a, b); __builtin_amdgcn_sched_barrier(0); valus<NV>(v); }
            else { valus<NV>(v); __builtin_amdgcn_sched_barrier(0); mfmas<NM, NV>(acc, a, b); }
        } else {
            if (w < 4) { mfmas<NM, NV>(acc, a, b); mfmas<NM, NV>(acc, a, b); }
            else { valus<NV>(v); valus<NV>(v); }
        }
        if (bar) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// MFMA-only rate against the number of independent accumulators a wave cycles through (waves per SIMD: 2)
template <int NACC>
__global__ __launch_bounds__(1024, 1) void mfma_rate(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x & 15) + j); b[j] = (__bf16)(0.002f * j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NACC], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NACC>
static void run_rate(float* out, int iters, int threads) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_rate<NACC>), dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double fl = 256.0 * (threads / 64) * 8.0 * iters * 32768.0;
    printf("MFMA only, %d accumulators, %d waves/CU: %.3f ms  %.0f TFLOP/s\n", NACC, threads / 64, best, fl / best / 1e9);
}

int main() {
    {
        float* o;
        hipMalloc(&o, 256 * 1024 * sizeof(float));
        run_rate<1>(o, 20000, 512); run_rate<2>(o, 20000, 512); run_rate<4>(o, 20000, 512); run_rate<8>(o, 20000, 512);
        run_rate<4>(o, 20000, 256); run_rate<8>(o, 20000, 256); run_rate<4>(o, 20000, 1024);
    }
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 8; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((probe<8, 140>), dim3(256), dim3(512), 0, 0, out, iters, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("mode %d: %.3f ms  (%.1f ns per step)\n", mode, best, best * 1e6 / iters);
    }
    return 0;
}
