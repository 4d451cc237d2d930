// Shared device helpers for the cwlt (compound-word linear transformer) HIP kernels.
// gfx950 / CDNA4 only: wave = 64 lanes, f32 MFMA 32x32x2, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CWLT_OK 0
#define CWLT_ERR_ARG 1001      // bad shape / null pointer / unsupported size
#define CWLT_ERR_DTYPE 1002    // dtype code not supported by this entry point

// dtype codes of the C-ABI (include/cwlt.h)
#define CWLT_F32 0
#define CWLT_BF16 1

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;  // raw bfloat16 bits

namespace cwlt {

constexpr int WAVE = 64;

__device__ __forceinline__ float bf16_to_f32(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }

// round-to-nearest-even; NaN stays NaN (MI355X_MICROARCH.md "Correctness boundaries")
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// 4 consecutive elements -> float4 (16-B load for f32, 8-B load for bf16)
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    float4 f;
    f.x = __uint_as_float(r.x << 16);
    f.y = __uint_as_float(r.x & 0xffff0000u);
    f.z = __uint_as_float(r.y << 16);
    f.w = __uint_as_float(r.y & 0xffff0000u);
    return f;
}
__device__ __forceinline__ void store4(float* p, float4 f) { *reinterpret_cast<float4*>(p) = f; }
__device__ __forceinline__ void store4(bf16_t* p, float4 f) {
    uint2 r;
    r.x = (uint32_t)f32_to_bf16(f.x) | ((uint32_t)f32_to_bf16(f.y) << 16);
    r.y = (uint32_t)f32_to_bf16(f.z) | ((uint32_t)f32_to_bf16(f.w) << 16);
    *reinterpret_cast<uint2*>(p) = r;
}
__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void store1(float* p, float f) { *p = f; }
__device__ __forceinline__ void store1(bf16_t* p, float f) { *p = f32_to_bf16(f); }

// 8 consecutive elements (two float4) -- 32-B f32 / 16-B bf16
__device__ __forceinline__ void load8(const float* p, float (&x)[8]) {
    float4 a = load4(p), b = load4(p + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&x)[8]) {
    uint4 r = *reinterpret_cast<const uint4*>(p);
    x[0] = __uint_as_float(r.x << 16); x[1] = __uint_as_float(r.x & 0xffff0000u);
    x[2] = __uint_as_float(r.y << 16); x[3] = __uint_as_float(r.y & 0xffff0000u);
    x[4] = __uint_as_float(r.z << 16); x[5] = __uint_as_float(r.z & 0xffff0000u);
    x[6] = __uint_as_float(r.w << 16); x[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&x)[8]) {
    store4(p, make_float4(x[0], x[1], x[2], x[3]));
    store4(p + 4, make_float4(x[4], x[5], x[6], x[7]));
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&x)[8]) {
    uint4 r;
    r.x = (uint32_t)f32_to_bf16(x[0]) | ((uint32_t)f32_to_bf16(x[1]) << 16);
    r.y = (uint32_t)f32_to_bf16(x[2]) | ((uint32_t)f32_to_bf16(x[3]) << 16);
    r.z = (uint32_t)f32_to_bf16(x[4]) | ((uint32_t)f32_to_bf16(x[5]) << 16);
    r.w = (uint32_t)f32_to_bf16(x[6]) | ((uint32_t)f32_to_bf16(x[7]) << 16);
    *reinterpret_cast<uint4*>(p) = r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based RNG for dropout: one 32-bit hash per element index, keyed by (seed, stream).
// The mask is regenerated in backward from the same (seed, index) so no mask tensor is stored.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t idx) {
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    return hash32(lo ^ hash32(hi + 0x9e3779b9u + s1) ^ (s0 * 0x85ebca6bu + 0xc2b2ae35u));
}
// keep-probability test: keep iff u32 >= thresh, thresh = p * 2^32
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
    return rng_u32(seed, idx) >= thresh;
}

// ---- host-side helpers shared by the launchers ------------------------------------------------
static inline uint32_t drop_thresh(float p) {
    if (p <= 0.f) return 0u;
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}
// out[c] (+)= scale * sum_b part[b*stride + c], fixed summation order (defined in ln.hip)
int launch_colsum_finalize(const float* part, float* out, int nblocks, long stride, int ncols, float scale,
                           int accumulate, hipStream_t st);


// bf16 throughput path of the causal linear attention (cla_bf16.hip); row strides must be multiples of 8
int launch_cla_fwd_bf16(const void* q, const void* k, const void* v, void* out, float* zinv, int N, int H, int L,
                        long ldq, long ldk, long ldv, long ldo, float eps, hipStream_t st);
int launch_cla_bwd_dq_bf16(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                           const void* dout, void* dq, int N, int H, int L, long ldq, long ldk, long ldv, long ldo,
                           long lddo, long lddq, hipStream_t st);
int launch_cla_bwd_dkdv_bf16(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                             const void* dout, void* dk, void* dv, int N, int H, int L, long ldq, long ldk, long ldv,
                             long ldo, long lddo, long lddk, long lddv, hipStream_t st);

}  // namespace cwlt
