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

// Buffer (SRD) loads: the hardware range check returns zeros for offsets at or past `bytes`, so row guards cost no
// exec-mask branch, and addressing is one 32-bit byte offset per lane instead of a 64-bit pointer.  Build the
// descriptor from wave-uniform values only (kernel arguments, blockIdx) so it lives in SGPRs.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// zero 16-byte vector written as a literal at every use: a NAMED constant that stays live across a kernel was
// being parked in scratch memory by the register allocator and fetched back with scratch_load at each guarded load
#define CWLT_U4Z make_uint4(0u, 0u, 0u, 0u)

__device__ __forceinline__ float bf16_to_f32(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }

// round-to-nearest-even; NaN stays NaN (MI355X_MICROARCH.md "Correctness boundaries").  gfx950 has the
// conversion in hardware (v_cvt_pk_bf16_f32, two elements per instruction); the software form costs ~7 VALU
// instructions and an exec-mask branch per element, which made the HBM-bound kernels issue-bound.
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
    hwbf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
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
    r.x = f32x2_to_bf16x2(f.x, f.y);
    r.y = f32x2_to_bf16x2(f.z, f.w);
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
    r.x = f32x2_to_bf16x2(x[0], x[1]);
    r.y = f32x2_to_bf16x2(x[2], x[3]);
    r.z = f32x2_to_bf16x2(x[4], x[5]);
    r.w = f32x2_to_bf16x2(x[6], x[7]);
    *reinterpret_cast<uint4*>(p) = r;
}

// 16-byte vector access: VecIO<T>::N elements per lane (4 x f32 or 8 x bf16)
template <typename T> struct VecIO;
template <> struct VecIO<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&x)[4]) {
        const float4 a = load4(p);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&x)[4]) {
        store4(p, make_float4(x[0], x[1], x[2], x[3]));
    }
};
template <> struct VecIO<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&x)[8]) { load8(p, x); }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&x)[8]) { store8(p, x); }
};
// N consecutive f32 parameters (gamma / beta / bias)
template <int N>
__device__ __forceinline__ void loadf(const float* p, float (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; i += 4) {
        const float4 a = load4(p + i);
        x[i] = a.x; x[i + 1] = a.y; x[i + 2] = a.z; x[i + 3] = a.w;
    }
}

// Wave-wide all-reduce without the LDS pipe.  __shfl_xor compiles to ds_bpermute_b32 (an LDS-crossbar op
// with LDS latency and an s_waitcnt each); here the 16 lanes of a row combine through four DPP moves
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror -- after them every lane of a row holds the row's
// total) and the four row totals are read through SGPRs (v_readlane).  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);   // row_half_mirror
    v += dpp_move<0x140>(v);   // row_mirror
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// Counter-based RNG for dropout.  One 32-bit hash serves TWO consecutive elements (a 16-bit uniform each),
// keyed by (seed, element_index >> 1); an element is kept iff its 16 bits >= thresh16 = round(p * 65536).
// The mask is regenerated in backward from the same (seed, index), so no mask tensor is stored, and it
// does not depend on the vector width a kernel happens to use.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rng_pair(uint64_t seed, uint64_t pair_idx) {
    const uint32_t lo = (uint32_t)pair_idx, hi = (uint32_t)(pair_idx >> 32);
    const uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9e3779b9u) ^ (hi * 0x85ebca6bu);
    return hash32(lo * 0x9e3779b1u ^ key);
}
// keep flags of N consecutive elements starting at even element offset `off` (N even), bit j = keep
template <int N>
__device__ __forceinline__ uint32_t dropout_mask(uint64_t seed, uint64_t off, uint32_t thresh16) {
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < N; j += 2) {
        const uint32_t r = rng_pair(seed, (off >> 1) + (j >> 1));
        m |= ((r & 0xffffu) >= thresh16 ? 1u : 0u) << j;
        m |= ((r >> 16) >= thresh16 ? 1u : 0u) << (j + 1);
    }
    return m;
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh16) {
    const uint32_t r = rng_pair(seed, idx >> 1);
    return ((idx & 1) ? (r >> 16) : (r & 0xffffu)) >= thresh16;
}
// The keep flags of one hash word as two 16-bit lane masks (0xffff keep / 0 drop in the half of each element): what a
// kernel ANDs onto a packed pair of bf16 results.  thresh2 = thresh16 | thresh16 << 16 (0: keep everything).
// Three packed 16-bit instructions per hash word, written as asm: from the elementwise form hipcc builds two 16-bit
// compares, two v_cndmask_b32 and a v_perm_b32 (the v_cndmask alone measured 13-19 cycles per wave-instruction,
// tools/probes/valu_rates: profiles/r03_valu_rates.txt).
__device__ __forceinline__ uint32_t keep_lanes16(uint32_t r, uint32_t thresh2) {
    uint32_t m;
    const uint32_t one = 0x00010001u;
    asm("v_pk_sub_u16 %0, %1, %2 clamp\n\t"          // sat(thresh - r): 0 keep, >= 1 drop
        "v_pk_min_u16 %0, %0, %3\n\t"                // 0 keep, 1 drop
        "v_pk_sub_u16 %0, %0, %3"                      // 0xffff keep, 0 drop
        : "=&v"(m)
        : "s"(thresh2), "v"(r), "s"(one));
    return m;
}

// ---- host-side helpers shared by the launchers ------------------------------------------------
// 16-bit drop threshold and the matching (exact) keep scale 1 / (1 - thresh16 / 65536)
static inline uint32_t drop_thresh(float p) {
    if (p <= 0.f) return 0u;
    long t = (long)((double)p * 65536.0 + 0.5);
    if (t < 1) t = 1;
    if (t > 65535) t = 65535;
    return (uint32_t)t;
}
static inline float drop_scale(float p) {
    const uint32_t t = drop_thresh(p);
    return t ? (float)(65536.0 / (65536.0 - (double)t)) : 1.0f;
}
// out[c] (+)= scale * sum_b part[b*stride + c], fixed summation order (defined in ln.hip)
int launch_colsum_finalize(const float* part, float* out, int nblocks, long stride, int ncols, float scale,
                           int accumulate, hipStream_t st);
// nq quantities at once: out + qi*out_stride <- sum_b part[b*stride + qi*ncols + c]
int launch_colsum_finalize_multi(const float* part, float* out, long out_stride, int nq, int nblocks, long stride,
                                 int ncols, hipStream_t st);


// FFN forms of the 256 x 256 persistent projection GEMM (gemm_bf16.hip), behind gemm_nt.hip's entry points at training
// sizes.  gelu != 0: g (written) = dropout(gelu(a w^T + bias)) -> c, gd -> g;  gelu == 0: c = (a w^T) * g, part = column
// sums of c per 256-row tile (gemm_ffn_big_tiles(M) x N floats, or NULL).
int launch_gemm_ffn_big(int gelu, const void* a, const void* w, const float* bias, void* g, void* c, float* part,
                        int64_t M, int N, int K, int64_t lda, int64_t ldw, uint32_t thresh, float keep_scale,
                        uint64_t seed, const uint64_t* seed_base, hipStream_t st);
long gemm_ffn_big_tiles(long M);

// wgrad2.hip: the weight-gradient GEMM on gemm_bf16.hip's main loop (N1, N2 multiples of 256; mslice a multiple of 64, >= 256)
int launch_wgrad2(const void* a, const void* b, float* part, long M, int N1, int N2, long lda, long ldb, long mslice, int S,
                  hipStream_t st);

// bf16 throughput path of the causal linear attention (cla_bf16.hip); row strides must be multiples of 8.
// P = segments per stream (scan_segments: 1 unless N * H is far below the CU count), ws = scan_seg_floats(...) floats.
int scan_segments(int N, int H, int L);
long scan_seg_floats(int N, int H, int P, int backward);
// fin (optional, P == 1): scan_final_state_floats(N, H) floats, the state after the last token -- what the one-sweep
// backward (launch_cla_bwd_sweep_bf16) starts its dQ scan from.
long scan_final_state_floats(int N, int H);
int launch_cla_fwd_bf16(const void* q, const void* k, const void* v, void* out, float* zinv, int N, int H, int L,
                        long ldq, long ldk, long ldv, long ldo, float eps, int P, float* ws, float* fin,
                        hipStream_t st);
int launch_cla_bwd_sweep_bf16(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                              const void* dout, const float* fin, void* dq, void* dk, void* dv, float* csum_q,
                              float* csum_k, float* csum_v, int N, int H, int L, long ldq, long ldk, long ldv, long ldo,
                              long lddo, long lddq, long lddk, long lddv, hipStream_t st);
int launch_cla_bwd_dq_bf16(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                           const void* dout, const float* dden, void* dq, float* csum, int N, int H, int L, long ldq,
                           long ldk, long ldv, long ldo, long lddo, long lddq, int P, float* ws, hipStream_t st);
int launch_cla_bwd_dkdv_bf16(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                             const void* dout, void* dk, void* dv, float* csum_k, float* csum_v, float* dden_out,
                             int N, int H, int L, long ldq, long ldk, long ldv, long ldo, long lddo, long lddk,
                             long lddv, int P, float* ws, hipStream_t st);

}  // namespace cwlt
