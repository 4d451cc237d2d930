// bf16 MFMA tile helpers for gfx950 shared by the scan kernels (cla_bf16.hip) and the banded attention
// (band_attn.hip): 64x64 operand tiles live in LDS row-major, bf16 [64][72] (144-B rows: conflict-free 16-B
// row-type fragment reads); operands contracted over the row index are fetched transposed by
// ds_read_b64_tr_b16; v_mfma_f32_32x32x16_bf16 accumulators hold rows on registers, columns on lanes.
#pragma once
#include "cwlt_common.h"

namespace cwlt {
namespace b16 {

constexpr int D = 64;    // head dim
constexpr int C = 64;    // tokens per chunk
constexpr int LD = 72;   // bf16 tile row stride (144 B)
constexpr int LDO = 68;  // f32 output tile row stride (272 B)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ constexpr int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// row-type fragment X[l&31][k .. k+7] = T[row][k..k+7] of a row-major tile (k already includes 8*hf)
__device__ __forceinline__ bf16x8 row8(const bf16_t* t, int row, int k) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(t + row * LD + k));
}
// row-type fragment in the k order of an accumulator-held partner operand:
// k = k0 + 4*hf + {0..3} and k0 + 8 + 4*hf + {0..3}
__device__ __forceinline__ bf16x8 perm8(const bf16_t* t, int row, int k0, int hf) {
    const uint2 a = *reinterpret_cast<const uint2*>(t + row * LD + k0 + 4 * hf);
    const uint2 b = *reinterpret_cast<const uint2*>(t + row * LD + k0 + 8 + 4 * hf);
    return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}
// transposed fragment X[l&31][8h + j] = T[k0 + 8h + j][c0 + (l&31)] via ds_read_b64_tr_b16: lane 4q+p
// of a 16-lane group supplies &T[r0+q][cb+4p] and receives T[r0..r0+3][cb + lane%16]
// (semantics verified on gfx950 with tools/probes/tr16_probe.hip).
__device__ __forceinline__ bf16x8 tfrag8(const bf16_t* t, int k0, int c0, int lane) {
    const int q = (lane >> 2) & 3, p = lane & 3;
    const bf16_t* base = t + (k0 + 8 * (lane >> 5) + q) * LD + c0 + 16 * ((lane >> 4) & 1) + 4 * p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * LD));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
// Running states are f32 accumulators for the whole sequence; as MFMA operands they are rounded to bf16.
// CWLT_STATE_LO=1 additionally feeds the rounding residual (hi + lo, two bf16 MFMAs, ~16 mantissa bits).
// Measured against the f64 oracle (tools/scan_accuracy.py, L = 1024 and 4096) the residual changes nothing:
// rms errors 1.65e-4 / 3.1e-5 / 3.1e-5 / 2.3e-4 (out, dq, dk, dv) with it, 1.66e-4 / 3.3e-5 / 3.2e-5 / 2.3e-4
// without, identical maxima -- the error is set by the single bf16 roundings of the score tile and of the
// results.  Default 0: 4-6 % faster scans (fewer conversions and MFMAs).
#ifndef CWLT_STATE_LO
#define CWLT_STATE_LO 0
#endif
constexpr bool STATE_LO = CWLT_STATE_LO != 0;
__device__ __forceinline__ f32x16 mfma_hl(const bf16x8& hi, const bf16x8& lo, bf16x8 b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, b, acc, 0, 0, 0);
    if (STATE_LO) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, b, acc, 0, 0, 0);
    return acc;
}
// accumulator registers 8s..8s+7 as an operand fragment, split hi + lo
__device__ __forceinline__ void acc_frag(const f32x16& S, int s, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = S[8 * s + j];
        const __bf16 h = (__bf16)x;
        hi[j] = h;
        lo[j] = (__bf16)(x - (float)h);
    }
}
// fragment [x, 0, 0, 0, 0, 0, 0, 0] in lane-half 0, zeros in lane-half 1: element (k = 0) of an
// augmentation k-step when paired with acc_frag(., 0) (whose element 0 of lane-half 0 is row 0)
__device__ __forceinline__ bf16x8 first_if(bool c, float x) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f;
    v[0] = (__bf16)(c ? x : 0.0f);
    return v;
}
__device__ __forceinline__ bf16x8 ones_if(bool c) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(c ? 1.0f : 0.0f);
    return v;
}

// acc[x][y] += sum_k a_t[arow][k] * b_t[brow][k] over k-steps [s0, s1)
__device__ __forceinline__ f32x16 prod_rows(f32x16 acc, const bf16_t* a_t, int arow, const bf16_t* b_t, int brow,
                                            int s0, int s1, int hf) {
#pragma unroll 2
    for (int s = s0; s < s1; ++s)
        acc = mfma(row8(a_t, arow, 16 * s + 8 * hf), row8(b_t, brow, 16 * s + 8 * hf), acc);
    return acc;
}
// Z[n][col] += sum_k X_t[k][n] * b_t[brow][32t + k]: accumulator tiles X0, X1 (rows k on regs) as the
// A operand, hi + lo; b_t row-type in the permuted k order.  8 MFMAs.
__device__ __forceinline__ f32x16 prod_accA(f32x16 acc, const f32x16& X0, const f32x16& X1, const bf16_t* b_t,
                                            int brow, int hf) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x16& X = t == 0 ? X0 : X1;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 hi, lo;
            acc_frag(X, s, hi, lo);
            const bf16x8 b = perm8(b_t, brow, 32 * t + 16 * s, hf);
            acc = mfma_hl(hi, lo, b, acc);
        }
    }
    return acc;
}

__device__ __forceinline__ void unpack8(uint4 r, float (&x)[8]) {
    x[0] = __uint_as_float(r.x << 16); x[1] = __uint_as_float(r.x & 0xffff0000u);
    x[2] = __uint_as_float(r.y << 16); x[3] = __uint_as_float(r.y & 0xffff0000u);
    x[4] = __uint_as_float(r.z << 16); x[5] = __uint_as_float(r.z & 0xffff0000u);
    x[6] = __uint_as_float(r.w << 16); x[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&x)[8]) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)x[j];
    return __builtin_bit_cast(uint4, v);
}
__device__ __forceinline__ void put_row(bf16_t* t, int row, int col, uint4 v) {
    *reinterpret_cast<uint4*>(t + row * LD + col) = v;
}
// accumulator tile (rows rr on regs, cols on lanes) -> x[xrow = this lane's col][c0 + rr], bf16, 8 B at
// a time; element kept iff lo <= rr <= hi (rr = row inside the 32-tile), `add` added first
__device__ __forceinline__ void put_acc_T(bf16_t* x, int xrow, int c0, const f32x16& acc, int hf, int lo, int hi,
                                          float add) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 p;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = 8 * g + 4 * hf + u;  // = acc_row(4g+u, hf)
            p[u] = (__bf16)((rr >= lo && rr <= hi) ? acc[4 * g + u] + add : 0.f);
        }
        *reinterpret_cast<uint2*>(x + xrow * LD + c0 + 8 * g + 4 * hf) = __builtin_bit_cast(uint2, p);
    }
}
// same, f32 destination tile (16 B at a time)
__device__ __forceinline__ void put_acc_T_f32(float* x, int xrow, int c0, const f32x16& acc, int hf) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(x + xrow * LDO + c0 + 8 * g + 4 * hf) =
            make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
}

}  // namespace b16
}  // namespace cwlt
