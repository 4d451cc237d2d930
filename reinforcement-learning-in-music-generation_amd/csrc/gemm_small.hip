// The dense projections at FEW token rows (the reference's own RL setting: 30 windows x 50 tokens = 1 500 rows,
// /root/reference/dqn_policy/IRL_dqn_train.py:267-345, /root/reference/ppo_policy/ppo_train.py:365-417):
//     C (M, N) [+]= A (M, K) . W (N, K)^T [+ bias]
// gemm_bf16.hip's 256 x 256 persistent tiles leave 12 workgroups on 256 CUs at M = 1 500, N = 512; here a workgroup
// owns 64 x 64 of the output (192 .. 768 workgroups at those sizes), a wave 32 x 32 = four v_mfma_f32_16x16x32_bf16
// tiles, and the operands never touch LDS: both are K-contiguous, so the 16 bytes a lane feeds an MFMA are one buffer
// load (rows past the end read back as zeros: hardware range check).  Such a product is latency-bound -- 0.1-3 GFLOP,
// operands resident in L2 -- so the loop keeps PD = 8 k-steps of fragments in flight per wave (128 registers) and the
// compiler's counted waits retire them in order.  Same arithmetic as gemm_bf16.hip: f32 accumulation over K in
// k-steps of 32, bias added in f32, one rounding to bf16.
//
// Long reductions (K = 1536, 2048) are a chain of 48-64 dependent k-steps for one wave; where K % 128 == 0 the four waves
// of a workgroup instead split K four ways over ONE output tile and add their accumulators through LDS in a fixed order:
// a quarter of the chain.  Below 256 rows the tile is 32 x 32 (a 50-row rollout step still gets 32 workgroups at
// N = 512); from 256 rows it is 64 x 64, a wave holding 4 x 4 MFMA tiles: with nothing shared through LDS every wave pulls
// (rows + columns) x K x 2 bytes from L2, 144-192 MB per product at 1 500 rows with the small tile -- that, not latency,
// bounded it there (21-27 us whole-K, 20-25 us split-K on 32 x 32, 13-17 us on 64 x 64; profiles/r04_layer_call.txt).
//
// cwlt_transpose_bf16_many: the transposed weight copies the input-gradient products read (dX = dY . W is an NT
// product on W^T), all matrices of an encoder in one launch.
#include <stdlib.h>

#include "cwlt_common.h"
#include "cwlt_gelu.h"

namespace cwlt {
namespace gs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TM = 64, TN = 64, PD = 8;

template <bool BIAS, bool ACCUM>
__global__ __launch_bounds__(256) void gemm_small_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                         const float* __restrict__ bias, bf16_t* C, long M, int N, int K,
                                                         long lda, long ldw, long ldc) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const long m0 = (long)blockIdx.y * TM + 32 * (w >> 1);
    const int n0 = blockIdx.x * TN + 32 * (w & 1);
    if (m0 >= M || n0 >= N) return;                      // no LDS, no barrier: a wave without rows or columns leaves
    const long mr = min(32l, M - m0);
    const int nr = min(32, N - n0);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(A + m0 * lda, (uint32_t)(((mr - 1) * lda + K) * 2));
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(W + (long)n0 * ldw, (uint32_t)(((long)(nr - 1) * ldw + K) * 2));
    // MFMA 16x16x32: lane (l15, kg) feeds operand row l15, k = 8 kg .. 8 kg + 7 of a 32-wide k-step.  The W rows of the
    // two column tiles are dealt so that accumulator register r of tile nb is column 8 kg + 4 nb + r of the wave's 32:
    // a lane ends up with 8 consecutive columns of one row (one 16-byte store).
    uint32_t a_off[2], w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_off[i] = (uint32_t)(((16 * i + l15) * lda + 8 * kg) * 2);
        w_off[i] = (uint32_t)(((8 * (l15 >> 2) + 4 * i + (l15 & 3)) * ldw + 8 * kg) * 2);
    }
    bf16x8 fa[PD][2], fw[PD][2];
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nks = K >> 5;
#define GS_LOAD(slot, ks)                                                                                       \
    {                                                                                                           \
        const int so_ = (ks) * 64;                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                      \
            fa[slot][i_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ars, (int)a_off[i_], so_, 0)); \
            fw[slot][i_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)w_off[i_], so_, 0)); \
        }                                                                                                       \
    }
#define GS_MFMA(slot)                                                                                           \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 2; ++mb_)                                                         \
        _Pragma("unroll") for (int nb_ = 0; nb_ < 2; ++nb_)                                                     \
            acc[mb_][nb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[slot][nb_], fa[slot][mb_], acc[mb_][nb_], 0, 0, 0);
    // full groups of PD k-steps, branch-free inside (a branch between a load and its use makes hipcc drain the queue
    // at every loop head): group g's MFMAs run while group g + 1's loads are in flight; the last group loads nothing
    const int ngrp = nks / PD;
    if (ngrp > 0) {
#pragma unroll
        for (int s = 0; s < PD; ++s) GS_LOAD(s, s)
        for (int g = 1; g < ngrp; ++g) {
#pragma unroll
            for (int s = 0; s < PD; ++s) {
                GS_MFMA(s)
                GS_LOAD(s, g * PD + s)
            }
        }
#pragma unroll
        for (int s = 0; s < PD; ++s) GS_MFMA(s)
    }
    // what is left of K (none at K = 512, 1536, 2048)
    for (int ks = ngrp * PD; ks < nks; ++ks) {
        GS_LOAD(0, ks)
        GS_MFMA(0)
    }
#undef GS_MFMA
#undef GS_LOAD
    // lane (l15, kg): row 16 mb + l15, columns 8 kg .. 8 kg + 7 (registers 0-3 of tile 0, then of tile 1)
    const int col = 8 * kg;
    if (col >= nr) return;                               // N % 8 == 0: a lane's 8 columns are all inside or all outside
    float bs[8];
    if (BIAS) loadf<8>(bias + n0 + col, bs);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int row = 16 * mb + l15;
        if (row >= mr) continue;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = acc[mb][0][r];
            v[4 + r] = acc[mb][1][r];
        }
        bf16_t* cp = C + (m0 + row) * ldc + n0 + col;
        if (ACCUM) {
            float o[8];
            load8(cp, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += o[j];
        }
        if (BIAS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += bs[j];
        }
        store8(cp, v);
    }
}

// Split-K form: workgroup = one 32 x 32 output tile, wave w = k-steps [w nks / 4, (w + 1) nks / 4), nks / 4 a multiple of
// PD (all of a wave's groups are full).  Partial accumulators meet in LDS as [wave][component][lane] (lane-contiguous:
// conflict-free); wave w then owns MFMA tile (mb, nb) = (w >> 1, w & 1): row 16 mb + l15, columns 8 kg + 4 nb .. + 3.
// GELU != 0 (FFN forward at few rows: x = bf16(a w^T) + bias; c = dropout(gelu(x)), gd = mask / (1 - p) * gelu'(x), the
// arithmetic and dropout stream of cwlt_bias_gelu_dropout_fwd on the rounded product, element index row * N + column):
// BIAS, no ACCUM, dense c / gd with row stride N.
struct GeluArgs {
    bf16_t* gd;                // NULL: not wanted (no backward will follow)
    uint32_t thresh;
    float keep_scale;
    uint64_t seed;
    const uint64_t* seed_base;
};
template <bool BIAS, bool ACCUM, int PD, bool GELU = false>
__global__ __launch_bounds__(256) void gemm_small_splitk_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                const float* __restrict__ bias, bf16_t* C, long M, int N,
                                                                int K, long lda, long ldw, long ldc, GeluArgs ga) {
    __shared__ float red[4][16][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const long m0 = (long)blockIdx.y * 32;
    const int n0 = blockIdx.x * 32;
    const long mr = min(32l, M - m0);                    // >= 1: the grid covers exactly the tiles with rows and columns
    const int nr = min(32, N - n0);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(A + m0 * lda, (uint32_t)(((mr - 1) * lda + K) * 2));
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(W + (long)n0 * ldw, (uint32_t)(((long)(nr - 1) * ldw + K) * 2));
    uint32_t a_off[2], w_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_off[i] = (uint32_t)(((16 * i + l15) * lda + 8 * kg) * 2);
        w_off[i] = (uint32_t)(((8 * (l15 >> 2) + 4 * i + (l15 & 3)) * ldw + 8 * kg) * 2);
    }
    bf16x8 fa[PD][2], fw[PD][2];
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkw = K >> 7;                              // k-steps of this wave
    const int k0 = w * nkw;
#define GS_LOAD(slot, ks)                                                                                       \
    {                                                                                                           \
        const int so_ = (k0 + (ks)) * 64;                                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                      \
            fa[slot][i_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ars, (int)a_off[i_], so_, 0)); \
            fw[slot][i_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)w_off[i_], so_, 0)); \
        }                                                                                                       \
    }
#define GS_MFMA(slot)                                                                                           \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 2; ++mb_)                                                         \
        _Pragma("unroll") for (int nb_ = 0; nb_ < 2; ++nb_)                                                     \
            acc[mb_][nb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[slot][nb_], fa[slot][mb_], acc[mb_][nb_], 0, 0, 0);
    const int ngrp = nkw / PD;                           // >= 1 (launcher)
#pragma unroll
    for (int s = 0; s < PD; ++s) GS_LOAD(s, s)
    for (int g = 1; g < ngrp; ++g) {
#pragma unroll
        for (int s = 0; s < PD; ++s) {
            GS_MFMA(s)
            GS_LOAD(s, g * PD + s)
        }
    }
#pragma unroll
    for (int s = 0; s < PD; ++s) GS_MFMA(s)
#undef GS_MFMA
#undef GS_LOAD
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w][8 * mb + 4 * nb + r][lane] = acc[mb][nb][r];
    __syncthreads();
    const int mb = w >> 1, nb = w & 1;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 8 * mb + 4 * nb + r;
        v[r] = (red[0][c][lane] + red[1][c][lane]) + (red[2][c][lane] + red[3][c][lane]);
    }
    const int row = 16 * mb + l15, col = 8 * kg + 4 * nb;
    if (row >= mr || col >= nr) return;                  // N % 8 == 0: a lane's 4 columns are all inside or all outside
    bf16_t* cp = C + (m0 + row) * ldc + n0 + col;
    if (ACCUM) {
        const float4 o = load4(cp);
        v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
    }
    if (GELU) {
        const float4 b = load4(bias + n0 + col);
        uint64_t seed = ga.seed;
        if (ga.seed_base) seed += *ga.seed_base;
        const GeluK gk = gelu_consts(ga.keep_scale);
        const uint32_t thresh2 = ga.thresh | (ga.thresh << 16);
        const uint64_t pair = ((uint64_t)(m0 + row) * (uint64_t)N + (uint64_t)(n0 + col)) >> 1;
        const float bb[4] = {b.x, b.y, b.z, b.w};
        uint32_t r[2], dq[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x2 x, y, dy;
            x[0] = (float)(__bf16)v[2 * c] + bb[2 * c];
            x[1] = (float)(__bf16)v[2 * c + 1] + bb[2 * c + 1];
            gelu_scaled2(x, gk, y, dy);
            const uint32_t hw = keep_lanes16(rng_pair(seed, pair + c), thresh2);
            r[c] = f32x2_to_bf16x2(y[0], y[1]) & hw;
            dq[c] = f32x2_to_bf16x2(dy[0], dy[1]) & hw;
        }
        *reinterpret_cast<uint2*>(cp) = make_uint2(r[0], r[1]);
        if (ga.gd) *reinterpret_cast<uint2*>(ga.gd + (m0 + row) * ldc + n0 + col) = make_uint2(dq[0], dq[1]);
        return;
    }
    if (BIAS) {
        const float4 b = load4(bias + n0 + col);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    store4(cp, make_float4(v[0], v[1], v[2], v[3]));
}

// The same with a 64 x 64 output tile per workgroup (a wave: 4 x 4 MFMA tiles, 8 fragment loads per 16 MFMAs), for launches
// with a few hundred rows or more: nothing is shared through LDS, so the bytes pulled from L2 are (rows + columns) x K x 2
// per WAVE -- 144-192 MB per product at 1 500 rows with 32 x 32 tiles, which is what bounded them (13-25 us at ~8 TB/s);
// the larger tile halves that.  Wave w finishes row block w (rows 16 w + l15): per 32-column half, 8 consecutive columns.
template <bool BIAS, bool ACCUM, int PD>
__global__ __launch_bounds__(256) void gemm_small_splitk64_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                  const float* __restrict__ bias, bf16_t* C, long M, int N,
                                                                  int K, long lda, long ldw, long ldc) {
    __shared__ float red[4][64][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const long m0 = (long)blockIdx.y * 64;
    const int n0 = blockIdx.x * 64;
    const long mr = min(64l, M - m0);
    const int nr = min(64, N - n0);
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(A + m0 * lda, (uint32_t)(((mr - 1) * lda + K) * 2));
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(W + (long)n0 * ldw, (uint32_t)(((long)(nr - 1) * ldw + K) * 2));
    uint32_t a_off[4], w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_off[i] = (uint32_t)(((16 * i + l15) * lda + 8 * kg) * 2);
        w_off[i] = (uint32_t)(((32 * (i >> 1) + 8 * (l15 >> 2) + 4 * (i & 1) + (l15 & 3)) * ldw + 8 * kg) * 2);
    }
    bf16x8 fa[PD][4], fw[PD][4];
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkw = K >> 7;
    const int k0 = w * nkw;
#define GS_LOAD(slot, ks)                                                                                       \
    {                                                                                                           \
        const int so_ = (k0 + (ks)) * 64;                                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                      \
            fa[slot][i_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ars, (int)a_off[i_], so_, 0)); \
            fw[slot][i_] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)w_off[i_], so_, 0)); \
        }                                                                                                       \
    }
#define GS_MFMA(slot)                                                                                           \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_)                                                         \
        _Pragma("unroll") for (int nb_ = 0; nb_ < 4; ++nb_)                                                     \
            acc[mb_][nb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[slot][nb_], fa[slot][mb_], acc[mb_][nb_], 0, 0, 0);
    const int ngrp = nkw / PD;
#pragma unroll
    for (int s = 0; s < PD; ++s) GS_LOAD(s, s)
    for (int g = 1; g < ngrp; ++g) {
#pragma unroll
        for (int s = 0; s < PD; ++s) {
            GS_MFMA(s)
            GS_LOAD(s, g * PD + s)
        }
    }
#pragma unroll
    for (int s = 0; s < PD; ++s) GS_MFMA(s)
#undef GS_MFMA
#undef GS_LOAD
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w][16 * mb + 4 * nb + r][lane] = acc[mb][nb][r];
    __syncthreads();
    const int row = 16 * w + l15;
    if (row >= mr) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {                        // 32-column half: tiles nb = 2 q, 2 q + 1
        const int col = 32 * q + 8 * kg;
        if (col >= nr) continue;                         // N % 8 == 0
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 16 * w + 4 * (2 * q + (j >> 2)) + (j & 3);
            v[j] = (red[0][c][lane] + red[1][c][lane]) + (red[2][c][lane] + red[3][c][lane]);
        }
        bf16_t* cp = C + (m0 + row) * ldc + n0 + col;
        if (ACCUM) {
            float o[8];
            load8(cp, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += o[j];
        }
        if (BIAS) {
            float b[8];
            loadf<8>(bias + n0 + col, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += b[j];
        }
        store8(cp, v);
    }
}

// dst_i (cols_i, rows_i) = src_i (rows_i, cols_i)^T for every matrix of a table: int64 quadruples (src element offset,
// dst element offset, rows, cols) in device memory, offsets relative to `src` / `dst`.  64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_many_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             const int64_t* __restrict__ table) {
    __shared__ bf16_t tile[64][66];
    const int64_t* e = table + 4 * blockIdx.y;
    const long so = e[0], dof = e[1];
    const int rows = (int)e[2], cols = (int)e[3];
    const int tc = (cols + 63) >> 6, tr = (rows + 63) >> 6;
    for (int t = blockIdx.x; t < tc * tr; t += gridDim.x) {
        const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + ty + 4 * i, c = c0 + tx;
            tile[ty + 4 * i][tx] = (r < rows && c < cols) ? src[so + (long)r * cols + c] : (bf16_t)0;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + ty + 4 * i, r = r0 + tx;
            if (r < rows && c < cols) dst[dof + (long)c * rows + r] = tile[tx][ty + 4 * i];
        }
        __syncthreads();
    }
}

// dst_i[0 .. n_i) = bf16(src_i[0 .. n_i)) for every entry of a table: int64 triples (source element offset from `src`,
// destination element offset from `dst`, n) in device memory.  blockIdx.y = entry, the blocks of a row stride over it.
__global__ __launch_bounds__(256) void cast_many_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                        const int64_t* __restrict__ table) {
    const int64_t* e = table + 3 * blockIdx.y;
    const float* s = src + e[0];
    bf16_t* d = dst + e[1];
    const long n = e[2];
    const bool vec = ((((uintptr_t)s) & 15) == 0) && ((((uintptr_t)d) & 15) == 0);
    const long n8 = vec ? n / 8 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float x[8];
        load8(s + 8 * i, x);
        store8(d + 8 * i, x);
    }
    for (long i = 8 * n8 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = f32_to_bf16(s[i]);
}

}  // namespace gs
}  // namespace cwlt

extern "C" {

/* dst_i = bf16(src_i), element by element, for the n arrays of `table` (DEVICE memory, n x 3 int64: source element offset
 * from src (f32), destination element offset from dst (bf16), element count) in one launch. */
int cwlt_cast_bf16_many(const float* src, void* dst, const int64_t* table, int n, void* stream) {
    using namespace cwlt;
    if (n < 0 || n > 65535) return CWLT_ERR_ARG;
    if (n == 0) return CWLT_OK;
    if (!src || !dst || !table) return CWLT_ERR_ARG;
    hipLaunchKernelGGL(gs::cast_many_kernel, dim3(48, (unsigned)n), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst,
                       table);
    return (int)hipGetLastError();
}

/* c (M, N) [+]= a (M, K) . w (N, K)^T [+ bias (N) f32] for few rows (see include/cwlt.h): bf16 operands and result, f32
 * accumulation.  N % 8 == 0, K % 32 == 0, row strides multiples of 8 elements, 16-byte aligned pointers. */
int cwlt_gemm_bf16_small(const void* a, const void* w, const float* bias, void* c, int64_t M, int N, int K, int64_t lda,
                         int64_t ldw, int64_t ldc, int accumulate, void* stream) {
    using namespace cwlt;
    if (M < 0 || N <= 0 || K < 32 || (N % 8) || (K % 32)) return CWLT_ERR_ARG;
    if (M == 0) return CWLT_OK;
    if (!a || !w || !c) return CWLT_ERR_ARG;
    if (((lda | ldw | ldc) & 7) || lda < K || ldw < K || ldc < N) return CWLT_ERR_ARG;
    if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)c | (uintptr_t)bias) & 15) return CWLT_ERR_ARG;
    if (32 * lda * 2 >= (1ll << 31) || 32 * ldw * 2 >= (1ll << 31)) return CWLT_ERR_ARG;     // 32-bit offsets inside a wave tile
    typedef void (*kfn_t)(const bf16_t*, const bf16_t*, const float*, bf16_t*, long, int, int, long, long, long);
    typedef void (*kfng_t)(const bf16_t*, const bf16_t*, const float*, bf16_t*, long, int, int, long, long, long,
                           gs::GeluArgs);
    // split-K over the four waves where every wave gets whole groups of k-steps and the launch stays small (the form
    // re-reads each operand strip twice as often: meant for the few-thousand-row launches this file is for)
    static const int splitk = [] { const char* e = getenv("CWLT_GEMM_SMALL_SPLITK"); return e ? atoi(e) : 1; }();
    if (splitk && (K % 128) == 0 && M <= 16384 && (M + 31) / 32 <= 65535) {
        const int nkw = K >> 7;
        const int pd = (nkw % 8) == 0 ? 8 : (nkw % 4) == 0 ? 4 : (nkw % 2) == 0 ? 2 : 1;
        const int sel = (bias ? 2 : 0) | (accumulate ? 1 : 0);
        static const kfng_t tab[4][4] = {
            {gs::gemm_small_splitk_kernel<false, false, 1>, gs::gemm_small_splitk_kernel<false, false, 2>,
             gs::gemm_small_splitk_kernel<false, false, 4>, gs::gemm_small_splitk_kernel<false, false, 8>},
            {gs::gemm_small_splitk_kernel<false, true, 1>, gs::gemm_small_splitk_kernel<false, true, 2>,
             gs::gemm_small_splitk_kernel<false, true, 4>, gs::gemm_small_splitk_kernel<false, true, 8>},
            {gs::gemm_small_splitk_kernel<true, false, 1>, gs::gemm_small_splitk_kernel<true, false, 2>,
             gs::gemm_small_splitk_kernel<true, false, 4>, gs::gemm_small_splitk_kernel<true, false, 8>},
            {gs::gemm_small_splitk_kernel<true, true, 1>, gs::gemm_small_splitk_kernel<true, true, 2>,
             gs::gemm_small_splitk_kernel<true, true, 4>, gs::gemm_small_splitk_kernel<true, true, 8>}};
        // from a few hundred rows on: 64 x 64 tiles (half the L2 traffic); PD <= 4 there (8 fragments per k-step)
        static const int big_rows = [] { const char* e = getenv("CWLT_GEMM_SMALL_TILE64_ROWS"); return e ? atoi(e) : 256; }();
        if (M >= big_rows) {
            static const kfn_t tab64[4][3] = {
                {gs::gemm_small_splitk64_kernel<false, false, 1>, gs::gemm_small_splitk64_kernel<false, false, 2>,
                 gs::gemm_small_splitk64_kernel<false, false, 4>},
                {gs::gemm_small_splitk64_kernel<false, true, 1>, gs::gemm_small_splitk64_kernel<false, true, 2>,
                 gs::gemm_small_splitk64_kernel<false, true, 4>},
                {gs::gemm_small_splitk64_kernel<true, false, 1>, gs::gemm_small_splitk64_kernel<true, false, 2>,
                 gs::gemm_small_splitk64_kernel<true, false, 4>},
                {gs::gemm_small_splitk64_kernel<true, true, 1>, gs::gemm_small_splitk64_kernel<true, true, 2>,
                 gs::gemm_small_splitk64_kernel<true, true, 4>}};
            const kfn_t kf64 = tab64[sel][pd >= 4 ? 2 : pd == 2 ? 1 : 0];
            hipLaunchKernelGGL(kf64, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64)), dim3(256), 0,
                               (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)w, bias, (bf16_t*)c, (long)M, N, K,
                               (long)lda, (long)ldw, (long)ldc);
            return (int)hipGetLastError();
        }
        const kfng_t kf = tab[sel][pd == 8 ? 3 : pd == 4 ? 2 : pd == 2 ? 1 : 0];
        hipLaunchKernelGGL(kf, dim3((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)a, (const bf16_t*)w, bias, (bf16_t*)c, (long)M, N, K, (long)lda, (long)ldw,
                           (long)ldc, gs::GeluArgs{});
        return (int)hipGetLastError();
    }
    const long gy = (M + gs::TM - 1) / gs::TM;
    if (gy > 65535) return CWLT_ERR_ARG;                 // 4 M rows: far past where gemm_bf16.hip takes over
    const dim3 grid((unsigned)((N + gs::TN - 1) / gs::TN), (unsigned)gy);
    const kfn_t kfn = bias ? (accumulate ? gs::gemm_small_kernel<true, true> : gs::gemm_small_kernel<true, false>)
                           : (accumulate ? gs::gemm_small_kernel<false, true> : gs::gemm_small_kernel<false, false>);
    hipLaunchKernelGGL(kfn, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)w, bias, (bf16_t*)c,
                       (long)M, N, K, (long)lda, (long)ldw, (long)ldc);
    return (int)hipGetLastError();
}

/* FFN forward at few rows (see include/cwlt.h): g = dropout(gelu(bf16(a w^T) + bias)) and, when gd != NULL,
 * gd = mask / (1 - p) * gelu'(.), dense (M, N).  K % 128 == 0, N % 8 == 0, M <= 65535 * 32. */
int cwlt_gemm_bf16_small_gelu(const void* a, const void* w, const float* bias, void* g, void* gd, int64_t M, int N, int K,
                              int64_t lda, int64_t ldw, float p, uint64_t seed, const uint64_t* seed_base, void* stream) {
    using namespace cwlt;
    if (M < 0 || N <= 0 || K < 128 || (N % 8) || (K % 128) || !(p >= 0.f && p < 1.f)) return CWLT_ERR_ARG;
    if (M == 0) return CWLT_OK;
    if (!a || !w || !bias || !g) return CWLT_ERR_ARG;
    if (((lda | ldw) & 7) || lda < K || ldw < K) return CWLT_ERR_ARG;
    if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)g | (uintptr_t)gd | (uintptr_t)bias) & 15) return CWLT_ERR_ARG;
    if (32 * lda * 2 >= (1ll << 31) || 32 * ldw * 2 >= (1ll << 31) || (M + 31) / 32 > 65535) return CWLT_ERR_ARG;
    typedef void (*kfng_t)(const bf16_t*, const bf16_t*, const float*, bf16_t*, long, int, int, long, long, long,
                           gs::GeluArgs);
    const int nkw = K >> 7;
    const kfng_t kf = (nkw % 8) == 0   ? gs::gemm_small_splitk_kernel<true, false, 8, true>
                      : (nkw % 4) == 0 ? gs::gemm_small_splitk_kernel<true, false, 4, true>
                      : (nkw % 2) == 0 ? gs::gemm_small_splitk_kernel<true, false, 2, true>
                                       : gs::gemm_small_splitk_kernel<true, false, 1, true>;
    const gs::GeluArgs ga{(bf16_t*)gd, drop_thresh(p), drop_scale(p), seed, seed_base};
    hipLaunchKernelGGL(kf, dim3((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)a, (const bf16_t*)w, bias, (bf16_t*)g, (long)M, N, K, (long)lda, (long)ldw, (long)N, ga);
    return (int)hipGetLastError();
}

/* dst_i (cols_i, rows_i) = src_i (rows_i, cols_i)^T, bf16, for the n matrices of `table` (device memory, n x 4 int64:
 * source element offset from src, destination element offset from dst, rows, cols; dense matrices). */
int cwlt_transpose_bf16_many(const void* src, void* dst, const int64_t* table, int n, void* stream) {
    using namespace cwlt;
    if (n < 0 || n > 65535) return CWLT_ERR_ARG;
    if (n == 0) return CWLT_OK;
    if (!src || !dst || !table) return CWLT_ERR_ARG;
    hipLaunchKernelGGL(gs::transpose_many_kernel, dim3(64, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, table);
    return (int)hipGetLastError();
}

}  // extern "C"
