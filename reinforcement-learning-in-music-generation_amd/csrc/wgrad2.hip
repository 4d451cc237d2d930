// Weight-gradient GEMM, second form:  C[N1][N2] (f32 partials) = sum_m A[m][n1] * B[m][n2]  on the main loop of
// gemm_bf16.hip -- 8 waves as 2 x 4 (128 x 64 each), v_mfma_f32_16x16x32_bf16, half-tile LDS-DMA ring, waves 4-7 one
// barrier behind waves 0-3 so that every SIMD has a computing wave beside a loading one.  Same job as wgrad.hip (the
// K = B*T "TN" GEMMs autograd runs for every nn.Linear weight of the encoder, /root/reference/dqn_policy/model.py:128-137;
// same split of the token rows over workgroups, same f32 partial tiles, same fixed-order reduce), taken when both widths
// are multiples of 256 and a slice holds at least 4 K-tiles; wgrad.hip keeps the edge-tile shapes and short slices.
//
// What differs from the NT projection kernel: the reduction runs over the ROW index of both operands, so
//   * a half-tile is 64 token rows x 128 columns (256-byte LDS rows; A: the two 64-column runs the two wave rows need, B:
//     the four 32-column runs the four wave columns need), a DMA piece is 4 such rows;
//   * fragments are fetched transposed, two ds_read_b64_tr_b16 per 8-k fragment (4 token rows x 16 columns each, lane
//     4q + p of a 16-lane group supplying row q, columns 4p .. 4p + 3);
//   * 16-byte chunk c of LDS row r is stored at chunk position c ^ 2 ((r & 3) | ((r >> 3) & 1) << 2) (source side): the
//     32 eight-byte accesses of a half-wave (rows 8g + q of two 8-row groups, 4 x 8 bytes per row) then fall into 32
//     different bank pairs (tools/probes/swizzle_search.py, model of the tr read: zero conflicts).
// The product is taken with B's columns on the MFMA's row index, so a lane holds 4 consecutive n2 of one n1: 16-byte f32
// stores into the partial tile.
#include "cwlt_common.h"
#include <stdlib.h>

namespace cwlt {
namespace w2 {

constexpr int TM = 256, TN = 256, BK = 64;
constexpr int HALF = BK * 256;                  // bytes of a half-tile slot: 64 rows x 256 B = 16 KiB
constexpr int D = 5;                            // half-tiles a DMA piece is issued ahead of its use

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;

#define W2_DMA2(v0, v1, rs, la, so)                                                                    \
    {                                                                                                  \
        unsigned keep;                                                                                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\t"                          \
                     "buffer_load_dwordx4 %1, %3, %5 offen lds\n\t"                                    \
                     "s_add_u32 m0, %4, 0x400\n\ts_nop 0\n\t"                                         \
                     "buffer_load_dwordx4 %2, %3, %5 offen lds\n\t"                                    \
                     "s_mov_b32 m0, %0"                                                                \
                     : "=&s"(keep)                                                                     \
                     : "v"(v0), "v"(v1), "s"(rs), "s"(la), "s"(so)                                     \
                     : "memory", "scc");                                                               \
    }
#define W2_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// 8 k values (token rows 8g .. 8g + 7 of the k-step) of one column per lane: two transposed reads 4 rows apart
__device__ __forceinline__ bf16x8 tfrag(const char* base) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * 256));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(512, 2) void wgrad2_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         float* __restrict__ part, long M, int N1, int N2, long lda,
                                                         long ldb, long mslice) {
    __shared__ __attribute__((aligned(1024))) char lds[8 * HALF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;                 // wave tile: n1 128 wm .., n2 64 wn ..
    const int l15 = lane & 15, kg = lane >> 4;
    // workgroup -> (token slice, output tile): all tiles of one slice on ONE XCD (their re-reads of the slice hit its L2)
    const int nt2 = N2 / TN;
    const int ntile = (N1 / TM) * nt2;
    const int id = blockIdx.x;
    const int S = gridDim.x / ntile;
    int slice, tile;
    if (S & 7) {
        slice = id / ntile;
        tile = id - slice * ntile;
    } else {
        const int xcd = id & 7, idx = id >> 3;
        slice = xcd + 8 * (idx / ntile);
        tile = idx % ntile;
    }
    const int t1 = tile / nt2, t2 = tile % nt2;
    const long m0 = (long)slice * mslice;
    const long m1 = min(M, m0 + mslice);
    const long nrow = m1 > m0 ? m1 - m0 : 0;
    const int nK = (int)(mslice / BK);                 // >= 4 (launcher); K-tiles past the slice end read zeros

    // the slice's rows of both operands as buffer resources (rows past the end read back as zeros)
    const uint64_t abase = (uint64_t)(A + m0 * lda + (long)t1 * TM), bbase = (uint64_t)(B + m0 * ldb + (long)t2 * TN);
    u32x4_t ars, brs;
    ars[0] = __builtin_amdgcn_readfirstlane((uint32_t)abase);
    ars[1] = __builtin_amdgcn_readfirstlane((uint32_t)(abase >> 32));
    ars[2] = __builtin_amdgcn_readfirstlane(nrow ? (uint32_t)(((nrow - 1) * lda + TM) * 2) : 0u);
    ars[3] = 0x00020000u;
    brs[0] = __builtin_amdgcn_readfirstlane((uint32_t)bbase);
    brs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(bbase >> 32));
    brs[2] = __builtin_amdgcn_readfirstlane(nrow ? (uint32_t)(((nrow - 1) * ldb + TN) * 2) : 0u);
    brs[3] = 0x00020000u;

    // ---- DMA side.  This wave's two pieces of a slot are LDS rows 8 w + 4 i + (lane >> 4), i = 0, 1; lane l lands at chunk
    // position l & 15, which holds chunk (l & 15) ^ f(row), f = 2 ((row & 3) | ((row >> 3) & 1) << 2) -- the same for both
    // pieces.  Chunk c of an A half h is columns 128 (c >> 3) + 64 h + 8 (c & 7) of the tile, of a B half 64 (c >> 2) + 32 h
    // + 8 (c & 3).
    const int cs = (lane & 15) ^ (2 * ((lane >> 4) | ((w & 1) << 2)));
    uint32_t a_voff[2][2], b_voff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t row = 8 * w + 4 * i + (lane >> 4);
            a_voff[h][i] = (row * (uint32_t)lda + 128 * (cs >> 3) + 64 * h + 8 * (cs & 7)) * 2;
            b_voff[h][i] = (row * (uint32_t)ldb + 64 * (cs >> 2) + 32 * h + 8 * (cs & 3)) * 2;
        }
    const uint32_t a_step = (uint32_t)(BK * lda * 2), b_step = (uint32_t)(BK * ldb * 2);    // bytes per K-tile
    const uint32_t lds_w = (uint32_t)(uintptr_t)(lds_void*)lds + w * 2048;

    // half-tile g = 4 t + j (j = 0: A half 0, 1: B half 0, 2: B half 1, 3: A half 1) lives in slot g & 7
#define W2_ISSUE(j, tp)                                                                   \
    {                                                                                     \
        const uint32_t la = lds_w + (uint32_t)((((tp) & 1) * 4 + (j)) * HALF);            \
        if ((j) == 0) W2_DMA2(a_voff[0][0], a_voff[0][1], ars, la, (uint32_t)(tp) * a_step) \
        else if ((j) == 1) W2_DMA2(b_voff[0][0], b_voff[0][1], brs, la, (uint32_t)(tp) * b_step) \
        else if ((j) == 2) W2_DMA2(b_voff[1][0], b_voff[1][1], brs, la, (uint32_t)(tp) * b_step) \
        else W2_DMA2(a_voff[1][0], a_voff[1][1], ars, la, (uint32_t)(tp) * a_step)        \
    }

    // ---- fragment side.  Lane (g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3) addresses LDS row 8 g + q (+ 4 for the
    // second read, + 32 for the second k-step), 8 bytes at chunk (tile's first chunk + (p >> 1)), half p & 1; f of every
    // one of those rows is 2 (q | (g & 1) << 2).
    const int q4 = l15 >> 2, p4 = lane & 3;
    const int fl = 2 * (q4 | ((kg & 1) << 2));
    const int rowb = (8 * kg + q4) * 256 + 8 * (p4 & 1);
    int a_off[4], b_off[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) a_off[t] = rowb + 16 * (((8 * wm + 2 * t) ^ fl) + (p4 >> 1));
#pragma unroll
    for (int t = 0; t < 2; ++t) b_off[t] = rowb + 16 * (((4 * wn + 2 * t) ^ fl) + (p4 >> 1));
    int bufo = 0;

    f32x4 acc[4][8];               // [n2 tile nb][n1 tile mb]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 AF[4][2], WF0[2][2], WF1[2][2], AN[4];

#define W2_READ_A(dst, mh, k)                                                              \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_)                                    \
        dst[mb_][k] = tfrag(lds + bufo + ((mh) ? 3 : 0) * HALF + a_off[mb_] + (k) * 32 * 256);
#define W2_READ_AN(bufn)                                                                   \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_) AN[mb_] = tfrag(lds + (bufn) + a_off[mb_]);
#define W2_READ_W(dst, nh)                                                                 \
    _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_)                                       \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                   \
            dst[b_][k_] = tfrag(lds + bufo + (1 + (nh)) * HALF + b_off[b_] + k_ * 32 * 256);
#define W2_MFMA(WF, mh, nh)                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_)                                       \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                   \
            _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_)                            \
                acc[2 * (nh) + b_][4 * (mh) + mb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
                    WF[b_][k_], AF[mb_][k_], acc[2 * (nh) + b_][4 * (mh) + mb_], 0, 0, 0);
#define W2_PHASE(WF, mh, nh, issue, j, tp, vm)                                             \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    if (issue) W2_ISSUE(j, tp)                                                             \
    W2_WAIT(vm);                                                                           \
    __builtin_amdgcn_s_barrier();                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    __builtin_amdgcn_s_setprio(1);                                                         \
    W2_MFMA(WF, mh, nh)                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    __builtin_amdgcn_s_barrier();                                                          \
    __builtin_amdgcn_sched_barrier(0);
    // One K-tile (see gemm_bf16.hip: phases, counted waits, slot reuse); the k-step 0 fragments of the next K-tile's first
    // quadrant are read a phase early (AN), which evens the transposed reads out to 16 / 8 / 16 / 8 per load segment
#define W2_KTILE(t, i1, i2, i3, i4, v1, v2, v3, v4, last)                                                  \
    {                                                                                                      \
        asm volatile("" : "+s"(bufo));                                                                     \
        _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_) AF[mb_][0] = AN[mb_];                          \
        W2_READ_A(AF, 0, 1)                                                                                \
        W2_READ_W(WF0, 0)                                                                                  \
        W2_PHASE(WF0, 0, 0, i1, (D + 0) & 3, (t) + ((D + 0) >> 2), v1)                                     \
        W2_READ_W(WF1, 1)                                                                                  \
        W2_PHASE(WF1, 0, 1, i2, (D + 1) & 3, (t) + ((D + 1) >> 2), v2)                                     \
        W2_READ_A(AF, 1, 0)                                                                                \
        W2_READ_A(AF, 1, 1)                                                                                \
        W2_PHASE(WF1, 1, 1, i3, (D + 2) & 3, (t) + ((D + 2) >> 2), v3)                                     \
        if (!(last)) { W2_READ_AN(bufo ^ (4 * HALF)) }                                                     \
        W2_PHASE(WF0, 1, 0, i4, (D + 3) & 3, (t) + ((D + 3) >> 2), v4)                                     \
        bufo ^= 4 * HALF;                                                                                  \
    }

#pragma unroll
    for (int g = 0; g < D; ++g) W2_ISSUE(g & 3, g >> 2)
    W2_WAIT(6);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    W2_READ_AN(0)
    if (wm) __builtin_amdgcn_s_barrier();                   // waves 4-7 run one barrier behind waves 0-3
    __builtin_amdgcn_sched_barrier(0);
    int t = 0;
    for (; t < nK - 2; ++t) W2_KTILE(t, true, true, true, true, 6, 6, 6, 6, false)
    W2_KTILE(t, true, true, true, false, 6, 6, 6, 4, false)
    ++t;
    W2_KTILE(t, false, false, false, false, 2, 0, 0, 0, true)
    if (!wm) __builtin_amdgcn_s_barrier();
#undef W2_KTILE
#undef W2_PHASE
#undef W2_MFMA
#undef W2_READ_W
#undef W2_READ_A
#undef W2_READ_AN
#undef W2_ISSUE

    // lane (l15, kg): n1 = 16 mb + l15, n2 = 16 nb + 4 kg .. + 3 of its wave tile
    float* pb = part + ((long)slice * N1 + t1 * TM + 128 * wm + l15) * N2 + t2 * TN + 64 * wn + 4 * kg;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
            *reinterpret_cast<f32x4*>(pb + (long)(16 * mb) * N2 + 16 * nb) = acc[nb][mb];
}

}  // namespace w2

// Launched by cwlt_wgrad_bf16 (wgrad.hip) in place of its own kernel; the fixed-order reduce over the S partial tiles is
// wgrad.hip's.  Conditions (checked by the caller): N1 % 256 == 0, N2 % 256 == 0, mslice % 64 == 0, mslice >= 256.
int launch_wgrad2(const void* a, const void* b, float* part, long M, int N1, int N2, long lda, long ldb, long mslice, int S,
                  hipStream_t st) {
    const int ntile = (N1 / w2::TM) * (N2 / w2::TN);
    hipLaunchKernelGGL(w2::wgrad2_kernel, dim3((unsigned)(ntile * S)), dim3(512), 0, st, (const bf16_t*)a, (const bf16_t*)b,
                       part, M, N1, N2, lda, ldb, mslice);
    return (int)hipGetLastError();
}

}  // namespace cwlt
