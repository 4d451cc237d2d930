// bf16 projection GEMM of the encoder layer, forward and input-gradient forms:
//
//   C (M, N) [+]= A (M, K) . W (N, K)^T [+ bias (N)]         bf16 operands and result, f32 accumulation
//
// -- the `nn.Linear`s that fast_transformers' AttentionLayer / TransformerEncoderLayer hold (query/key/value/out
// projections, linear1/linear2), built at /root/reference/dqn_policy/model.py:128-137, and the six output heads of
// /root/reference/dqn_policy/model.py:156-161,241-249 as one projection.  Forward: W = the layer's weight (out, in).
// Input gradient: W = the weight TRANSPOSED (a 0.5-2 MB copy kept beside the bf16 shadow of the weight), so that both
// operands are K-contiguous and ONE kernel serves every projection; `accumulate` adds the product onto the residual
// gradient already in C (what `ds.addmm_(dh, W1)` did through hipBLASLt).
//
// Structure (one 512-thread workgroup per CU, 256 x 256 output tile, BK = 64, v_mfma_f32_16x16x32_bf16):
//   * 8 waves as 2 (rows) x 4 (columns), each 128 x 64 of the tile = 32 accumulator tiles (128 registers);
//   * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) in HALF-TILES of
//     128 rows x 64 k (16 KiB): A rows-half 0, W columns-half 0, W columns-half 1, A rows-half 1 per K-tile, 8 slots =
//     128 KiB, each slot rewritten two K-tiles later.  Rows are unpadded 128 B; 16-byte chunk c of row r is stored at
//     chunk position c ^ f(r) (applied on the SOURCE address, an LDS-DMA lands linearly) with f chosen so that the
//     16-lane groups of a ds_read_b128 hit 16 different 16-byte bank slots: zero conflicts by construction
//     (tools/probes/swizzle_search.py models the groups of MI355X_MICROARCH.md's LDS table);
//   * a K-tile is 4 phases of {load segment: fragment reads + 2 DMA pieces + counted vmcnt | barrier | compute segment:
//     16 MFMAs of one 64 x 32 quadrant x K = 64 | barrier}.  Waves 4-7 run ONE barrier behind waves 0-3, so on every
//     SIMD (waves w and w + 4 share one) a wave in its compute segment sits beside a wave in its load segment:
//     the matrix pipe always has a wave issuing, the LDS reads and DMA issue of the other are hidden behind it;
//   * DMA pieces run D = 5 half-tiles ahead of their use and are never drained inside the loop: each load segment ends
//     with `s_waitcnt vmcnt(2 (D - 2))`, which retires exactly the pieces the NEXT phase reads; a slot is rewritten
//     at the earliest two phases after its last read (one phase for the staggered half, one for its reads to return);
//   * the product is taken transposed (W rows on the MFMA's A operand) and the W rows of an MFMA tile are dealt so that a
//     lane ends up with 8 CONSECUTIVE columns of one output row per pair of tiles: 16-byte stores straight from the
//     accumulators, no LDS round trip in the epilogue.
// Bound: MFMA for K >= 1024 (2 M N K flop), HBM for K = 512 (M (K + N) 2 bytes [+ M N 2 with accumulate]).
#include "cwlt_common.h"
#include <stdlib.h>

namespace cwlt {
namespace gb {

constexpr int TM = 256, TN = 256, BK = 64;
constexpr int HALF = 128 * BK * 2;              // bytes of a half-tile slot: 128 rows x 128 B = 16 KiB
constexpr int NSLOT = 8;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

// two 1 KiB pieces (rows r0 .. r0 + 7 and r0 + 8 .. r0 + 15 of a slot) issued from inline asm so that the waits can be
// counted by hand (through the builtin hipcc drains every piece in flight before the next LDS read); M0 carries the
// LDS address
#define GB_DMA2(v0, v1, rs, la, so)                                                                    \
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

#define GB_FRAG(off) __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + (off)))

// EPI bits: 1 = bias, 2 = accumulate onto C
template <int D, bool PF, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                            const float* __restrict__ bias, bf16_t* C, long M, int N,
                                                            int K, long lda, long ldw, long ldc) {
    __shared__ __attribute__((aligned(1024))) char lds[NSLOT * HALF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;                 // wave tile: rows 128 wm .., columns 64 wn ..
    const int l15 = lane & 15, kg = lane >> 4;
    const int nt = (N + TN - 1) / TN;
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const long mt = (long)(loc / nt) * 8 + xcd;        // the column tiles of a row tile run on ONE XCD, back to back
    const int ct = loc % nt;
    const long m0 = mt * TM;
    if (m0 >= M) return;
    const int n0 = ct * TN;
    const long mrows = min((long)TM, M - m0);
    const int nrows = min(TN, N - n0);

    // descriptors as four SGPRs each; rows past the tile's end read back as zeros (hardware range check)
    const uint64_t abase = (uint64_t)(A + m0 * lda), wbase = (uint64_t)(W + (long)n0 * ldw);
    u32x4_t ars, wrs;
    ars[0] = __builtin_amdgcn_readfirstlane((uint32_t)abase);
    ars[1] = __builtin_amdgcn_readfirstlane((uint32_t)(abase >> 32));
    ars[2] = __builtin_amdgcn_readfirstlane((uint32_t)(((mrows - 1) * lda + K) * 2));
    ars[3] = 0x00020000u;
    wrs[0] = __builtin_amdgcn_readfirstlane((uint32_t)wbase);
    wrs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(wbase >> 32));
    wrs[2] = __builtin_amdgcn_readfirstlane((uint32_t)(((long)(nrows - 1) * ldw + K) * 2));
    wrs[3] = 0x00020000u;

    // ---- DMA side.  Slot row r (0..127) of an A half mh is tile row 128 (r >> 6) + 64 mh + (r & 63), of a W half nh
    // tile column 64 (r >> 5) + 32 nh + (r & 31): every wave finds the rows of its quadrant in one slot.  This wave's
    // two pieces of a slot are rows 16 w + 8 i + (lane >> 3), i = 0, 1; lane l lands at chunk position l & 7, which
    // holds chunk (l & 7) ^ f(r):  f_A(r) = (r >> 1) & 7,  f_W(r) = bit1(r) | bit3(r) << 1 | bit4(r) << 2.
    const int lr = lane >> 3, lp = lane & 7;
    uint32_t a_voff[2][2], w_voff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int arow = 128 * (w >> 2) + 64 * h + 16 * (w & 3) + 8 * i + lr;
            const int ac = lp ^ (4 * i + (lane >> 4));
            a_voff[h][i] = ((uint32_t)arow * (uint32_t)lda + ac * 8) * 2;
            const int wrow = 64 * (w >> 1) + 32 * h + 16 * (w & 1) + 8 * i + lr;
            const int wc = lp ^ (((lane >> 4) & 1) | (i << 1) | ((w & 1) << 2));
            w_voff[h][i] = ((uint32_t)wrow * (uint32_t)ldw + wc * 8) * 2;
        }
    const uint32_t lds_w = (uint32_t)(uintptr_t)(lds_void*)lds + w * 2048;      // this wave's rows inside a slot

    // half-tile g = 4 t + j (j = 0: A half 0, 1: W half 0, 2: W half 1, 3: A half 1) lives in slot g & 7
#define GB_ISSUE(j, tp)                                                                   \
    {                                                                                     \
        const uint32_t la = lds_w + (uint32_t)((((tp) & 1) * 4 + (j)) * HALF);            \
        const uint32_t so = (uint32_t)(tp) * (BK * 2);                                    \
        if ((j) == 0) GB_DMA2(a_voff[0][0], a_voff[0][1], ars, la, so)                    \
        else if ((j) == 1) GB_DMA2(w_voff[0][0], w_voff[0][1], wrs, la, so)               \
        else if ((j) == 2) GB_DMA2(w_voff[1][0], w_voff[1][1], wrs, la, so)               \
        else GB_DMA2(a_voff[1][0], a_voff[1][1], ars, la, so)                             \
    }

    // ---- fragment side.  MFMA 16x16x32: lane l holds operand row l & 15, k = 8 (l >> 4) + j of a 32-wide k-step.
    // A rows (the MFMA's B operand): slot row 64 wm + 16 mb + l15.   W rows (its A operand), dealt so that accumulator
    // register r of lane (l15, kg) in tile nb is column 8 kg + 4 (nb & 1) + r of the 32-column half nb >> 1:
    // slot row 32 wn + 8 (l15 >> 2) + 4 (nb & 1) + (l15 & 3).  For both f(r) = l15 >> 1.
    const int fsw = l15 >> 1;
    int a_off[2], w_off[2];        // k-step 0 / 1 (chunks kg and 4 + kg)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pos = ((kg | (4 * k)) ^ fsw) << 4;
        a_off[k] = (64 * wm + l15) * 128 + pos;
        w_off[k] = (32 * wn + 8 * (l15 >> 2) + (l15 & 3)) * 128 + pos;
    }
    // slot offsets of the current K-tile's buffer, toggled every K-tile
    int bufo = 0;

    f32x4 acc[4][8];               // [column tile nb][row tile mb]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 AF[4][2], WF0[2][2], WF1[2][2], AN[4];

#define GB_READ_A(dst, mh, k)                                                              \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_)                                    \
        dst[mb_][k] = GB_FRAG(bufo + ((mh) ? 3 : 0) * HALF + a_off[k] + mb_ * 2048);
#define GB_READ_AN(bufn)                                                                   \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_)                                    \
        AN[mb_] = GB_FRAG((bufn) + a_off[0] + mb_ * 2048);
#define GB_READ_W(dst, nh)                                                                 \
    _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_)                                       \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                   \
            dst[b_][k_] = GB_FRAG(bufo + (1 + (nh)) * HALF + w_off[k_] + b_ * 512);
#define GB_MFMA(WF, mh, nh)                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_)                                       \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                   \
            _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_)                            \
                acc[2 * (nh) + b_][4 * (mh) + mb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
                    WF[b_][k_], AF[mb_][k_], acc[2 * (nh) + b_][4 * (mh) + mb_], 0, 0, 0);

    // end of a load segment: DMA issue, counted wait, barrier; then the compute segment between two barriers
#define GB_MID(issue, j, tp, vm)                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    if (issue) GB_ISSUE(j, tp)                                                             \
    asm volatile("s_waitcnt vmcnt(" #vm ")" ::: "memory");                                 \
    __builtin_amdgcn_s_barrier();                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    __builtin_amdgcn_s_setprio(1);
#define GB_END                                                                             \
    __builtin_amdgcn_s_setprio(0);                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    __builtin_amdgcn_s_barrier();                                                          \
    __builtin_amdgcn_sched_barrier(0);

    // One K-tile.  i1..i4: whether phase p still issues a half-tile (g = 4 t + p - 1 + D < 4 nK); v1..v4: the counted
    // waits (steady state 2 (D - 2)); last: no next K-tile to pre-read from.
#define GB_KTILE(t, i1, i2, i3, i4, v1, v2, v3, v4, last)                                                  \
    {                                                                                                      \
        /* phase 1: quadrant (rows half 0, columns half 0) */                                              \
        if (PF) {                                                                                          \
            _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_) AF[mb_][0] = AN[mb_];                      \
        } else {                                                                                           \
            GB_READ_A(AF, 0, 0)                                                                            \
        }                                                                                                  \
        GB_READ_A(AF, 0, 1)                                                                                \
        GB_READ_W(WF0, 0)                                                                                  \
        GB_MID(i1, (D + 0) & 3, (t) + ((D + 0) >> 2), v1)                                                  \
        GB_MFMA(WF0, 0, 0)                                                                                 \
        GB_END                                                                                             \
        /* phase 2: (rows half 0, columns half 1) */                                                       \
        GB_READ_W(WF1, 1)                                                                                  \
        GB_MID(i2, (D + 1) & 3, (t) + ((D + 1) >> 2), v2)                                                  \
        GB_MFMA(WF1, 0, 1)                                                                                 \
        GB_END                                                                                             \
        /* phase 3: (rows half 1, columns half 1) */                                                       \
        GB_READ_A(AF, 1, 0)                                                                                \
        GB_READ_A(AF, 1, 1)                                                                                \
        GB_MID(i3, (D + 2) & 3, (t) + ((D + 2) >> 2), v3)                                                  \
        GB_MFMA(WF1, 1, 1)                                                                                 \
        GB_END                                                                                             \
        /* phase 4: (rows half 1, columns half 0); W half 0 is still in registers */                       \
        if (PF && !(last)) { GB_READ_AN(bufo ^ (4 * HALF)) }                                               \
        GB_MID(i4, (D + 3) & 3, (t) + ((D + 3) >> 2), v4)                                                  \
        GB_MFMA(WF0, 1, 0)                                                                                 \
        GB_END                                                                                             \
        bufo ^= 4 * HALF;                                                                                  \
    }

    const int nK = K / BK;                                  // >= 2 (launcher)
    // prologue: half-tiles 0 .. D - 1, then the first two have landed everywhere
#pragma unroll
    for (int g = 0; g < D; ++g) GB_ISSUE(g & 3, g >> 2)
    if (D == 5)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (PF) { GB_READ_AN(0) }
    if (wm) __builtin_amdgcn_s_barrier();                   // waves 4-7 run one barrier behind waves 0-3
    __builtin_amdgcn_sched_barrier(0);

    int t = 0;
    if (D == 5) {
        for (; t < nK - 2; ++t) GB_KTILE(t, true, true, true, true, 6, 6, 6, 6, false)
        // the last two K-tiles: phase P of G = 4 nK issues while P <= G - D, then waits 2 max(0, G - P - 2)
        GB_KTILE(t, true, true, true, false, 6, 6, 6, 4, false)
        ++t;
        GB_KTILE(t, false, false, false, false, 2, 0, 0, 0, true)
    } else {
        for (; t < nK - 2; ++t) GB_KTILE(t, true, true, true, true, 8, 8, 8, 8, false)
        GB_KTILE(t, true, true, false, false, 8, 8, 6, 4, false)
        ++t;
        GB_KTILE(t, false, false, false, false, 2, 0, 0, 0, true)
    }
    if (!wm) __builtin_amdgcn_s_barrier();                  // every wave has passed the same number of barriers
#undef GB_KTILE
#undef GB_MID
#undef GB_END
#undef GB_MFMA
#undef GB_READ_W
#undef GB_READ_A
#undef GB_READ_AN
#undef GB_ISSUE

    // ---- epilogue: lane (l15, kg) holds, for row tile mb and column half q, the 8 columns 32 q + 8 kg .. + 7 of row
    // 16 mb + l15 of its wave tile: registers 0-3 of acc[2 q][mb] then of acc[2 q + 1][mb]
    float bs[2][8];
    if (EPI & 1) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = n0 + 64 * wn + 32 * q + 8 * kg;
            if (col < N) {
                loadf<8>(bias + col, bs[q]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) bs[q][j] = 0.f;
            }
        }
    }
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
        const long row = m0 + 128 * wm + 16 * mb + l15;
        if (row < M) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int col = n0 + 64 * wn + 32 * q + 8 * kg;
                if (col < N) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[2 * q][mb][r];
                        v[4 + r] = acc[2 * q + 1][mb][r];
                    }
                    bf16_t* dst = C + row * ldc + col;
                    if (EPI & 2) {
                        float o[8];
                        load8(dst, o);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += o[j];
                    }
                    if (EPI & 1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += bs[q][j];
                    }
                    store8(dst, v);
                }
            }
        }
    }
}

}  // namespace gb
}  // namespace cwlt

static int g_variant = -1;      // -1: default; bit 0: D = 6 instead of 5; bit 1: no pre-read of the next K-tile

extern "C" {

/* Tuning switch for A/B measurements (tools/bench_gemm.py): variant < 0 restores the default. */
int cwlt_gemm_bf16_tune(int variant) {
    g_variant = variant;
    return CWLT_OK;
}

/* c (M, N) [+]= a (M, K) . w (N, K)^T [+ bias (N) f32]: bf16 operands and result, f32 accumulation.
 * N % 8 == 0, K % 64 == 0, K >= 128, row strides multiples of 8 elements, 16-byte aligned pointers;
 * accumulate != 0: the product (and bias) is added onto the bf16 values already in c. */
int cwlt_gemm_bf16(const void* a, const void* w, const float* bias, void* c, int64_t M, int N, int K, int64_t lda,
                   int64_t ldw, int64_t ldc, int accumulate, void* stream) {
    using namespace cwlt;
    if (M < 0 || N <= 0 || K < 128 || (N % 8) || (K % 64)) return CWLT_ERR_ARG;
    if (M == 0) return CWLT_OK;
    if (!a || !w || !c) return CWLT_ERR_ARG;
    if (((lda | ldw | ldc) & 7) || lda < K || ldw < K || ldc < N) return CWLT_ERR_ARG;
    if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)c | (uintptr_t)bias) & 15) return CWLT_ERR_ARG;
    /* byte offsets inside one row tile / one weight strip are 32-bit (buffer resources); tile bases are 64-bit */
    if ((int64_t)gb::TM * lda * 2 >= (1ll << 31) || (int64_t)gb::TN * ldw * 2 >= (1ll << 31)) return CWLT_ERR_ARG;
    const long mtiles = (M + gb::TM - 1) / gb::TM;
    const long mt8 = (mtiles + 7) / 8 * 8;            // row tiles are dealt to the 8 XCDs: pad to a multiple of 8
    const long nblk = mt8 * ((N + gb::TN - 1) / gb::TN);
    if (nblk >= (1ll << 31)) return CWLT_ERR_ARG;
    const int epi = (bias ? 1 : 0) | (accumulate ? 2 : 0);
    const int var = g_variant < 0 ? 0 : g_variant;
    typedef void (*kfn_t)(const bf16_t*, const bf16_t*, const float*, bf16_t*, long, int, int, long, long, long);
    kfn_t kfn = nullptr;
#define GB_PICK(D_, PF_)                                                                        \
    switch (epi) {                                                                              \
        case 0: kfn = gb::gemm_bf16_kernel<D_, PF_, 0>; break;                                  \
        case 1: kfn = gb::gemm_bf16_kernel<D_, PF_, 1>; break;                                  \
        case 2: kfn = gb::gemm_bf16_kernel<D_, PF_, 2>; break;                                  \
        default: kfn = gb::gemm_bf16_kernel<D_, PF_, 3>; break;                                 \
    }
    switch (var & 3) {
        case 0: GB_PICK(5, true) break;
        case 1: GB_PICK(6, true) break;
        case 2: GB_PICK(5, false) break;
        default: GB_PICK(6, false) break;
    }
#undef GB_PICK
    hipLaunchKernelGGL(kfn, dim3((unsigned)nblk), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)w,
                       bias, (bf16_t*)c, (long)M, N, K, (long)lda, (long)ldw, (long)ldc);
    return (int)hipGetLastError();
}

}  // extern "C"
