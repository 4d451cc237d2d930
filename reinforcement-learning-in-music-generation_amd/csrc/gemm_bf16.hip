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
// Structure (PERSISTENT: one 512-thread workgroup per CU walks the 256 x 256 output tiles; BK = 64,
// v_mfma_f32_16x16x32_bf16):
//   * 8 waves as 2 (rows) x 4 (columns), each 128 x 64 of the tile = 32 accumulator tiles (128 registers);
//   * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) in HALF-TILES of
//     128 rows x 64 k (16 KiB): A rows-half 0, W columns-half 0, W columns-half 1, A rows-half 1 per K-tile, 8 slots =
//     128 KiB, each slot rewritten two K-tiles later.  Rows are unpadded 128 B; 16-byte chunk c of row r is stored at
//     chunk position c ^ f(r) (applied on the SOURCE address, an LDS-DMA lands linearly) with f chosen so that the
//     16-lane groups of a ds_read_b128 hit 16 different 16-byte bank slots: zero conflicts by construction
//     (tools/probes/swizzle_search.py models the groups of MI355X_MICROARCH.md's LDS table);
//   * a K-tile is 4 phases of {load segment: fragment reads, counted vmcnt | barrier | compute segment: 16 MFMAs of one
//     64 x 32 quadrant x K = 64 | barrier}.  Waves 4-7 run ONE barrier behind waves 0-3, so on every SIMD (waves w and
//     w + 4 share one) a wave in its compute segment sits beside a wave in its load segment: the matrix pipe always
//     has a wave issuing, the LDS reads of the other are hidden behind it;
//   * DMA pieces run D half-tiles ahead of their use and are never drained inside the loop: each load segment ends with
//     a counted `s_waitcnt vmcnt`, which retires exactly the pieces the NEXT phase reads; a slot is rewritten at the
//     earliest two phases after its last read (one phase for the staggered half, one for its reads to return).  The two
//     pieces a wave issues per phase go out at the END of its load segment (between the MFMAs of the compute segment they
//     cost 4-25 %: profiles/r04_gemm_dma_placement.txt);
//   * the next tile's first four half-tiles are requested from inside this tile's LAST K-tile (its successor starts in
//     the buffer that K-tile leaves free), the fifth before the epilogue: the stores of one tile and the load latency of
//     the next overlap; the stores are buffer stores (row / column edges by the hardware range check: no
//     branch, always 16 per lane) so that the counted waits of the next tile can step over them (vmcnt counts loads,
//     stores and LDS-DMA together, in issue order); the bias strip sits in LDS (a vector load in the epilogue would make
//     the compiler drain the DMA pieces in flight);
//   * the product is taken transposed (W rows on the MFMA's A operand) and the W rows of an MFMA tile are dealt so that a
//     lane ends up with 8 CONSECUTIVE columns of one output row per pair of tiles: 16-byte stores straight from the
//     accumulators, no LDS round trip in the epilogue.
// Bound: MFMA for K >= 1024 (2 M N K flop), HBM for K = 512 (M (K + N) 2 bytes [+ M N 2 with accumulate]).
#include "cwlt_common.h"
#include "cwlt_gelu.h"
#include <stdlib.h>

namespace cwlt {
namespace gb {

constexpr int TM = 256, TN = 256, BK = 64;
constexpr int HALF = 128 * BK * 2;              // bytes of a half-tile slot: 128 rows x 128 B = 16 KiB
constexpr int NSLOT = 8;
constexpr int RING = NSLOT * HALF;              // 128 KiB
constexpr int EXTRA = 32768;                    // bias strip (N <= 8192 floats) / stamps of a trace build
constexpr int MAXN_BIAS = EXTRA / 4;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

// one 1 KiB piece (8 rows of a slot) issued from inline asm so that the waits can be counted by hand (through the
// builtin hipcc drains every piece in flight before the next LDS read); M0 carries the LDS address
#define GB_DMA1(v0, rs, la, so)                                                                        \
    {                                                                                                  \
        unsigned keep;                                                                                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"                          \
                     "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"                                    \
                     "s_mov_b32 m0, %0"                                                                \
                     : "=&s"(keep)                                                                     \
                     : "v"(v0), "s"(rs), "s"(la), "s"(so)                                              \
                     : "memory", "scc");                                                               \
    }
// the two pieces of a half-tile (rows r0 .. r0 + 7 and r0 + 8 .. r0 + 15 of the slot) back to back
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
#define GB_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// EPI: 0..3 = bits 1 (bias) | 2 (accumulate onto C);
//      EPI_GELU: x = bf16(a w^T) + bias; c = mask * keep_scale * gelu(x), g (WRITTEN) = mask * keep_scale * gelu'(x) -- the
//                FFN forward `dropout(gelu(linear1(.)))` with the backward's factor alongside (cwlt_gemm_nt_bias_gelu_dropout);
//      EPI_MUL:  c = bf16(a w^T) * g (g READ), part (row tiles, N) = column sums of c -- the FFN backward
//                (cwlt_gemm_nt_mul).  Same arithmetic, rounding points and dropout stream as gemm_nt.hip's 128 x 256 kernel.
// TRACE: s_memtime stamps of one tile's segments (diagnostic build).
// ABL (timing experiments only, results are wrong): 1 = no DMA pieces inside the main loop, 2 = no fragment reads inside
// it, 4 = no barriers inside it, 8 = no counted waits inside it.
// EP: the next tile's first four half-tiles are requested from inside this tile's LAST K-tile (one per phase) instead of
// after the main loop.
enum { EPI_GELU = 4, EPI_MUL = 8 };
struct FfnArgs {               // EPI_GELU / EPI_MUL only
    bf16_t* G;                 // gd: written (GELU) / read (MUL), dense (M, N) like C
    float* part;               // MUL: column-sum partials, (row tiles of 256) x N, or NULL
    uint32_t thresh;
    float keep_scale;
    uint64_t seed;
    const uint64_t* seed_base;
};

template <int EPI, bool TRACE, int ABL = 0, bool EP = true>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                            const float* __restrict__ bias, bf16_t* C, long M, int N,
                                                            int K, long lda, long ldw, long ldc, int nblk,
                                                            uint32_t* __restrict__ trace, int stagger, FfnArgs ffn) {
    constexpr bool BIAS = (EPI < 4 && (EPI & 1)) || EPI == EPI_GELU;
    constexpr bool ACCUM = EPI < 4 && (EPI & 2);
    constexpr bool CIN = ACCUM || EPI == EPI_MUL;          // a (M, N) tile is read back in the epilogue
    constexpr int NSTORE = EPI == EPI_GELU ? 32 : EPI == EPI_MUL ? 17 : 16;   // vector stores per lane and tile
    constexpr int D = 5;           // half-tiles a DMA piece is issued ahead of its use
    __shared__ __attribute__((aligned(1024))) char lds[RING + EXTRA];
    float* lds_bias = reinterpret_cast<float*>(lds + RING);
    uint32_t* lds_trace = reinterpret_cast<uint32_t*>(lds + RING);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;                 // wave tile: rows 128 wm .., columns 64 wn ..
    const int l15 = lane & 15, kg = lane >> 4;
    const int nt = (N + TN - 1) / TN;
    const int nK = K / BK;                             // >= 2 (launcher)

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

    // half-tile g = 4 t + j (j = 0: A half 0, 1: W half 0, 2: W half 1, 3: A half 1) of a tile whose K-tile 0 sits in
    // buffer `pr` lives in slot 4 ((t + pr) & 1) + j
#define GB_ISSUE(j, tp, pr)                                                               \
    {                                                                                     \
        const uint32_t la = lds_w + (uint32_t)(((((tp) + (pr)) & 1) * 4 + (j)) * HALF);   \
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

    bf16x8 AF[4][2], WF0[2][2], WF1[2][2];
    f32x4 acc[4][8];               // [column tile nb][row tile mb]
    int bufo = 0;                  // slot offset of the current K-tile's buffer, toggled every K-tile
    int tq = 0;                    // trace: stamp index

#define GB_STAMP()                                                                         \
    if (TRACE && tracing) {                                                                \
        const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();                      \
        if (lane == 0) lds_trace[w * 1024 + (tq & 1023)] = now_;                           \
        ++tq;                                                                              \
    }
#define GB_READ_A_(dst, mh, k)                                                             \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_)                                    \
        dst[mb_][k] = GB_FRAG(bufo + ((mh) ? 3 : 0) * HALF + a_off[k] + mb_ * 2048);
#define GB_READ_A(dst, mh, k) if (!(ABL & 2)) { GB_READ_A_(dst, mh, k) }
#define GB_READ_W_(dst, nh)                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_)                                       \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                   \
            dst[b_][k_] = GB_FRAG(bufo + (1 + (nh)) * HALF + w_off[k_] + b_ * 512);
#define GB_READ_W(dst, nh) if (!(ABL & 2)) { GB_READ_W_(dst, nh) }
#define GB_MFMA4(WF, mh, nh, b_, k_)                                                       \
    _Pragma("unroll") for (int mb_ = 0; mb_ < 4; ++mb_)                                    \
        acc[2 * (nh) + (b_)][4 * (mh) + mb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(    \
            WF[b_][k_], AF[mb_][k_], acc[2 * (nh) + (b_)][4 * (mh) + mb_], 0, 0, 0);

    // end of a load segment: the phase's two DMA pieces, counted wait, barrier; then the compute segment -- 16 MFMAs --
    // and the closing barrier.  (Pieces issued BETWEEN the MFMAs instead -- every wave behind its 4th and 12th, or each
    // of the four computing waves at its own place -- cost 4-25 % on every shape: a wave cannot issue a vector-memory
    // instruction without holding up the MFMAs behind it.  profiles/r04_gemm_dma_placement.txt)
#define GB_PHASE(WF, mh, nh, issue, j, tp, pr, vm, vm0)                                    \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    if (issue) {              /* compile-time true / false, or one wave-uniform branch (the last K-tile) */ \
        if (!(ABL & 1)) GB_ISSUE(j, tp, pr)                                                \
        if (!(ABL & 8) && (vm) < 32) GB_WAIT(vm);                                          \
    } else {                                                                               \
        if (!(ABL & 8) && (vm0) < 32) GB_WAIT(vm0);                                        \
    }                                                                                      \
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();                                          \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    GB_STAMP()                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                         \
    GB_MFMA4(WF, mh, nh, 0, 0)                                                             \
    GB_MFMA4(WF, mh, nh, 1, 0)                                                             \
    GB_MFMA4(WF, mh, nh, 0, 1)                                                             \
    GB_MFMA4(WF, mh, nh, 1, 1)                                                             \
    __builtin_amdgcn_s_setprio(0);                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();                                          \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    GB_STAMP()

    // One K-tile.  Phase p issues half-tile g0 + p - 1 counted from K-tile t (g0 = D inside a tile; g0 = 0 with t = 0
    // and the NEXT tile's buffer parity for the pieces a tile's last K-tile requests for its successor); i1..i4: whether
    // it issues at all (compile-time, or one wave-uniform condition); v1..v4: the counted waits behind an issue, u1..u4
    // without one (>= 32: none).
#define GB_KTILE(t, g0, pr, i1, i2, i3, i4, v1, v2, v3, v4, u1, u2, u3, u4)                               \
    {                                                                                                      \
        /* the buffer offset is made opaque per K-tile: knowing how it runs from one peeled K-tile to the  */ \
        /* next, hipcc hoists every (offset + lane address) sum out of them and spills 30 registers        */ \
        asm volatile("" : "+s"(bufo));                                                                     \
        /* phase 1: quadrant (rows half 0, columns half 0) */                                              \
        GB_READ_A(AF, 0, 0)                                                                                \
        GB_READ_A(AF, 0, 1)                                                                                \
        GB_READ_W(WF0, 0)                                                                                  \
        GB_PHASE(WF0, 0, 0, i1, ((g0) + 0) & 3, (t) + (((g0) + 0) >> 2), pr, v1, u1)                       \
        /* phase 2: (rows half 0, columns half 1) */                                                       \
        GB_READ_W(WF1, 1)                                                                                  \
        GB_PHASE(WF1, 0, 1, i2, ((g0) + 1) & 3, (t) + (((g0) + 1) >> 2), pr, v2, u2)                       \
        /* phase 3: (rows half 1, columns half 1) */                                                       \
        GB_READ_A(AF, 1, 0)                                                                                \
        GB_READ_A(AF, 1, 1)                                                                                \
        GB_PHASE(WF1, 1, 1, i3, ((g0) + 2) & 3, (t) + (((g0) + 2) >> 2), pr, v3, u3)                       \
        /* phase 4: (rows half 1, columns half 0); W half 0 is still in registers */                       \
        GB_PHASE(WF0, 1, 0, i4, ((g0) + 3) & 3, (t) + (((g0) + 3) >> 2), pr, v4, u4)                       \
        bufo ^= 4 * HALF;                                                                                  \
    }

    // Counted waits.  The wait that ends phase P's load segment must retire half-tile P + 1 and may leave the D - 2
    // half-tiles behind it in flight: S = 2 (D - 2) = 6 pieces in steady state.  In the FIRST K-tile of a tile the 16
    // stores of the previous tile's epilogue sit in the queue behind that tile's last requests: while the half-tile
    // waited for is older than the stores (P <= D - 2 = 3) they may stay outstanding too (+ 16; 32 / 17 for the FFN
    // epilogues).
    constexpr int S = 2 * (D - 2), SF = S + NSTORE, NONE = 63;

    // bias strip -> LDS, once per workgroup (read back in every epilogue without touching the vector-memory queue)
    if (BIAS) {
        for (int i = tid; i < N; i += 512) lds_bias[i] = bias[i];
    }
    uint64_t seed = ffn.seed;
    if (EPI == EPI_GELU && ffn.seed_base) seed += *ffn.seed_base;   // device-resident offset: fresh masks per graph replay

    u32x4_t ars, wrs;
    ars[3] = 0x00020000u;
    wrs[3] = 0x00020000u;
    // tile index -> (row tile, column tile): ids are dealt round-robin over the 8 XCDs, the column tiles of a row tile
    // run on ONE XCD, back to back (the second finds the A strip in that XCD's L2)
#define GB_TILE(idx, m0_, n0_)                                                             \
    const int xcd_ = (idx) & 7, loc_ = (idx) >> 3;                                         \
    const long m0_ = ((long)(loc_ / nt) * 8 + xcd_) * TM;                                  \
    const int n0_ = (loc_ % nt) * TN;
    // descriptors as four SGPRs each; rows past the tile's end read back as zeros (hardware range check)
#define GB_DESC(m0_, n0_)                                                                                   \
    {                                                                                                       \
        const long mr_ = min((long)TM, M - (m0_));                                                          \
        const int nr_ = min(TN, N - (n0_));                                                                 \
        const uint64_t ab_ = (uint64_t)(A + (m0_) * lda), wb_ = (uint64_t)(W + (long)(n0_) * ldw);          \
        ars[0] = __builtin_amdgcn_readfirstlane((uint32_t)ab_);                                             \
        ars[1] = __builtin_amdgcn_readfirstlane((uint32_t)(ab_ >> 32));                                     \
        ars[2] = __builtin_amdgcn_readfirstlane((uint32_t)(((mr_ - 1) * lda + K) * 2));                     \
        wrs[0] = __builtin_amdgcn_readfirstlane((uint32_t)wb_);                                             \
        wrs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(wb_ >> 32));                                     \
        wrs[2] = __builtin_amdgcn_readfirstlane((uint32_t)(((long)(nr_ - 1) * ldw + K) * 2));               \
    }
#define GB_NEXT(from, nx)                                                                  \
    int nx = (from) + gridDim.x;                                                           \
    while (nx < nblk) {            /* padding tiles (row tiles are dealt in eights) hold no rows */ \
        GB_TILE(nx, mc_, nc_)                                                              \
        if (mc_ < M) break;                                                                \
        nx += gridDim.x;                                                                   \
    }

    // Every workgroup takes the same time per tile, so left alone the whole chip stores its 256 x 128 KiB of output and
    // requests the next tiles' first operands in the same few microseconds and then leaves HBM idle for a main loop.
    // Workgroups start `stagger` x 64 cycles apart, one group per ROW tile of an XCD's share (spread over a fraction of
    // a tile period), after which the bursts of different CUs fall into each other's main loops.  The column tiles of a
    // row tile stay together: they read the same A strip, and apart in time each of them would fetch it from HBM again
    // (counter traffic of the N = 2048 forms: 1.25 x algorithmic with the groups dealt by workgroup id).
    if (stagger) {
        for (int left = (int)((blockIdx.x >> 3) / nt) * stagger; left > 0; left -= 64) __builtin_amdgcn_s_sleep(64);
    }
    GB_NEXT((int)blockIdx.x - (int)gridDim.x, idx)
    if (idx >= nblk) return;
    int par = 0;                   // buffer of the current tile's K-tile 0
    {
        GB_TILE(idx, m0p, n0p)
        GB_DESC(m0p, n0p)
        __syncthreads();           // the bias strip is in LDS (and its loads are behind us)
#pragma unroll
        for (int g = 0; g < D; ++g) GB_ISSUE(g & 3, g >> 2, 0)
    }
    bool first = true;
    int it = 0;
    while (true) {
        GB_TILE(idx, m0, n0)
        GB_NEXT(idx, nxt)
        const bool has_next = nxt < nblk;
        const int parn = (par + nK) & 1;                    // the next tile starts in the buffer this tile's last K-tile leaves free
        const bool tracing = TRACE && blockIdx.x == 0 && it == 1;
        GB_STAMP()
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bufo = par * 4 * HALF;
        // half-tiles 0 and 1 have landed everywhere.  A workgroup's first tile has no stores in the queue: it drains
        // its prologue once, after which the "+ 16" waits of the first K-tile hold trivially
        if (first) {
            GB_WAIT(0);
        } else {
            GB_WAIT(SF);
        }
        first = false;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GB_STAMP()
        if (ABL & 2) {                                      // timing experiment: the fragments are read once per tile
            GB_READ_A_(AF, 0, 0) GB_READ_A_(AF, 0, 1) GB_READ_W_(WF0, 0) GB_READ_W_(WF1, 1)
        }
        if (wm) __builtin_amdgcn_s_barrier();               // waves 4-7 run one barrier behind waves 0-3
        __builtin_amdgcn_sched_barrier(0);

        // phase P of G = 4 nK issues half-tile P - 1 + D while that is < G; the waits of the last phases shrink with
        // what is left in flight.  The last K-tile then requests the next tile's half-tiles 0..3 (EP), one per phase:
        // the buffer they go to was last read a K-tile ago, and what is still in flight of THIS tile stays counted.
        int t = 0;
        if (nK == 2) {
            GB_KTILE(0, D, par, true, true, true, false, SF, SF, SF, NONE, NONE, NONE, NONE, 4)
        } else {
            GB_KTILE(0, D, par, true, true, true, true, SF, SF, SF, S, NONE, NONE, NONE, NONE)
            for (t = 1; t < nK - 2; ++t) GB_KTILE(t, D, par, true, true, true, true, S, S, S, S, NONE, NONE, NONE, NONE)
            GB_KTILE(t, D, par, true, true, true, false, S, S, S, NONE, NONE, NONE, NONE, 4)
        }
        {
            const bool early = EP && has_next;
            if (early) {
                GB_TILE(nxt, m0n, n0n)
                GB_DESC(m0n, n0n)                           // this tile's last piece went out a phase ago
            }
            // ONE body for both cases (two copies of a K-tile behind an if / else made the register allocator spill the
            // accumulators at the join): with a successor its four phases issue that tile's half-tiles 0..3 and wait
            // for what is left of this one (4, 4, -, -); without, they wait (2, 0, 0, 0)
            GB_KTILE(0, 0, parn, early, early, early, early, 4, 4, NONE, NONE, 2, 0, 0, 0)
        }
        if (!wm) __builtin_amdgcn_s_barrier();              // every wave has passed the same number of barriers
        __builtin_amdgcn_sched_barrier(0);
        GB_STAMP()

        // ---- epilogue: lane (l15, kg) holds, for row tile mb and column half q, the 8 columns 32 q + 8 kg .. + 7 of
        // row 16 mb + l15 of its wave tile: registers 0-3 of acc[2 q][mb] then of acc[2 q + 1][mb].  Buffer stores:
        // rows past the end fall outside the descriptor, columns past the end get an offset outside it.
        const long mrows = min((long)TM, M - m0);
        const int ncols = min(TN, N - n0);
        const __amdgpu_buffer_rsrc_t crs = make_rsrc(C + m0 * ldc + n0, (uint32_t)(((mrows - 1) * ldc + ncols) * 2));
        uint32_t c_voff[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = 64 * wn + 32 * q + 8 * kg;
            c_voff[q] = col < ncols ? ((uint32_t)(128 * wm + l15) * (uint32_t)ldc + col) * 2 : 0x7ffffff0u;
        }
        const uint32_t c_step = (uint32_t)(16 * ldc * 2);
        u32x4_t cin[8][2];
        if (CIN) {
            // the residual gradient this product is added onto / the factor it is multiplied with: requested first,
            // waited for by hand (the compiler's own counted waits do not know about the DMA pieces in the queue), then
            // the rest of the next tile's prologue.  gd is read once: non-temporal.
            const __amdgpu_buffer_rsrc_t irs =
                EPI == EPI_MUL ? make_rsrc(ffn.G + m0 * ldc + n0, (uint32_t)(((mrows - 1) * ldc + ncols) * 2)) : crs;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    cin[mb][q] = __builtin_amdgcn_raw_buffer_load_b128(irs, (int)(c_voff[q] + mb * c_step), 0,
                                                                       EPI == EPI_MUL ? 2 : 0);
#pragma unroll
            for (int mb = 0; mb < 8; mb += 2)
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(cin[mb][0]), "+v"(cin[mb][1]), "+v"(cin[mb + 1][0]), "+v"(cin[mb + 1][1])::"memory");
        }
        // ---- the next tile of this workgroup: what is left of its first D half-tiles goes out now, before the stores
        // (every LDS read of this tile is done; its K-tile 1 goes where this tile's last K-tile was)
        if (has_next) {
            if (EP) {
#pragma unroll
                for (int g = 4; g < D; ++g) GB_ISSUE(g & 3, g >> 2, parn)
            } else {
                GB_TILE(nxt, m0n, n0n)
                GB_DESC(m0n, n0n)
#pragma unroll
                for (int g = 0; g < D; ++g) GB_ISSUE(g & 3, g >> 2, parn)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        GB_STAMP()
        float bs[2][8];
        if (BIAS) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int col = n0 + 64 * wn + 32 * q + 8 * kg;
                const int cc = col < N ? col : 0;            // the strip holds N floats
                const float4 b0 = *reinterpret_cast<const float4*>(lds_bias + cc);
                const float4 b1 = *reinterpret_cast<const float4*>(lds_bias + cc + 4);
                bs[q][0] = b0.x; bs[q][1] = b0.y; bs[q][2] = b0.z; bs[q][3] = b0.w;
                bs[q][4] = b1.x; bs[q][5] = b1.y; bs[q][6] = b1.z; bs[q][7] = b1.w;
            }
        }
        if (EPI == EPI_GELU) {
            // gemm_nt.hip's EPI_GELU arithmetic chunk by chunk: the pre-activation rounded to bf16, bias in f32, one
            // v_exp per element for value and derivative (cwlt_gelu.h), the dropout keep flags of
            // dropout_mask<8>(seed, row * N + col, thresh) as two 16-bit lane masks per hash word (keep_lanes16)
            const __amdgpu_buffer_rsrc_t grs = make_rsrc(ffn.G + m0 * ldc + n0, (uint32_t)(((mrows - 1) * ldc + ncols) * 2));
            const GeluK gk = gelu_consts(ffn.keep_scale);
            const uint32_t thresh2 = ffn.thresh | (ffn.thresh << 16);
            const uint32_t seed_key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9e3779b9u);
#pragma unroll
            for (int mb = 0; mb < 8; ++mb)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const uint64_t row = (uint64_t)(m0 + 128 * wm + 16 * mb + l15);
                    const uint64_t pair = (row * (uint64_t)N + (uint64_t)(n0 + 64 * wn + 32 * q + 8 * kg)) >> 1;
                    const uint32_t base = (uint32_t)pair * 0x9e3779b1u;
                    const uint32_t key = seed_key ^ ((uint32_t)(pair >> 32) * 0x85ebca6bu);
                    u32x4_t r, dq;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int i0 = 2 * c, i1 = 2 * c + 1;
                        const float a0 = i0 < 4 ? acc[2 * q][mb][i0] : acc[2 * q + 1][mb][i0 - 4];
                        const float a1 = i1 < 4 ? acc[2 * q][mb][i1] : acc[2 * q + 1][mb][i1 - 4];
                        f32x2 x;
                        x[0] = (float)(__bf16)a0 + bs[q][i0];
                        x[1] = (float)(__bf16)a1 + bs[q][i1];
                        f32x2 y, dy;
                        gelu_scaled2(x, gk, y, dy);
                        const uint32_t hw = keep_lanes16(hash32((base + (uint32_t)c * 0x9e3779b1u) ^ key), thresh2);
                        r[c] = f32x2_to_bf16x2(y[0], y[1]) & hw;
                        dq[c] = f32x2_to_bf16x2(dy[0], dy[1]) & hw;
                    }
                    // both outputs are streamed past the caches: at the sizes this kernel runs at (>= 32 768 rows: 128 MiB
                    // per output) neither survives in L2 / MALL until its reader, and with the default policy the 128 KiB of
                    // g per tile pushed the A strips and W slices out of L2 between their uses (counter traffic of the
                    // kernel 6.71 GB per launch against 4.83 algorithmic; variant bit 17 = default policy, for the A/B)
                    __builtin_amdgcn_raw_buffer_store_b128(r, crs, (int)(c_voff[q] + mb * c_step), 0, (ABL & 16) ? 0 : 2);
                    __builtin_amdgcn_raw_buffer_store_b128(dq, grs, (int)(c_voff[q] + mb * c_step), 0, 2);
                }
        } else {
        float cs[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) cs[q][j] = 0.f;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[2 * q][mb][r];
                    v[4 + r] = acc[2 * q + 1][mb][r];
                }
                if (ACCUM) {
                    float o[8];
                    load8(reinterpret_cast<const bf16_t*>(&cin[mb][q]), o);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += o[j];
                }
                if (BIAS) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += bs[q][j];
                }
                if (EPI == EPI_MUL) {
                    // the product rounded to bf16 first (what the unfused GEMM wrote), multiplied in f32, rounded once
                    // more: the arithmetic of the two-kernel path; rows past the end read gd back as zero
                    float o[8];
                    load8(reinterpret_cast<const bf16_t*>(&cin[mb][q]), o);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v[j] = (float)(__bf16)v[j] * o[j];
                        cs[q][j] += v[j];
                    }
                }
                u32x4_t pk;
                pk[0] = f32x2_to_bf16x2(v[0], v[1]);
                pk[1] = f32x2_to_bf16x2(v[2], v[3]);
                pk[2] = f32x2_to_bf16x2(v[4], v[5]);
                pk[3] = f32x2_to_bf16x2(v[6], v[7]);
                __builtin_amdgcn_raw_buffer_store_b128(pk, crs, (int)(c_voff[q] + mb * c_step), 0, EPI == EPI_MUL ? 2 : 0);
            }
        if (EPI == EPI_MUL) {
            // column sums of the tile (the upstream Linear's bias gradient): 8 rows per lane above, the 16 lanes of a
            // row group by DPP, the two row halves of the tile through LDS (the bias strip's place), fixed order.  EVERY
            // wave issues the one store (lanes without a column point outside the descriptor): the counted waits of the
            // next tile step over a fixed number of stores.
            float* red = lds_bias;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float x = cs[q][j];
                    x += dpp_move<0xB1>(x);
                    x += dpp_move<0x4E>(x);
                    x += dpp_move<0x141>(x);
                    x += dpp_move<0x140>(x);
                    cs[q][j] = x;
                }
            if (l15 == 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) red[wm * TN + 64 * wn + 32 * q + 8 * kg + j] = cs[q][j];
            }
            __syncthreads();
            const float tot = tid < TN ? red[tid] + red[TN + tid] : 0.f;
            const __amdgpu_buffer_rsrc_t prs =
                make_rsrc(ffn.part ? ffn.part + (m0 / TM) * (long)N + n0 : nullptr, ffn.part ? (uint32_t)(ncols * 4) : 0u);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, tot), prs, tid < TN ? tid * 4 : 0x7ffffff0, 0, 0);
            __syncthreads();                                   // the strip is free for the next tile's sums
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        GB_STAMP()
        if (!has_next) break;
        idx = nxt;
        par = parn;
        ++it;
    }
    if (TRACE && blockIdx.x == 0) {
        __syncthreads();
        for (int i = tid; i < 8 * 1024; i += 512) trace[i] = lds_trace[i];
    }
#undef GB_KTILE
#undef GB_PHASE
#undef GB_MFMA4
#undef GB_READ_W
#undef GB_READ_A
#undef GB_READ_W_
#undef GB_READ_A_
#undef GB_ISSUE
#undef GB_TILE
#undef GB_DESC
#undef GB_NEXT
#undef GB_STAMP
}

}  // namespace gb
}  // namespace cwlt

static int g_variant = -1;      // -1: default; see cwlt_gemm_bf16_tune
static uint32_t* g_trace = nullptr;

namespace cwlt {

// epi: 0..3 (bias | accumulate << 1), gb::EPI_GELU, gb::EPI_MUL.  Arguments checked by the callers.
static int launch_gemm_big(int epi, const void* a, const void* w, const float* bias, void* c, int64_t M, int N, int K,
                           int64_t lda, int64_t ldw, int64_t ldc, gb::FfnArgs ffn, hipStream_t stream) {
    const long mtiles = (M + gb::TM - 1) / gb::TM;
    const long mt8 = (mtiles + 7) / 8 * 8;            // row tiles are dealt to the 8 XCDs: pad to a multiple of 8
    const long nblk = mt8 * ((N + gb::TN - 1) / gb::TN);
    if (nblk >= (1ll << 30)) return CWLT_ERR_ARG;
    // one workgroup per CU (128 KiB of LDS, 256 registers x 8 waves), in multiples of 8 so that a workgroup's tiles
    // stay on its XCD's share of the tile order
    static int ncu[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!ncu[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        ncu[dev] = n / 8 * 8;
    }
    long grid = nblk < ncu[dev] ? nblk : ncu[dev];
    const int var = g_variant < 0 ? 0 : g_variant;
    const int glim = ((var >> 8) & 255) * 8;
    if (glim && glim < grid) grid = glim;
    // Start stagger, in units of 64 cycles per group step (one group per row tile an XCD holds at a time; a tile period
    // is ~3 500 cycles per K-tile + ~14 000).  Default: half a tile period for K <= 1024 (there a tile's 128 KiB of stores + the next tile's first
    // operands are a third of its time, and with every workgroup at the same point the chip alternates between
    // saturating HBM and leaving it idle: 336 -> 300 us at K = N = 512, R = 524 288), none above (a K = 2048 tile is
    // 55 us of main loop: 905 us with or without, and the stagger costs its own length once per launch).
    const int stag8 = g_variant < 0 ? (K <= 1024 ? 4 : 0) : ((var >> 1) & 7);
    const long ntc = (N + gb::TN - 1) / gb::TN;
    const long ngroup = (grid / 8 + ntc - 1) / ntc > 0 ? (grid / 8 + ntc - 1) / ntc : 1;     // row tiles an XCD holds at a time
    const int stagger = (int)(stag8 * ((K / gb::BK) * 3500L + 14000L) / (8 * ngroup * 64));
    const int abl = (var >> 4) & 15;
    typedef void (*kfn_t)(const bf16_t*, const bf16_t*, const float*, bf16_t*, long, int, int, long, long, long, int,
                          uint32_t*, int, gb::FfnArgs);
    kfn_t kfn = nullptr;
    uint32_t* tr = (epi == 0) ? g_trace : nullptr;
    const bool late = var & 1;
    switch (epi) {
        case 0:
            kfn = tr ? (late ? gb::gemm_bf16_kernel<0, true, 0, false> : gb::gemm_bf16_kernel<0, true, 0, true>)
                     : (late ? gb::gemm_bf16_kernel<0, false, 0, false> : gb::gemm_bf16_kernel<0, false, 0, true>);
            break;
        case 1: kfn = late ? gb::gemm_bf16_kernel<1, false, 0, false> : gb::gemm_bf16_kernel<1, false, 0, true>; break;
        case 2: kfn = late ? gb::gemm_bf16_kernel<2, false, 0, false> : gb::gemm_bf16_kernel<2, false, 0, true>; break;
        case 3: kfn = late ? gb::gemm_bf16_kernel<3, false, 0, false> : gb::gemm_bf16_kernel<3, false, 0, true>; break;
        case gb::EPI_GELU:
            kfn = (var & (1 << 17)) ? gb::gemm_bf16_kernel<gb::EPI_GELU, false, 16, true>
                                    : gb::gemm_bf16_kernel<gb::EPI_GELU, false, 0, true>;
            break;
        case gb::EPI_MUL: kfn = gb::gemm_bf16_kernel<gb::EPI_MUL, false, 0, true>; break;
        default: return CWLT_ERR_ARG;
    }
    if (epi == 0 && !tr && abl) {
        switch (abl) {
            case 1: kfn = gb::gemm_bf16_kernel<0, false, 1>; break;
            case 2: kfn = gb::gemm_bf16_kernel<0, false, 2>; break;
            case 3: kfn = gb::gemm_bf16_kernel<0, false, 3>; break;
            case 4: kfn = gb::gemm_bf16_kernel<0, false, 4>; break;
            case 5: kfn = gb::gemm_bf16_kernel<0, false, 5>; break;
            case 6: kfn = gb::gemm_bf16_kernel<0, false, 6>; break;
            case 7: kfn = gb::gemm_bf16_kernel<0, false, 7>; break;
            default: kfn = gb::gemm_bf16_kernel<0, false, 8>; break;
        }
    }
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(512), 0, stream, (const bf16_t*)a, (const bf16_t*)w, bias,
                       (bf16_t*)c, (long)M, N, K, (long)lda, (long)ldw, (long)ldc, (int)nblk, tr, stagger, ffn);
    return (int)hipGetLastError();
}

// FFN forms on the 256 x 256 persistent kernel (called from gemm_nt.hip's entry points, which have checked the
// arguments): gelu != 0: g = c (written), gd = G (written), bias, dropout (thresh, keep_scale, seed, seed_base);
// gelu == 0: c = (a w^T) * G, part = column-sum partials of c per 256-row tile (or NULL).
int launch_gemm_ffn_big(int gelu, const void* a, const void* w, const float* bias, void* g, void* c, float* part,
                        int64_t M, int N, int K, int64_t lda, int64_t ldw, uint32_t thresh, float keep_scale,
                        uint64_t seed, const uint64_t* seed_base, hipStream_t st) {
    if (K < 128 || (K % 64) || (N % 8) || N > gb::MAXN_BIAS) return CWLT_ERR_ARG;
    if ((int64_t)gb::TM * N * 2 >= (1ll << 30)) return CWLT_ERR_ARG;
    gb::FfnArgs ffn{(bf16_t*)g, part, thresh, keep_scale, seed, seed_base};
    return launch_gemm_big(gelu ? gb::EPI_GELU : gb::EPI_MUL, a, w, bias, c, M, N, K, lda, ldw, N, ffn, st);
}
long gemm_ffn_big_tiles(long M) { return (M + gb::TM - 1) / gb::TM; }

}  // namespace cwlt

extern "C" {

/* Tuning switch for A/B measurements (tools/bench_gemm.py); variant < 0 restores the default.  Bit 0: the next tile's
 * first operands are requested after the main loop instead of from inside its last K-tile.  Bits 1-3: start stagger of
 * the workgroups, in eighths of a tile period (default: 4 for K <= 1024, else 0).  Bits 4-7 (bias-free,
 * non-accumulating launches only): timing experiments with WRONG results -- 1 no DMA pieces, 2 no fragment reads, 4 no
 * barriers, 8 no counted waits inside the main loop.  Bits 8-15: at most that many x 8 workgroups (0: one per CU).
 * trace != NULL (8192 uint32 of device memory): the next bias-free, non-accumulating launches run the diagnostic build,
 * which leaves the s_memtime stamps of workgroup 0's second tile there (8 waves x 1024). */
int cwlt_gemm_bf16_tune(int variant, void* trace) {
    g_variant = variant;
    g_trace = (uint32_t*)trace;
    return CWLT_OK;
}

/* c (M, N) [+]= a (M, K) . w (N, K)^T [+ bias (N) f32]: bf16 operands and result, f32 accumulation.
 * N % 8 == 0 (N <= 8192 with a bias), K % 64 == 0, K >= 128, row strides multiples of 8 elements, 16-byte aligned
 * pointers; accumulate != 0: the product (and bias) is added onto the bf16 values already in c. */
int cwlt_gemm_bf16(const void* a, const void* w, const float* bias, void* c, int64_t M, int N, int K, int64_t lda,
                   int64_t ldw, int64_t ldc, int accumulate, void* stream) {
    using namespace cwlt;
    if (M < 0 || N <= 0 || K < 128 || (N % 8) || (K % 64)) return CWLT_ERR_ARG;
    if (bias && N > gb::MAXN_BIAS) return CWLT_ERR_ARG;
    if (M == 0) return CWLT_OK;
    if (!a || !w || !c) return CWLT_ERR_ARG;
    if (((lda | ldw | ldc) & 7) || lda < K || ldw < K || ldc < N) return CWLT_ERR_ARG;
    if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)c | (uintptr_t)bias) & 15) return CWLT_ERR_ARG;
    /* byte offsets inside one row tile / one weight strip are 32-bit (buffer resources); tile bases are 64-bit */
    if ((int64_t)gb::TM * lda * 2 >= (1ll << 31) || (int64_t)gb::TN * ldw * 2 >= (1ll << 31) ||
        (int64_t)gb::TM * ldc * 2 >= (1ll << 30))
        return CWLT_ERR_ARG;
    const int epi = (bias ? 1 : 0) | (accumulate ? 2 : 0);
    return launch_gemm_big(epi, a, w, bias, c, M, N, K, lda, ldw, ldc, gb::FfnArgs{}, (hipStream_t)stream);
}

}  // extern "C"
