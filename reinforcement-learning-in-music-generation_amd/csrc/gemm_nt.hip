// bf16 "NT" GEMM with the FFN activation gradient in its epilogue (bf16 throughput path).
//
//   dh = (dy . W2t^T) * gd          dy (rows, d_model), W2t (d_ff, d_model) = linear2.weight transposed, gd (rows, d_ff)
//   db1[n] = sum_rows dh[row][n]
// where gd = dropout_mask * keep_scale * gelu'(h + b1) was written by the forward's activation kernel in place of the
// pre-activation h (cwlt_bias_gelu_dropout_fwd, gd_out).  This is the backward of
// `self.dropout(self.activation(self.linear1(y)))` + the input gradient of `linear2` of fast_transformers'
// TransformerEncoderLayer (/root/reference/dqn_policy/model.py:128-137), so far a hipBLASLt GEMM that wrote
// dgact = dy . W2 (2 GiB per layer at B = 512) followed by cwlt_bias_gelu_dropout_bwd, which read it back together
// with h and wrote dh.  Here dgact never exists: the product leaves the accumulators through one multiply.
//
// Roofline: per 128 x 256 output tile 33.5 MFLOP against 16 KB (dy strip, shared by the 8 column tiles of a row
// tile through one XCD's L2) + 64 KB (gd) + 64 KB (dh) of HBM traffic -- at 8 TB/s the bytes take as long as the
// MFMAs at peak, so the kernel is HBM-bound: R * (D + 2 F) * 2 bytes per launch.
//
// Workgroup = 8 waves (2 x 4), tile 128 rows x 256 columns, BK = 32; operands K-contiguous, fragments are plain
// 16-byte LDS reads (row stride 40 bf16 = 80 B); the product is taken transposed (W rows on registers, dy rows on
// lanes) so a lane holds runs of 4 consecutive columns of one output row; the tile goes through LDS and leaves with
// 16-byte coalesced stores.  Operand rows come through buffer resources (hardware range check).  Workgroup ids are
// dealt so that the column tiles of one row tile run on ONE XCD.
#include "cwlt_common.h"
#include "cwlt_gelu.h"
#include <stdlib.h>

namespace cwlt {
namespace gn {

constexpr int TMR = 128, TNC = 256, BK = 32;
constexpr int NSTAGE = 3;                       // LDS ring: 3 x (128 dy rows + 256 W rows) x 64 B = 72 KiB
constexpr int STG = (TMR + TNC) * BK * 2;       // bytes of one stage (24 KiB); the W rows start at TMR * 64
constexpr int LDE = 264;                        // epilogue tile row stride (bf16): 528 B

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// Epilogue modes of the kernel:
//   EPI_MUL   c = bf16(a w^T) * g (g read),  part: (row tiles, N) f32 column sums of c (NULL: not wanted)   [FFN backward]
//   EPI_GELU  x = bf16(a w^T) + bias;  c = mask * keep_scale * gelu(x),  g (WRITTEN) = mask * keep_scale * gelu'(x)
//             -- linear1 + bias + GELU + dropout of the FFN forward with the backward's factor alongside: the
//             arithmetic, the rounding of the pre-activation to bf16 and the dropout stream (seed, element index) of
//             cwlt_bias_gelu_dropout_fwd(gd_out) applied to a hipBLASLt product, without the pre-activation ever
//             reaching HBM                                                                                  [FFN forward]
enum { EPI_MUL = 0, EPI_GELU = 1 };

template <int EPI, bool NT_STREAMS, bool ILV = true>
__global__ __launch_bounds__(512, 2) void gemm_nt_mul_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* G, bf16_t* __restrict__ Cout,
    float* __restrict__ part, long M, int N, int K, long lda, long ldw, long ldg, long ldc,
    const float* __restrict__ bias, uint32_t thresh, float keep_scale, uint64_t seed,
    const uint64_t* __restrict__ seed_base, int spread) {
    // operand ring: NSTAGE x [dy rows | W rows], 64-byte rows (32 k values), unpadded; the epilogue tile
    // [128][264] bf16 = 67 584 B reuses it
    __shared__ __attribute__((aligned(16))) char lds[NSTAGE * STG];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;                // wave tile: rows 64 wm.., columns 64 wn..
    const int l31 = lane & 31, hf = lane >> 5;
    const int nt = N / TNC;
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const long mt = (long)(loc / nt) * 8 + xcd;       // row tile: all its column tiles on one XCD
    const int ct = loc % nt;
    const long m0 = mt * TMR;
    if (m0 >= M) return;
    const int n0 = ct * TNC;
    // The first resident workgroups all start together and every tile takes the same time, so without this the whole
    // chip alternates between a phase in which nobody touches HBM (main loops) and one in which everybody stores
    // (epilogues).  A start offset of 0..7 x ~4 us for the first two workgroups of every CU spreads the epilogues over
    // the tile time; later workgroups inherit the spread from the ones they replace.
    if (spread && blockIdx.x < 2 * 256 && gridDim.x > 2 * 256)
        for (int i = (blockIdx.x >> 3) & 7; i > 0; --i) __builtin_amdgcn_s_sleep(127);
    const long mrows = min((long)TMR, M - m0);

    // Staging goes global -> LDS directly (buffer_load_dwordx4 ... lds), as in wgrad.hip: no VGPR round trip and no
    // 16-byte LDS stores (the register-staged version's ds_write_b128 had 2-way bank conflicts at the padded 80-byte
    // row stride and, with the fragment reads, kept the LDS pipe busier than the MFMA pipe: 61 % against 30 %).  An
    // LDS-DMA wave-instruction writes 1 KiB linearly = 16 unpadded rows; the 16-byte fragment reads of 16 consecutive
    // rows stay conflict-free through an XOR swizzle applied on the SOURCE side: the 16-byte chunk c of row r is
    // stored at chunk position c ^ ((r >> 2) & 3).
    // Descriptors as four SGPRs each; rows past the tile's end read back as zeros (hardware range check).
    const uint64_t abase = (uint64_t)(A + m0 * lda), wbase = (uint64_t)(W + (long)n0 * ldw);
    u32x4_t ars, wrs;
    ars[0] = __builtin_amdgcn_readfirstlane((uint32_t)abase);
    ars[1] = __builtin_amdgcn_readfirstlane((uint32_t)(abase >> 32));
    ars[2] = __builtin_amdgcn_readfirstlane((uint32_t)(((mrows - 1) * lda + K) * 2));
    ars[3] = 0x00020000u;
    wrs[0] = __builtin_amdgcn_readfirstlane((uint32_t)wbase);
    wrs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(wbase >> 32));
    wrs[2] = __builtin_amdgcn_readfirstlane((uint32_t)(((long)(TNC - 1) * ldw + K) * 2));
    wrs[3] = 0x00020000u;
    // this wave's three DMA pieces per step: dy rows 16w .. 16w + 15, W rows 16w .. and 128 + 16w ..  Lane l lands at
    // piece base + 16 l = (row l >> 2, chunk position l & 3), which holds chunk (l & 3) ^ ((l >> 4) & 3).
    const int drow = 16 * w + (lane >> 2);
    const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);
    const uint32_t a_voff = ((uint32_t)drow * (uint32_t)lda + dchunk * 8) * 2;
    const uint32_t w_voff = ((uint32_t)drow * (uint32_t)ldw + dchunk * 8) * 2;
    const uint32_t w_half = (uint32_t)(128 * ldw * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)lds + w * 1024;      // this wave's dy piece inside a stage
    // issued from inline asm so that the waits can be counted by hand (see wgrad.hip); M0 carries the LDS address
#define GN_DMA(stage, step)                                                                                       \
    {                                                                                                             \
        unsigned keep;                                                                                            \
        const uint32_t la = lds0 + (uint32_t)(stage) * STG;                                                       \
        const uint32_t sk_ = (uint32_t)(step) * (BK * 2), sk2 = sk_ + w_half;                                     \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"                                     \
                     "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"                                               \
                     "s_add_u32 m0, %3, 0x2000\n\ts_nop 0\n\t"                                                   \
                     "buffer_load_dwordx4 %5, %6, %4 offen lds\n\t"                                               \
                     "s_add_u32 m0, %3, 0x4000\n\ts_nop 0\n\t"                                                   \
                     "buffer_load_dwordx4 %5, %6, %7 offen lds\n\t"                                               \
                     "s_mov_b32 m0, %0"                                                                           \
                     : "=&s"(keep)                                                                                \
                     : "v"(a_voff), "s"(ars), "s"(la), "s"(sk_), "v"(w_voff), "s"(wrs), "s"(sk2)                  \
                     : "memory", "scc");                                                                          \
    }
    // the same three pieces one at a time (which = 0: the dy piece, 1 / 2: the two W pieces), for the main loop, where they
    // are issued BETWEEN the step's MFMA groups: 24 pieces issued at once behind the barrier fill the texture-address FIFOs
    // (SQ_VMEM_TA_ADDR_FIFO_FULL / _CMD_FIFO_FULL: a quarter of the CU-busy cycles, profiles/r03_ffn1_pmc.txt) and a wave
    // that cannot issue its piece cannot issue the MFMAs behind it either
#define GN_DMA1(stage, step, which)                                                                               \
    {                                                                                                             \
        unsigned keep;                                                                                            \
        const uint32_t la = lds0 + (uint32_t)(stage) * STG + ((which) == 0 ? 0u : (which) == 1 ? 0x2000u : 0x4000u); \
        const uint32_t sk_ = (uint32_t)(step) * (BK * 2) + ((which) == 2 ? w_half : 0u);                         \
        if ((which) == 0)                                                                                         \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"                              \
                         "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"                         \
                         : "=&s"(keep)                                                                            \
                         : "v"(a_voff), "s"(ars), "s"(la), "s"(sk_)                                               \
                         : "memory", "scc");                                                                      \
        else                                                                                                      \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"                              \
                         "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"                         \
                         : "=&s"(keep)                                                                            \
                         : "v"(w_voff), "s"(wrs), "s"(la), "s"(sk_)                                               \
                         : "memory", "scc");                                                                      \
    }
    // fragment byte offsets of this lane inside a row block: row l31, chunk (2 ks + hf) at position ^ ((l31 >> 2) & 3)
    const int swz = (l31 >> 2) & 3;
    const int of0 = l31 * 64 + ((hf ^ swz) << 4), of1 = l31 * 64 + (((2 + hf) ^ swz) << 4);
    const int oa = (64 * wm) * 64, ow = TMR * 64 + (64 * wn) * 64;      // wave tile bases; + 32 rows = + 2048 bytes
#define GN_FRAG(p) __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p))
#define GN_COMPUTE(stage)                                                                     \
    {                                                                                         \
        const char* sb = lds + (stage) * STG;                                                 \
        const bf16x8 w00 = GN_FRAG(sb + ow + of0), w01 = GN_FRAG(sb + ow + 2048 + of0);       \
        const bf16x8 x00 = GN_FRAG(sb + oa + of0), x01 = GN_FRAG(sb + oa + 2048 + of0);       \
        const bf16x8 w10 = GN_FRAG(sb + ow + of1), w11 = GN_FRAG(sb + ow + 2048 + of1);       \
        const bf16x8 x10 = GN_FRAG(sb + oa + of1), x11 = GN_FRAG(sb + oa + 2048 + of1);       \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w00, x00, acc[0][0], 0, 0, 0);    \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w01, x00, acc[0][1], 0, 0, 0);    \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w00, x01, acc[1][0], 0, 0, 0);    \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w01, x01, acc[1][1], 0, 0, 0);    \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w10, x10, acc[0][0], 0, 0, 0);    \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w11, x10, acc[0][1], 0, 0, 0);    \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w10, x11, acc[1][0], 0, 0, 0);    \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w11, x11, acc[1][1], 0, 0, 0);    \
    }

    // a step with the next-but-one step's three pieces issued between its MFMA groups (more = false: no pieces)
#define GN_COMPUTE_DMA(stage, nstage, nstepi, more)                                           \
    {                                                                                         \
        const char* sb = lds + (stage) * STG;                                                 \
        {                                                                                     \
            const bf16x8 w00 = GN_FRAG(sb + ow + of0), w01 = GN_FRAG(sb + ow + 2048 + of0);   \
            const bf16x8 x00 = GN_FRAG(sb + oa + of0), x01 = GN_FRAG(sb + oa + 2048 + of0);   \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w00, x00, acc[0][0], 0, 0, 0); \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w01, x00, acc[0][1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                \
            if (more) GN_DMA1(nstage, nstepi, 0);                                             \
            __builtin_amdgcn_sched_barrier(0);                                                \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w00, x01, acc[1][0], 0, 0, 0); \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w01, x01, acc[1][1], 0, 0, 0); \
        }                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (more) GN_DMA1(nstage, nstepi, 1);                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        {                                                                                     \
            const bf16x8 w10 = GN_FRAG(sb + ow + of1), w11 = GN_FRAG(sb + ow + 2048 + of1);   \
            const bf16x8 x10 = GN_FRAG(sb + oa + of1), x11 = GN_FRAG(sb + oa + 2048 + of1);   \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w10, x10, acc[0][0], 0, 0, 0); \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w11, x10, acc[0][1], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                \
            if (more) GN_DMA1(nstage, nstepi, 2);                                             \
            __builtin_amdgcn_sched_barrier(0);                                                \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w10, x11, acc[1][0], 0, 0, 0); \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w11, x11, acc[1][1], 0, 0, 0); \
        }                                                                                     \
    }

    f32x16 acc[2][2];   // [row half i][column half j]: registers = columns (n), lanes = rows (m)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float b[8];                                       // EPI_GELU: this thread's 8 bias values, fetched under the main loop
    if (EPI == EPI_GELU) loadf<8>(bias + n0 + (tid & 31) * 8, b);

    // Ring of 3 stages, DMA two steps ahead, ONE barrier per step:
    //   wait until this wave's pieces of step s have landed (vmcnt leaves the younger step's 3 pieces in flight);
    //   barrier: every wave's pieces of step s have landed, and every wave has finished reading step s - 1;
    //   issue step s + 2 into the stage step s - 1 used;  compute step s.
    const int nstep = K / BK;                         // K is a multiple of 64 (launcher): nstep >= 2
    GN_DMA(0, 0);
    GN_DMA(1, 1);
    for (int s = 0; s < nstep; ++s) {
        if (s + 1 < nstep)
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int st = s % NSTAGE;
        if (ILV) {
            GN_COMPUTE_DMA(st, (s + 2) % NSTAGE, s + 2, s + 2 < nstep);
        } else {
            if (s + 2 < nstep) GN_DMA((s + 2) % NSTAGE, s + 2);
            GN_COMPUTE(st);
        }
    }
    __syncthreads();                                  // every wave is done with the ring: the epilogue tile reuses it
#undef GN_DMA
#undef GN_DMA1
#undef GN_COMPUTE_DMA
#undef GN_FRAG
#undef GN_COMPUTE

    // epilogue.  This thread's 8 chunks of the tile: rows (tid >> 5) + 16 i, columns 8 (tid & 31) .. + 7.
    const int erow = tid >> 5, ecol = (tid & 31) * 8;
    uint4 gv[8];
    if (EPI == EPI_MUL) {
        // the gd chunks are requested first (the accumulators are still being written to LDS while they fly)
        const __amdgpu_buffer_rsrc_t gr =
            make_rsrc(G + m0 * ldg + n0, (uint32_t)(((mrows - 1) * ldg + TNC) * 2));      // rows >= mrows read zeros
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // once-read stream: non-temporal (aux = 2) so that it does not evict the operand strips the co-resident
            // workgroup's main loop re-reads from L2 (CWLT_GEMM_NT=0 builds use the default policy for A/B comparison)
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(
                gr, (int)(((uint32_t)(erow + 16 * i) * (uint32_t)ldg + ecol) * 2), 0, NT_STREAMS ? 2 : 0);
            gv[i] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }

    // f32 product * gd would need the f32 tile in LDS (135 KB); the tile is rounded to bf16 first (as the unfused
    // GEMM's output was) and multiplied in f32, rounded once more: the arithmetic of the two-kernel path.
    bf16_t* et = reinterpret_cast<bf16_t*>(lds);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 64 * wm + 32 * i + l31, c0 = 64 * wn + 32 * j;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 p;
#pragma unroll
                for (int u = 0; u < 4; ++u) p[u] = (__bf16)acc[i][j][4 * g + u];
                *reinterpret_cast<uint2*>(et + row * LDE + c0 + 8 * g + 4 * hf) = __builtin_bit_cast(uint2, p);
            }
        }
    __syncthreads();
    if (EPI == EPI_GELU) {
        // The bias values were requested under the main loop.  They must be waited for HERE, outside the row loop: each
        // row's body is a branch (`row < mrows`), the compiler's wait for b[] sits inside the first body only, so on the
        // path that skips a body b[] stays "in flight" in its bookkeeping and every later body begins with
        // `s_waitcnt vmcnt(0)` -- which by then waits for the previous row's two STORES to complete: eight write round
        // trips in a row, 11.8 of the epilogue's 13.2 us per tile (tools/probes/ffn1_trace.py).
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(b[j]));
        if (seed_base) seed += *seed_base;   // device-resident offset: a captured hipGraph draws fresh masks per replay
        const GeluK gk = gelu_consts(keep_scale);
        const uint32_t thresh2 = thresh | (thresh << 16);
        // Everything that depends on the row only through `erow + 16 i` is carried from row to row by additions: the two
        // output addresses (+ 16 rows), the dropout key's pair index (+ 16 N / 2) and its first multiply
        // ((lo + d) C = lo C + d C, exact mod 2^32) -- the row loop held a 64-bit multiply and three 32-bit ones per row
        // for them, in a kernel whose epilogue instructions cost the neighbouring workgroup's main loop their time
        // (HISTORY 4.2c).
        // element index of (row, column) in the dense (M, N) activation: what cwlt_bias_gelu_dropout_fwd keys its mask
        // with (the launcher insists on ldc == ldg == N).  The keep flags of dropout_mask<8>(seed, off, thresh), bit for
        // bit: the four pair indices (off >> 1) + 0..3 share their upper word and key (off is a multiple of 8, so the low
        // word does not carry), and (lo + j) * C = lo * C + j * C -- one 32-bit multiply instead of four in front of the
        // four hashes; the two 16-bit halves of a hash word become two 16-bit lane masks that are ANDed onto the packed
        // pair of bf16 results (no compare / select per element).
        uint64_t pair = ((uint64_t)(m0 + erow) * (uint64_t)N + (uint64_t)(n0 + ecol)) >> 1;
        const uint64_t pair_step = (uint64_t)N * 8;                                 // 16 rows of N elements, in pairs
        uint32_t base = (uint32_t)pair * 0x9e3779b1u;
        const uint32_t base_step = (uint32_t)pair_step * 0x9e3779b1u;
        const uint32_t seed_key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9e3779b9u);
        bf16_t* pc = Cout + (m0 + erow) * ldc + n0 + ecol;
        bf16_t* pg = G + (m0 + erow) * ldg + n0 + ecol;
        const long cstep = 16 * ldc, gstep = 16 * ldg;
#pragma unroll
        for (int i = 0; i < 8; ++i, pair += pair_step, base += base_step, pc += cstep, pg += gstep) {
            const int row = erow + 16 * i;
            if (row < mrows) {
                const uint4 hv = *reinterpret_cast<const uint4*>(et + row * LDE + ecol);
                float t[8];
                load8(reinterpret_cast<const bf16_t*>(&hv), t);
                const uint32_t key = seed_key ^ ((uint32_t)(pair >> 32) * 0x85ebca6bu);
                u32x4_t r, q;
                // The four pairs of a chunk are worked on STAGE BY STAGE, not pair by pair: left alone the scheduler keeps
                // live ranges short and emits each pair's chain of ~45 dependent instructions in one piece (a dependent
                // VALU instruction issues every ~8 cycles, and two of a SIMD's four waves are in this loop); with the
                // stages fenced every instruction has three independent ones between itself and its consumer.
                uint32_t hw[4];
                f32x2 xx[4], ee[4], aa[4], qq[4], us[4], yy[4], dd[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    hw[c] = (base + (uint32_t)c * 0x9e3779b1u) ^ key;
                    xx[c] = f32x2{t[2 * c] + b[2 * c], t[2 * c + 1] + b[2 * c + 1]};
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) hw[c] ^= hw[c] >> 16;
#pragma unroll
                for (int c = 0; c < 4; ++c) ee[c] = (xx[c] * xx[c]) * GELU_NEG_HALF_LOG2E;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) hw[c] *= 0x7feb352du;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ee[c][0] = __builtin_amdgcn_exp2f(ee[c][0]);
                    ee[c][1] = __builtin_amdgcn_exp2f(ee[c][1]);
                    aa[c][0] = fabsf(xx[c][0]);
                    aa[c][1] = fabsf(xx[c][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) hw[c] ^= hw[c] >> 15;
#pragma unroll
                for (int c = 0; c < 4; ++c) qq[c] = __builtin_elementwise_fma(aa[c], f32x2{gk.c5, gk.c5}, f32x2{gk.c4, gk.c4});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) hw[c] *= 0x846ca68bu;
#pragma unroll
                for (int c = 0; c < 4; ++c) qq[c] = __builtin_elementwise_fma(aa[c], qq[c], f32x2{gk.c3, gk.c3});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) hw[c] ^= hw[c] >> 16;
#pragma unroll
                for (int c = 0; c < 4; ++c) qq[c] = __builtin_elementwise_fma(aa[c], qq[c], f32x2{gk.c2, gk.c2});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) hw[c] = keep_lanes16(hw[c], thresh2);
#pragma unroll
                for (int c = 0; c < 4; ++c) qq[c] = __builtin_elementwise_fma(aa[c], qq[c], f32x2{gk.c1, gk.c1});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) qq[c] = __builtin_elementwise_fma(aa[c], qq[c], f32x2{gk.c0, gk.c0});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) us[c] = __builtin_elementwise_fma(-ee[c], qq[c], f32x2{gk.c0, gk.c0});
#pragma unroll
                for (int c = 0; c < 4; ++c) ee[c] = ee[c] * xx[c];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x2 cdf;
                    cdf[0] = gk.c0 + copysignf(us[c][0], xx[c][0]);
                    cdf[1] = gk.c0 + copysignf(us[c][1], xx[c][1]);
                    dd[c] = __builtin_elementwise_fma(ee[c], f32x2{gk.pdfc, gk.pdfc}, cdf);
                    yy[c] = __builtin_elementwise_fma(aa[c], us[c], xx[c] * gk.c0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    r[c] = f32x2_to_bf16x2(yy[c][0], yy[c][1]) & hw[c];
                    q[c] = f32x2_to_bf16x2(dd[c][0], dd[c][1]) & hw[c];
                }
                // g is the next GEMM's operand: default policy; gd waits for the backward: streamed past the caches
                *reinterpret_cast<u32x4_t*>(pc) = r;
                u32x4_t* dst = reinterpret_cast<u32x4_t*>(pg);
                if (NT_STREAMS)
                    __builtin_nontemporal_store(q, dst);
                else
                    *dst = q;
            }
        }
        return;
    }
    // All eight gd chunks are waited for HERE (they were requested before the tile went to LDS).  Left to the compiler the
    // waits sit in the row bodies as vmcnt(7), vmcnt(6), ... vmcnt(0), counted as if no store had been issued in between
    // (the bodies are branches), so from the fourth row on each wait also retires the STORES of earlier rows: several
    // write round trips in a row (see the EPI_GELU branch above).
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(gv[i].x), "v"(gv[i].y), "v"(gv[i].z), "v"(gv[i].w));
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
    bf16_t* pc = Cout + (m0 + erow) * ldc + n0 + ecol;      // carried from row to row (+ 16 rows), as in the EPI_GELU branch
    const long cstep = 16 * ldc;
#pragma unroll
    for (int i = 0; i < 8; ++i, pc += cstep) {
        const int row = erow + 16 * i;
        const uint4 hv = *reinterpret_cast<const uint4*>(et + row * LDE + ecol);
        float t[8], gg[8];
        load8(reinterpret_cast<const bf16_t*>(&hv), t);
        load8(reinterpret_cast<const bf16_t*>(&gv[i]), gg);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            t[j] *= gg[j];               // rows past the end: gd read back as zero -> contributes nothing
            cs[j] += t[j];
        }
        if (row < mrows) {
            u32x4_t r;
            r[0] = f32x2_to_bf16x2(t[0], t[1]);
            r[1] = f32x2_to_bf16x2(t[2], t[3]);
            r[2] = f32x2_to_bf16x2(t[4], t[5]);
            r[3] = f32x2_to_bf16x2(t[6], t[7]);
            u32x4_t* dst = reinterpret_cast<u32x4_t*>(pc);
            if (NT_STREAMS)
                __builtin_nontemporal_store(r, dst);
            else
                *dst = r;
        }
    }
    if (part) {
        // column sums over the tile's 128 rows: 16 threads share a column chunk (erow = 0..15)
        __syncthreads();                                   // the tile is consumed: reuse LDS as f32 [16][256]
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int j = 0; j < 8; ++j) red[erow * TNC + ecol + j] = cs[j];
        __syncthreads();
        if (tid < TNC) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * TNC + tid];
            part[mt * N + n0 + tid] = s;
        }
    }
}


}  // namespace gn
}  // namespace cwlt

static int interleave_dma() {   // CWLT_GEMM_NT_ILV=0: all of a step's pieces issued behind the barrier (A/B switch)
    static const int v = [] { const char* e = getenv("CWLT_GEMM_NT_ILV"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
}
// CWLT_FFN_BIG=0: always the 128 x 256 two-workgroups-per-CU kernel below (A/B switch).  Default: from 32 768 rows on the
// FFN forms run on gemm_bf16.hip's 256 x 256 persistent kernel (one workgroup per CU needs many row tiles to fill the chip).
// (CWLT_FFN_BIG=f / b: only the forward / only the backward form.)
static bool ffn_big(int64_t M, int N, int K, bool fwd = true) {
    static const int v = [] {
        const char* e = getenv("CWLT_FFN_BIG");
        return !e ? 3 : e[0] == '0' ? 0 : e[0] == 'f' ? 1 : e[0] == 'b' ? 2 : 3;
    }();
    static const long min_rows = [] { const char* e = getenv("CWLT_FFN_BIG_MIN_ROWS"); return e ? atol(e) : 32768L; }();
    return (v & (fwd ? 1 : 2)) && M >= min_rows && K >= 128 && N <= 8192;
}
static int spread_starts() {   // CWLT_GEMM_NT_SPREAD=0: all workgroups start at once (A/B switch)
    static const int v = [] { const char* e = getenv("CWLT_GEMM_NT_SPREAD"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
}

extern "C" {

/* Row tiles of cwlt_gemm_nt_mul = rows of its column-sum partials. */
int64_t cwlt_gemm_nt_tiles(int64_t M) { return (M + cwlt::gn::TMR - 1) / cwlt::gn::TMR; }

/* c (M, N) = (a (M, K) . w (N, K)^T) * g (M, N), bf16 operands and output, f32 accumulation; the product is rounded
 * to bf16 before the multiply (what the unfused pair did).  colsum (N) f32, may be NULL with part: column sums of c
 * (the bias gradient), through part (cwlt_gemm_nt_tiles(M) * N floats of workspace), fixed summation order.
 * N % 256 == 0, K % 64 == 0, row strides multiples of 8 elements, 16-byte aligned pointers. */
int cwlt_gemm_nt_mul(const void* a, const void* w, const void* g, void* c, float* part, float* colsum, int64_t M, int N,
                     int K, int64_t lda, int64_t ldw, int64_t ldg, int64_t ldc, void* stream) {
    using namespace cwlt;
    if (M < 0 || N <= 0 || K <= 0 || (N % gn::TNC) || (K % 64)) return CWLT_ERR_ARG;
    if ((part == nullptr) != (colsum == nullptr)) return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) return colsum ? (int)hipMemsetAsync(colsum, 0, sizeof(float) * N, st) : CWLT_OK;
    if (!a || !w || !g || !c) return CWLT_ERR_ARG;
    if (((lda | ldw | ldg | ldc) & 7) || lda < K || ldw < K || ldg < N || ldc < N) return CWLT_ERR_ARG;
    if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)g | (uintptr_t)c) & 15) return CWLT_ERR_ARG;
    /* byte offsets inside one row tile / one weight strip are 32-bit (buffer resources); tile bases are 64-bit */
    if ((int64_t)gn::TMR * (lda > ldg ? lda : ldg) * 2 >= (1ll << 31) || (int64_t)gn::TNC * ldw * 2 >= (1ll << 31))
        return CWLT_ERR_ARG;
    if (ffn_big(M, N, K, false) && ldg == N && ldc == N) {
        // training sizes: the 256 x 256 persistent kernel of gemm_bf16.hip with this epilogue (same arithmetic; the column
        // sums come per 256-row tile)
        int e = launch_gemm_ffn_big(0, a, w, nullptr, const_cast<void*>(g), c, part, M, N, K, lda, ldw, 0u, 1.0f, 0, nullptr, st);
        if (e || !colsum) return e;
        return launch_colsum_finalize(part, colsum, (int)gemm_ffn_big_tiles(M), (long)N, N, 1.0f, 0, st);
    }
    const long mtiles = (M + gn::TMR - 1) / gn::TMR;
    const long mt8 = (mtiles + 7) / 8 * 8;            // row tiles are dealt to the 8 XCDs: pad to a multiple of 8
    const long nblk = mt8 * (N / gn::TNC);
    static const bool nts = [] { const char* e = getenv("CWLT_GEMM_NT"); return !(e && e[0] == '0'); }();   // A/B switch
    auto kfn = !interleave_dma() ? gn::gemm_nt_mul_kernel<gn::EPI_MUL, true, false>
               : nts             ? gn::gemm_nt_mul_kernel<gn::EPI_MUL, true, true>
                                 : gn::gemm_nt_mul_kernel<gn::EPI_MUL, false, true>;
    hipLaunchKernelGGL(kfn, dim3((unsigned)nblk), dim3(512), 0, st, (const bf16_t*)a, (const bf16_t*)w,
                       const_cast<bf16_t*>((const bf16_t*)g), (bf16_t*)c, part, (long)M, N, K, (long)lda, (long)ldw,
                       (long)ldg, (long)ldc, (const float*)nullptr, 0u, 1.0f, (uint64_t)0, (const uint64_t*)nullptr, spread_starts());
    int e = (int)hipGetLastError();
    if (e || !colsum) return e;
    return launch_colsum_finalize(part, colsum, (int)mtiles, (long)N, N, 1.0f, 0, st);
}

/* FFN forward in one kernel:  x = bf16(a (M, K) . w (N, K)^T) + bias (N) f32;  g = dropout(gelu(x)),
 * gd = mask * keep_scale * gelu'(x)  -- `self.dropout(self.activation(self.linear1(y)))` of fast_transformers'
 * TransformerEncoderLayer (/root/reference/dqn_policy/model.py:128-137) with the factor its backward needs
 * (cwlt_gemm_nt_mul's `g`).  Same arithmetic, same rounding of the pre-activation and same dropout stream as a plain
 * GEMM followed by cwlt_bias_gelu_dropout_fwd(gd_out); the pre-activation never reaches HBM.  g and gd: dense (M, N)
 * bf16 (the mask is keyed by the element index).  N % 256 == 0, K % 64 == 0, lda / ldw multiples of 8, 16-byte
 * aligned pointers, 0 <= p < 1. */
int cwlt_gemm_nt_bias_gelu_dropout(const void* a, const void* w, const float* bias, void* g, void* gd, int64_t M, int N,
                                   int K, int64_t lda, int64_t ldw, float p, uint64_t seed, const uint64_t* seed_base,
                                   void* stream) {
    using namespace cwlt;
    if (M < 0 || N <= 0 || K <= 0 || (N % gn::TNC) || (K % 64) || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (M == 0) return CWLT_OK;
    if (!a || !w || !bias || !g || !gd || g == gd) return CWLT_ERR_ARG;
    if (((lda | ldw) & 7) || lda < K || ldw < K) return CWLT_ERR_ARG;
    if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)g | (uintptr_t)gd | (uintptr_t)bias) & 15) return CWLT_ERR_ARG;
    if ((int64_t)gn::TMR * lda * 2 >= (1ll << 31) || (int64_t)gn::TNC * ldw * 2 >= (1ll << 31)) return CWLT_ERR_ARG;
    if (ffn_big(M, N, K))
        return launch_gemm_ffn_big(1, a, w, bias, gd, g, nullptr, M, N, K, lda, ldw, drop_thresh(p), drop_scale(p), seed,
                                   seed_base, (hipStream_t)stream);
    const long mtiles = (M + gn::TMR - 1) / gn::TMR;
    const long mt8 = (mtiles + 7) / 8 * 8;
    const long nblk = mt8 * (N / gn::TNC);
    static const bool nts = [] { const char* e = getenv("CWLT_GEMM_NT"); return !(e && e[0] == '0'); }();
    auto kfn = !interleave_dma() ? gn::gemm_nt_mul_kernel<gn::EPI_GELU, true, false>
               : nts             ? gn::gemm_nt_mul_kernel<gn::EPI_GELU, true, true>
                                 : gn::gemm_nt_mul_kernel<gn::EPI_GELU, false, true>;
    hipLaunchKernelGGL(kfn, dim3((unsigned)nblk), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)w,
                       (bf16_t*)gd, (bf16_t*)g, (float*)nullptr, (long)M, N, K, (long)lda, (long)ldw, (long)N, (long)N,
                       bias, drop_thresh(p), drop_scale(p), seed, seed_base, spread_starts());
    return (int)hipGetLastError();
}

}  // extern "C"
