// FFN inner projection with the activation fused into the GEMM epilogue (bf16 throughput path).
//
//   h = x . W1^T                     (rows, d_ff)   kept for the backward (gelu' needs the pre-activation)
//   g = dropout(gelu(h + b1))        (rows, d_ff)   input of the second FFN projection
// Replaces `linear1` + `F.gelu` + `dropout` of fast_transformers' TransformerEncoderLayer.forward
// (/root/reference/dqn_policy/model.py:128-137) -- so far one hipBLASLt GEMM plus cwlt_bias_gelu_dropout_fwd, a
// separate pass that reads h (1 GiB per layer at B = 256) and writes g.  Here the activation runs on the
// output tile while it is still on chip.  Same arithmetic as the two-kernel path, bit for bit: h is rounded
// to bf16 first and the activation is evaluated on the rounded value, with the same dropout keys (seed,
// element index).
//
// STATUS (round 1): correct (h identical to hipBLASLt's, g identical to cwlt_bias_gelu_dropout_fwd on it) but
// NOT used by the encoder yet.  Measured at M = 262 144, K = 512, N = 2048 (tools/bench_kernels.py ffn1):
//   main loop alone 537 us = 1 023 TFLOP/s (hipBLASLt incl. its C write: 585 us = 945 TFLOP/s),
//   + tile to LDS 560 us, + the two output streams (2 GiB) 1 100 us  vs  585 + 487 = 1 072 us unfused.
// The stores do not overlap the other resident workgroup's main loop (wave priorities and a start-up stagger
// of the second resident workgroup changed nothing).  A persistent one-workgroup-per-CU variant that drained the
// previous tile one 16-byte chunk per pair of k-steps of the current tile (main loop alone: 939 TFLOP/s with
// 8 waves per CU) was also correct but slower still, 1 373 us: the activation + dropout hash cost ~30 VALU
// instructions per element (~250 M wave-instructions per layer, 0.2-0.4 ms of issue time chip-wide) and with two
// waves per SIMD that work does not hide under the MFMAs.  What would make the fusion pay: a cheaper mask
// generator (no quarter-rate 32-bit multiplies) and dedicated epilogue waves.
//
// Workgroup = 8 waves (2 x 4), tile 128 rows x 256 columns, BK = 32.  Both operands are K-contiguous, so
// fragments are plain 16-byte LDS reads (row stride 40 bf16 = 80 B).  The product is taken transposed (W
// rows on registers, x rows on lanes) so that each lane holds runs of 4 consecutive columns of one output
// row: the tile goes to LDS with 8-byte writes and leaves with 16-byte coalesced stores.  Operand rows come
// through buffer resources (hardware range check instead of row guards).  Workgroup ids are dealt so that
// the 8 column tiles of one row tile run on ONE XCD (its L2 serves the 7 re-reads of the x strip).
#include "cwlt_common.h"
#include "cwlt_gelu.h"

namespace cwlt {
namespace fg {

constexpr int TMR = 128, TNC = 256, BK = 32;
constexpr int LDK = 40;    // staging row stride (bf16): 80 B
constexpr int LDE = 264;   // epilogue tile row stride (bf16): 528 B

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ constexpr int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }
__device__ __forceinline__ bf16x8 frag(const bf16_t* t, int row, int k) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(t + row * LDK + k));
}

__global__ __launch_bounds__(512, 2) void ffn1_gelu_fwd_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    bf16_t* __restrict__ Hout, bf16_t* __restrict__ Gout, long M, int N, int K, long ldx, long ldw, long ldh, long ldg,
    uint32_t thresh, float keep_scale, uint64_t seed, const uint64_t* __restrict__ seed_base) {
    // staging: As[2][128][40] + Ws[2][256][40] bf16 = 61 440 B; the epilogue tile [128][264] bf16 = 67 584 B reuses it
    __shared__ __attribute__((aligned(16))) bf16_t lds[TMR * LDE];
    bf16_t* As0 = lds;
    bf16_t* As1 = lds + TMR * LDK;
    bf16_t* Ws0 = lds + 2 * TMR * LDK;
    bf16_t* Ws1 = lds + 2 * TMR * LDK + TNC * LDK;
    if (seed_base) seed += *seed_base;

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
    const long mrows = min((long)TMR, M - m0);

    const __amdgpu_buffer_rsrc_t xr = make_rsrc(X + m0 * ldx, (uint32_t)(((mrows - 1) * ldx + K) * 2));
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(W + (long)n0 * ldw, (uint32_t)(((long)(TNC - 1) * ldw + K) * 2));
    const int srow = tid >> 2, sk = (tid & 3) * 8;    // staging slot: row srow (and srow + 128 of W), 8 k values
    uint4 xa0, wa0, wb0, xa1, wa1, wb1;
#define FG_LOAD(S, k0)                                                                        \
    {                                                                                         \
        xa##S = buf_load16(xr, ((uint32_t)srow * (uint32_t)ldx + (k0) + sk) * 2);             \
        wa##S = buf_load16(wr, ((uint32_t)srow * (uint32_t)ldw + (k0) + sk) * 2);             \
        wb##S = buf_load16(wr, ((uint32_t)(srow + 128) * (uint32_t)ldw + (k0) + sk) * 2);     \
    }
#define FG_STAGE(S, Ab, Wb)                                                                   \
    {                                                                                         \
        *reinterpret_cast<uint4*>(Ab + srow * LDK + sk) = xa##S;                              \
        *reinterpret_cast<uint4*>(Wb + srow * LDK + sk) = wa##S;                              \
        *reinterpret_cast<uint4*>(Wb + (srow + 128) * LDK + sk) = wb##S;                      \
    }
#define FG_COMPUTE(Ab, Wb)                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                  \
        bf16x8 fw[2], fx[2];                                                                  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) fw[j] = frag(Wb, 64 * wn + 32 * j + l31, 16 * ks + 8 * hf); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) fx[i] = frag(Ab, 64 * wm + 32 * i + l31, 16 * ks + 8 * hf); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fx[i], acc[i][j], 0, 0, 0); \
    }

    f32x16 acc[2][2];   // [row half i][column half j]: registers = columns (n), lanes = rows (m)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    FG_LOAD(0, 0);
    FG_STAGE(0, As0, Ws0);
    FG_LOAD(0, BK);
    __syncthreads();
    // unrolled by two with static stage names (loads run two steps ahead); K is a multiple of 64 (launcher)
    for (int k0 = 0; k0 < K; k0 += 2 * BK) {
        FG_LOAD(1, k0 + 2 * BK);
        FG_COMPUTE(As0, Ws0);
        FG_STAGE(0, As1, Ws1);
        __syncthreads();
        FG_LOAD(0, k0 + 3 * BK);
        FG_COMPUTE(As1, Ws1);
        FG_STAGE(1, As0, Ws0);
        __syncthreads();
    }
#undef FG_LOAD
#undef FG_STAGE
#undef FG_COMPUTE

    // epilogue 1: accumulators -> bf16 tile et[row][col] (8-byte writes: 4 consecutive columns of one row per lane)
    bf16_t* et = lds;
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
    // epilogue 2: 16-byte chunks: h out, g = dropout(gelu(h + b)) out
    const int F = N;
#pragma unroll 2
    for (int c = tid; c < TMR * (TNC / 8); c += 512) {
        const int row = c >> 5, col = (c & 31) * 8;
        if (row >= mrows) continue;
        const uint4 hv = *reinterpret_cast<const uint4*>(et + row * LDE + col);
        const long grow = m0 + row;
        *reinterpret_cast<uint4*>(Hout + grow * ldh + n0 + col) = hv;
        float t[8], b[8];
        load8(reinterpret_cast<const bf16_t*>(&hv), t);
        loadf<8>(bias + n0 + col, b);
        const long off = grow * F + n0 + col;
        const uint32_t km = thresh ? dropout_mask<8>(seed, off, thresh) : 0xffffffffu;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            f32x2 x, cdf, pdf;
            x[0] = t[j] + b[j];
            x[1] = t[j + 1] + b[j + 1];
            gelu_parts2(x, cdf, pdf);
            const f32x2 y = x * cdf * keep_scale;
            t[j] = ((km >> j) & 1u) ? y[0] : 0.f;
            t[j + 1] = ((km >> (j + 1)) & 1u) ? y[1] : 0.f;
        }
        store8(Gout + grow * ldg + n0 + col, t);
    }
}

}  // namespace fg
}  // namespace cwlt

extern "C" {

/* h (M, N) = x (M, K) . w (N, K)^T ; g = dropout_p(gelu(h + bias)); bf16 operands and outputs, bias (N) f32.
 * N % 256 == 0, K % 64 == 0, row strides multiples of 8, 16-byte aligned pointers. */
int cwlt_ffn1_gelu_dropout_fwd(const void* x, const void* w, const float* bias, void* h, void* g, int64_t M, int N,
                               int K, int64_t ldx, int64_t ldw, int64_t ldh, int64_t ldg, float p, uint64_t seed,
                               const uint64_t* seed_base, void* stream) {
    using namespace cwlt;
    if (M < 0 || N <= 0 || K <= 0 || (N % fg::TNC) || (K % 64) || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (M == 0) return CWLT_OK;
    if (!x || !w || !bias || !h || !g) return CWLT_ERR_ARG;
    if (((ldx | ldw | ldh | ldg) & 7) || ldx < K || ldw < K || ldh < N || ldg < N) return CWLT_ERR_ARG;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)h | (uintptr_t)g) & 15) return CWLT_ERR_ARG;
    const long mtiles = (M + fg::TMR - 1) / fg::TMR;
    const long mt8 = (mtiles + 7) / 8 * 8;            // row tiles are dealt to the 8 XCDs: pad to a multiple of 8
    const long nblk = mt8 * (N / fg::TNC);
    hipLaunchKernelGGL(fg::ffn1_gelu_fwd_kernel, dim3((unsigned)nblk), dim3(512), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)h, (bf16_t*)g, (long)M, N, K, (long)ldx,
                       (long)ldw, (long)ldh, (long)ldg, drop_thresh(p), drop_scale(p), seed, seed_base);
    return (int)hipGetLastError();
}

}  // extern "C"
