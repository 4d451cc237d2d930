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
#include <stdlib.h>

namespace cwlt {
namespace gn {

constexpr int TMR = 128, TNC = 256, BK = 32;
constexpr int LDK = 40;    // staging row stride (bf16): 80 B
constexpr int LDE = 264;   // epilogue tile row stride (bf16): 528 B

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 frag(const bf16_t* t, int row, int k) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(t + row * LDK + k));
}

// part: (row tiles, N) f32 column sums of this workgroup's rows of C (NULL: not wanted)
template <bool NT_STREAMS>
__global__ __launch_bounds__(512, 2) void gemm_nt_mul_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const bf16_t* __restrict__ G, bf16_t* __restrict__ Cout,
    float* __restrict__ part, long M, int N, int K, long lda, long ldw, long ldg, long ldc) {
    // staging: As[2][128][40] + Ws[2][256][40] bf16 = 61 440 B; the epilogue tile [128][264] bf16 = 67 584 B reuses it
    __shared__ __attribute__((aligned(16))) bf16_t lds[TMR * LDE];
    bf16_t* As0 = lds;
    bf16_t* As1 = lds + TMR * LDK;
    bf16_t* Ws0 = lds + 2 * TMR * LDK;
    bf16_t* Ws1 = lds + 2 * TMR * LDK + TNC * LDK;

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

    const __amdgpu_buffer_rsrc_t ar = make_rsrc(A + m0 * lda, (uint32_t)(((mrows - 1) * lda + K) * 2));
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(W + (long)n0 * ldw, (uint32_t)(((long)(TNC - 1) * ldw + K) * 2));
    const int srow = tid >> 2, sk = (tid & 3) * 8;    // staging slot: row srow (and srow + 128 of W), 8 k values
    uint4 xa0, wa0, wb0, xa1, wa1, wb1;
#define GN_LOAD(S, k0)                                                                        \
    {                                                                                         \
        xa##S = buf_load16(ar, ((uint32_t)srow * (uint32_t)lda + (k0) + sk) * 2);             \
        wa##S = buf_load16(wr, ((uint32_t)srow * (uint32_t)ldw + (k0) + sk) * 2);             \
        wb##S = buf_load16(wr, ((uint32_t)(srow + 128) * (uint32_t)ldw + (k0) + sk) * 2);     \
    }
#define GN_STAGE(S, Ab, Wb)                                                                   \
    {                                                                                         \
        *reinterpret_cast<uint4*>(Ab + srow * LDK + sk) = xa##S;                              \
        *reinterpret_cast<uint4*>(Wb + srow * LDK + sk) = wa##S;                              \
        *reinterpret_cast<uint4*>(Wb + (srow + 128) * LDK + sk) = wb##S;                      \
    }
#define GN_COMPUTE(Ab, Wb)                                                                    \
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

    GN_LOAD(0, 0);
    GN_STAGE(0, As0, Ws0);
    GN_LOAD(0, BK);
    __syncthreads();
    // unrolled by two with static stage names (loads run two steps ahead); K is a multiple of 64 (launcher)
    for (int k0 = 0; k0 < K; k0 += 2 * BK) {
        GN_LOAD(1, k0 + 2 * BK);
        GN_COMPUTE(As0, Ws0);
        GN_STAGE(0, As1, Ws1);
        __syncthreads();
        GN_LOAD(0, k0 + 3 * BK);
        GN_COMPUTE(As1, Ws1);
        GN_STAGE(1, As0, Ws0);
        __syncthreads();
    }
#undef GN_LOAD
#undef GN_STAGE
#undef GN_COMPUTE

    // epilogue.  This thread's 8 chunks of the tile: rows (tid >> 5) + 16 i, columns 8 (tid & 31) .. + 7.  Their gd
    // chunks are requested first (the accumulators are still being written to LDS while they fly).
    const int erow = tid >> 5, ecol = (tid & 31) * 8;
    const __amdgpu_buffer_rsrc_t gr =
        make_rsrc(G + m0 * ldg + n0, (uint32_t)(((mrows - 1) * ldg + TNC) * 2));      // rows >= mrows read zeros
    uint4 gv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // once-read stream: non-temporal (aux = 2) so that it does not evict the operand strips the co-resident
        // workgroup's main loop re-reads from L2 (CWLT_GEMM_NT=0 builds use the default policy for A/B comparison)
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(
            gr, (int)(((uint32_t)(erow + 16 * i) * (uint32_t)ldg + ecol) * 2), 0, NT_STREAMS ? 2 : 0);
        gv[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }

    // f32 product * gd would need the f32 tile in LDS (135 KB); the tile is rounded to bf16 first (as the unfused
    // GEMM's output was) and multiplied in f32, rounded once more: the arithmetic of the two-kernel path.
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
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
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
            u32x4_t* dst = reinterpret_cast<u32x4_t*>(Cout + (m0 + row) * ldc + n0 + ecol);
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
    const long mtiles = (M + gn::TMR - 1) / gn::TMR;
    const long mt8 = (mtiles + 7) / 8 * 8;            // row tiles are dealt to the 8 XCDs: pad to a multiple of 8
    const long nblk = mt8 * (N / gn::TNC);
    static const bool nts = [] { const char* e = getenv("CWLT_GEMM_NT"); return !(e && e[0] == '0'); }();   // A/B switch
    hipLaunchKernelGGL(nts ? gn::gemm_nt_mul_kernel<true> : gn::gemm_nt_mul_kernel<false>, dim3((unsigned)nblk),
                       dim3(512), 0, st, (const bf16_t*)a, (const bf16_t*)w, (const bf16_t*)g, (bf16_t*)c, part,
                       (long)M, N, K, (long)lda, (long)ldw, (long)ldg, (long)ldc);
    int e = (int)hipGetLastError();
    if (e || !colsum) return e;
    return launch_colsum_finalize(part, colsum, (int)mtiles, (long)N, N, 1.0f, 0, st);
}

}  // extern "C"
