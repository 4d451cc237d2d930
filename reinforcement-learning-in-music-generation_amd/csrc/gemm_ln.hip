// bf16 "NT" GEMM whose epilogue is the post-LN residual block of the encoder layer:
//
//   o = bf16(a w^T) + bias;   s = x + dropout(o);   y = LayerNorm(s) * gamma + beta;   mean, rstd per row
//
// -- `x = norm1(x + dropout(attention.out_projection(.)))` and `norm2(x + dropout(linear2(.)))` of fast_transformers'
// TransformerEncoderLayer (post-LN), reached from /root/reference/dqn_policy/model.py:128-137,231-232.  So far: a
// hipBLASLt GEMM that wrote o (R x 512 bf16) followed by cwlt_add_dropout_layernorm_fwd, which read o and x back and wrote
// s and y.  Here o never reaches HBM: 2 of the 6 R x D x 2-byte streams of the pair are gone, and one launch.
//
// A LayerNorm needs whole rows, so a workgroup owns a ROW tile: 128 rows x all N = 512 columns (d_model 512:
// dqn_policy/config.py:11-15; other widths keep the two-kernel path).  16 waves (2 x 8, 64 x 64 each), operands
// K-contiguous through the LDS-DMA ring of gemm_nt.hip (3 stages of (128 + 512) rows x 64 B = 40 KiB, source-side XOR
// swizzle, one barrier per k-step, waits counted by hand); the tile leaves the accumulators as bf16 through LDS (the
// rounding hipBLASLt's output had), then ONE WAVE PER ROW -- 64 lanes x 8 columns -- does what the LayerNorm kernel does:
// bias, dropout lanes keyed by (seed, row * N + column) exactly as cwlt_add_dropout_layernorm_fwd keys them, residual,
// row statistics in f32 by DPP reductions, two 1 KiB coalesced stores per row.
// Bound: HBM.  Algorithmic bytes per launch R x (K + 3 N) x 2 (a, x read; s, y written); the weight (N x K) stays in L2.
#include "cwlt_common.h"
#include <stdlib.h>

namespace cwlt {
namespace gl {

constexpr int TMR = 128, TNC = 512, BK = 32;
constexpr int NSTAGE = 3;
constexpr int STG = (TMR + TNC) * BK * 2;       // 40 KiB per stage; the W rows start at TMR * 64
constexpr int LDE = TNC + 8;                    // epilogue tile row stride (bf16): 1040 B
constexpr int LDS_BYTES = TMR * LDE * 2 > NSTAGE * STG ? TMR * LDE * 2 : NSTAGE * STG;      // 133 120 B

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__global__ __launch_bounds__(1024) void gemm_ln_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    const bf16_t* __restrict__ X, const float* __restrict__ gamma, const float* __restrict__ beta,
    bf16_t* __restrict__ S, bf16_t* __restrict__ Y, float* __restrict__ mean, float* __restrict__ rstd, long M, int K,
    long lda, long ldw, float eps, uint32_t thresh, float keep_scale, uint64_t seed,
    const uint64_t* __restrict__ seed_base) {
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 3, wn = w & 7;                // wave tile: rows 64 wm.., columns 64 wn..
    const int l31 = lane & 31, hf = lane >> 5;
    const long m0 = (long)blockIdx.x * TMR;
    const long mrows = min((long)TMR, M - m0);

    // Descriptors as four SGPRs each; rows past the tile's end read back as zeros (hardware range check).
    const uint64_t abase = (uint64_t)(A + m0 * lda), wbase = (uint64_t)W;
    u32x4_t ars, wrs;
    ars[0] = __builtin_amdgcn_readfirstlane((uint32_t)abase);
    ars[1] = __builtin_amdgcn_readfirstlane((uint32_t)(abase >> 32));
    ars[2] = __builtin_amdgcn_readfirstlane((uint32_t)(((mrows - 1) * lda + K) * 2));
    ars[3] = 0x00020000u;
    wrs[0] = __builtin_amdgcn_readfirstlane((uint32_t)wbase);
    wrs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(wbase >> 32));
    wrs[2] = __builtin_amdgcn_readfirstlane((uint32_t)(((long)(TNC - 1) * ldw + K) * 2));
    wrs[3] = 0x00020000u;
    // DMA pieces of a step (1 KiB = 16 rows each): the 8 a-pieces go to waves 0-7, the 32 w-pieces two to every wave.
    // Lane l lands at piece base + 16 l = (row l >> 2, chunk position l & 3), which holds chunk (l & 3) ^ ((l >> 4) & 3).
    const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);
    const uint32_t a_voff = ((uint32_t)(16 * (w & 7) + (lane >> 2)) * (uint32_t)lda + dchunk * 8) * 2;
    const uint32_t w_voff = ((uint32_t)(32 * w + (lane >> 2)) * (uint32_t)ldw + dchunk * 8) * 2;
    const uint32_t w_16 = (uint32_t)(16 * ldw * 2);
    const uint32_t lds_a = (uint32_t)(uintptr_t)(lds_void*)lds + (w & 7) * 1024;
    const uint32_t lds_w = (uint32_t)(uintptr_t)(lds_void*)lds + TMR * 64 + w * 2048;
    // issued from inline asm so that the waits can be counted by hand (see wgrad.hip); M0 carries the LDS address
#define GL_DMA_W(stage, step)                                                                                     \
    {                                                                                                             \
        unsigned keep;                                                                                            \
        const uint32_t lw = lds_w + (uint32_t)(stage) * STG;                                                      \
        const uint32_t sk_ = (uint32_t)(step) * (BK * 2), sk2 = sk_ + w_16;                                       \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\t"                                     \
                     "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"                                               \
                     "s_add_u32 m0, %2, 0x400\n\ts_nop 0\n\t"                                                    \
                     "buffer_load_dwordx4 %1, %3, %5 offen lds\n\t"                                               \
                     "s_mov_b32 m0, %0"                                                                           \
                     : "=&s"(keep)                                                                                \
                     : "v"(w_voff), "s"(lw), "s"(wrs), "s"(sk_), "s"(sk2)                                         \
                     : "memory", "scc");                                                                          \
    }
#define GL_DMA_A(stage, step)                                                                                     \
    {                                                                                                             \
        unsigned keep;                                                                                            \
        const uint32_t la = lds_a + (uint32_t)(stage) * STG;                                                      \
        const uint32_t sk_ = (uint32_t)(step) * (BK * 2);                                                         \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\t"                                     \
                     "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"                                               \
                     "s_mov_b32 m0, %0"                                                                           \
                     : "=&s"(keep)                                                                                \
                     : "v"(a_voff), "s"(la), "s"(ars), "s"(sk_)                                                   \
                     : "memory", "scc");                                                                          \
    }
#define GL_DMA(stage, step)        \
    {                              \
        if (w < 8) GL_DMA_A(stage, step); \
        GL_DMA_W(stage, step);     \
    }
    // fragment byte offsets of this lane inside a row block: row l31, chunk (2 ks + hf) at position ^ ((l31 >> 2) & 3)
    const int swz = (l31 >> 2) & 3;
    const int of0 = l31 * 64 + ((hf ^ swz) << 4), of1 = l31 * 64 + (((2 + hf) ^ swz) << 4);
    const int oa = (64 * wm) * 64, ow = TMR * 64 + (64 * wn) * 64;      // wave tile bases; + 32 rows = + 2048 bytes
#define GL_FRAG(p) __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p))
    // fragments of one k16 half of a stage: two 32-row blocks of the wave's W rows, two of its a rows
#define GL_READ(F, stage, ofh)                                            \
    {                                                                     \
        const char* sb = lds + (stage) * STG;                             \
        F[0] = GL_FRAG(sb + ow + (ofh));                                  \
        F[1] = GL_FRAG(sb + ow + 2048 + (ofh));                           \
        F[2] = GL_FRAG(sb + oa + (ofh));                                  \
        F[3] = GL_FRAG(sb + oa + 2048 + (ofh));                           \
    }
#define GL_MFMA(F)                                                                          \
    {                                                                                       \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0], F[2], acc[0][0], 0, 0, 0); \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1], F[2], acc[0][1], 0, 0, 0); \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[0], F[3], acc[1][0], 0, 0, 0); \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[1], F[3], acc[1][1], 0, 0, 0); \
    }

    f32x16 acc[2][2];   // [row half i][column half j]: registers = columns (n), lanes = rows (m)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Ring of 3 stages, DMA THREE steps ahead, one barrier per step -- placed in the MIDDLE of a step's MFMAs, with the
    // fragment reads one k16 half ahead of the MFMAs that use them:
    //   read half 1 of step s | MFMA half 0 of step s | wait: own pieces of step s + 1 landed, own reads of stage s done |
    //   barrier | DMA step s + 3 into the stage step s used | read half 0 of step s + 1 | MFMA half 1 of step s
    // so every MFMA group finds its operands in registers, and the LDS reads, the DMA issue and the barrier wait of a
    // wave fall under MFMAs already queued.  Worth 1-3 % here (0.597 -> 0.589 ms at K = 512): the step is not paced by
    // fragment latency but by the operand pieces themselves -- 40 KiB per step from L2 through the CU's one
    // vector-memory queue, ~1 us per step against 0.49 us of MFMA work (HISTORY 4.2c); issuing the pieces after the
    // second MFMA group instead of behind the barrier is slower (0.62 ms).
    // A wave leaves its OWN pieces of the youngest step in flight at the wait: 3 for waves 0-7, 2 for waves 8-15.
    const int nstep = K / BK;                         // K is a multiple of 64 (launcher): nstep >= 2
    bf16x8 F0[4], F1[4];
    GL_DMA(0, 0);
    GL_DMA(1, 1);
    if (nstep > 2) GL_DMA(2, 2);
    if (nstep > 2) {
        if (w < 8)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        if (w < 8)
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    GL_READ(F0, 0, of0);
    int st = 0;                                       // s % NSTAGE
    for (int s = 0; s < nstep; ++s) {
        const int st1 = st == NSTAGE - 1 ? 0 : st + 1;
        GL_READ(F1, st, of1);
        __builtin_amdgcn_sched_barrier(0);
        GL_MFMA(F0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nstep) {
            // own pieces of step s + 1 have landed (the pieces of step s + 2, issued a step ago, may still fly), and every
            // fragment read this wave made of stage `st` has returned
            if (s + 2 < nstep) {
                if (w < 8)
                    asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (s + 3 < nstep) GL_DMA(st, s + 3);
            GL_READ(F0, st1, of0);
            __builtin_amdgcn_sched_barrier(0);
        }
        GL_MFMA(F1);
        __builtin_amdgcn_sched_barrier(0);
        st = st1;
    }
#undef GL_DMA
#undef GL_DMA_A
#undef GL_DMA_W
#undef GL_FRAG
#undef GL_READ
#undef GL_MFMA

    // ---- epilogue.  The tile leaves the accumulators as bf16 through LDS (the ring is done with) ...
    __syncthreads();
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
    // ... then wave w owns rows w, w + 16, ..., w + 112 of the tile, lane l columns 8 l .. 8 l + 7.  The residual rows
    // are requested before the barrier (the accumulators' registers are free now).
    const int ecol = lane * 8;
    uint4 xr[8];
    {
        const __amdgpu_buffer_rsrc_t xrs =
            make_rsrc(X + m0 * TNC, (uint32_t)(mrows * TNC * 2));                    // rows past the end: zeros
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = buf_load16(xrs, ((uint32_t)(w + 16 * i) * TNC + ecol) * 2);
    }
    float gm[8], bt[8], bs[8];
    loadf<8>(gamma + ecol, gm);
    loadf<8>(beta + ecol, bt);
    loadf<8>(bias + ecol, bs);
    __syncthreads();
    if (seed_base) seed += *seed_base;   // device-resident offset: a captured hipGraph draws fresh masks per replay
    const float invN = 1.0f / (float)TNC;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = w + 16 * i;
        if (row >= mrows) break;                      // wave-uniform
        const uint4 hv = *reinterpret_cast<const uint4*>(et + row * LDE + ecol);
        float v[8], r[8];
        load8(reinterpret_cast<const bf16_t*>(&hv), v);
        load8(reinterpret_cast<const bf16_t*>(&xr[i]), r);
        const long off = (m0 + row) * TNC + ecol;     // element index in the dense (M, N) activation
        // cwlt_add_dropout_layernorm_fwd's arithmetic on a = bf16(product) + bias: scale the kept, add the residual
        const uint32_t km = thresh ? dropout_mask<8>(seed, (uint64_t)off, thresh) : 0xffffffffu;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = v[j] + bs[j];
            v[j] = (((km >> j) & 1u) ? a * keep_scale : 0.f) + r[j];
            sum += v[j];
        }
        store8(S + off, v);
        const float mu = wave_sum(sum) * invN;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[j] - mu;
            sq = fmaf(d, d, sq);
        }
        const float rs = rsqrtf(wave_sum(sq) * invN + eps);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[j] - mu) * rs * gm[j] + bt[j];
        store8(Y + off, o);
        if (lane == 0) {
            mean[m0 + row] = mu;
            rstd[m0 + row] = rs;
        }
    }
}

}  // namespace gl
}  // namespace cwlt

extern "C" {

/* s = x + dropout(bf16(a (M, K) . w (N, K)^T) + bias);  y = LayerNorm(s) * gamma + beta  -- the residual block behind
 * `attention.out_projection` and `linear2` of fast_transformers' post-LN TransformerEncoderLayer
 * (/root/reference/dqn_policy/model.py:128-137,231-232) in one kernel: the GEMM's output never reaches HBM.
 * a, w bf16 with row strides lda, ldw (multiples of 8); x (M, N) bf16 dense; bias, gamma, beta (N) f32; s, y (M, N) bf16
 * dense; mean, rstd (M) f32.  N must be 512 (a workgroup owns whole rows); K % 64 == 0; 16-byte aligned pointers;
 * 0 <= p < 1.  Dropout stream, statistics and rounding points are those of a GEMM with bf16 output followed by
 * cwlt_add_dropout_layernorm_fwd (the bias is added in f32 here, where hipBLASLt adds a bf16-rounded one). */
int cwlt_gemm_nt_bias_dropout_add_layernorm(const void* a, const void* w, const float* bias, const void* x,
                                            const float* gamma, const float* beta, void* s, void* y, float* mean,
                                            float* rstd, int64_t M, int N, int K, int64_t lda, int64_t ldw, float eps,
                                            float p, uint64_t seed, const uint64_t* seed_base, void* stream) {
    using namespace cwlt;
    if (M < 0 || N != gl::TNC || K <= 0 || (K % 64) || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (M == 0) return CWLT_OK;
    if (!a || !w || !bias || !x || !gamma || !beta || !s || !y || !mean || !rstd) return CWLT_ERR_ARG;
    if (((lda | ldw) & 7) || lda < K || ldw < K) return CWLT_ERR_ARG;
    if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)x | (uintptr_t)s | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)gamma |
         (uintptr_t)beta) & 15)
        return CWLT_ERR_ARG;
    /* byte offsets inside one row tile / the weight are 32-bit (buffer resources); tile bases are 64-bit */
    if ((int64_t)gl::TMR * lda * 2 >= (1ll << 31) || (int64_t)gl::TNC * ldw * 2 >= (1ll << 31)) return CWLT_ERR_ARG;
    // the dynamic-LDS opt-in (130 KiB) is per device
    static unsigned long long lds_set = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= 64 || !((lds_set >> dev) & 1ull)) {
        const int e = (int)hipFuncSetAttribute((const void*)gl::gemm_ln_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               gl::LDS_BYTES);
        if (e) return e;
        if (dev < 64) lds_set |= 1ull << dev;
    }
    const long mtiles = (M + gl::TMR - 1) / gl::TMR;
    hipLaunchKernelGGL(gl::gemm_ln_kernel, dim3((unsigned)mtiles), dim3(1024), gl::LDS_BYTES, (hipStream_t)stream,
                       (const bf16_t*)a, (const bf16_t*)w, bias, (const bf16_t*)x, gamma, beta, (bf16_t*)s, (bf16_t*)y, mean,
                       rstd, (long)M, K, (long)lda, (long)ldw, eps, drop_thresh(p), drop_scale(p), seed, seed_base);
    return (int)hipGetLastError();
}

}  // extern "C"
