// The two fused FFN GEMMs (gemm_nt.hip's entry points) as ONE persistent, wave-specialised workgroup per CU.
//
//   forward   x = bf16(a w^T) + bias;  g = dropout(gelu(x)),  gd = mask * keep_scale * gelu'(x)        (EPI_GELU)
//   backward  dh = bf16(dy W2t^T) * gd,  db1 = column sums of dh                                        (EPI_MUL)
// -- `self.dropout(self.activation(self.linear1(y)))` and its backward in fast_transformers' TransformerEncoderLayer,
// reached from /root/reference/dqn_policy/model.py:128-137.
//
// Why this shape.  Round 3 took gemm_nt.hip's forward kernel apart at R = 524 288 (profiles/r03_ffn1_ablation.txt):
// main loop alone 1.11 ms, the 4 GiB of output stores alone 0.82 ms (5.2 TB/s), the activation arithmetic 0.21 ms on
// top of either, everything together 2.09 ms -- the SUM.  Nothing overlapped: a workgroup dumps its 128 KB of stores
// in one burst behind its main loop, a CU's vector-memory pipe is one in-order queue, so the co-resident
// workgroup's operand loads sit behind the burst and the two workgroups of a CU fall into step.  Here the overlap is
// built in:
//   * 16 waves: waves 0-7 ("MFMA waves", the 2 x 4 wave tiling and LDS-DMA operand ring of gemm_nt.hip, unchanged)
//     never issue a store, so their counted `s_waitcnt vmcnt` sees operand pieces only; waves 8-15 ("epilogue waves")
//     never touch the ring: they take the PREVIOUS tile from a bf16 LDS copy, do the arithmetic and issue the stores,
//     one sixteenth of the tile per k-step, i.e. 8 KB of stores per CU per step instead of 128 KB per tile end.
//   * one workgroup per CU walks its tiles in a fixed order (persistent): the operand ring runs across tile
//     boundaries (the pieces of the next tile's first two steps are in flight while the last two steps of this one
//     compute), and the accumulators are handed over through LDS once per tile.
//   * every wave of the workgroup meets at the one s_barrier per k-step the ring needs anyway; the epilogue work of a
//     step is sized below the MFMA work of a step, so the barrier is paced by the MFMA waves.
// LDS: ring 3 x 24 KiB + tile 128 x 264 bf16 (66 KiB) + 16 KiB (bias vector / column-sum scratch) = 154 KiB.
#include "cwlt_common.h"
#include "cwlt_gelu.h"
#include <stdlib.h>
#include <stdio.h>

namespace cwlt {
namespace gw {

constexpr int TMR = 128, TNC = 256, BK = 32;
constexpr int NSTAGE = 3;
constexpr int STG = (TMR + TNC) * BK * 2;       // 24 KiB per stage; W rows start at TMR * 64
constexpr int RING = NSTAGE * STG;
constexpr int LDE = 264;                        // tile row stride (bf16): 528 B
constexpr int TILE_B = TMR * LDE * 2;
constexpr int AUX_B = 16384;                    // EPI_GELU: the bias vector (N <= 4096 f32); EPI_MUL: [16][256] f32
constexpr int NPHASE = 16;                      // epilogue phases per tile: 8 row chunks x (A, B)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

enum { EPI_MUL = 0, EPI_GELU = 1 };

struct Tile {
    long m0;
    int n0, mrows;
    long mt;
};

// tile `ti` of this workgroup: workgroups b, b + 8, ... share an XCD (round-robin dispatch; speed only); the row tiles
// mt = xcd (mod 8) belong to that XCD and its workgroups take (row tile, column tile) pairs in row-major order, so the
// column tiles of a row tile run at the same time on one XCD and read the operand strip from its L2.
__device__ __forceinline__ Tile tile_of(int ti, int xcd, int loc, int per_xcd, int nt, long M) {
    const uint32_t q = (uint32_t)loc + (uint32_t)ti * (uint32_t)per_xcd;      // < 2^31 (launcher)
    const uint32_t r = q / (uint32_t)nt;
    Tile t;
    t.mt = (long)r * 8 + xcd;
    t.m0 = t.mt * TMR;
    t.n0 = (int)(q - r * (uint32_t)nt) * TNC;
    t.mrows = (int)min((long)TMR, M - t.m0);
    return t;
}

template <int EPI, bool NT_STREAMS>
__global__ __launch_bounds__(1024) void gemm_ws_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* G, bf16_t* __restrict__ Cout,
    float* __restrict__ part, long M, int N, int K, long lda, long ldw, long ldg, long ldc,
    const float* __restrict__ bias, uint32_t thresh, float keep_scale, uint64_t seed,
    const uint64_t* __restrict__ seed_base, int abl) {
    __shared__ __attribute__((aligned(16))) char lds[RING + TILE_B + AUX_B];
    bf16_t* const tile = reinterpret_cast<bf16_t*>(lds + RING);
    float* const aux = reinterpret_cast<float*>(lds + RING + TILE_B);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = N / TNC, nk = K / BK;
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const long mtiles = (M + TMR - 1) / TMR;
    const long rows_x = (mtiles - xcd + 7) / 8;                 // row tiles of this XCD
    const long tiles_x = rows_x * nt;
    const int ntile = tiles_x > loc ? (int)((tiles_x - loc + per_xcd - 1) / per_xcd) : 0;
    if (ntile == 0) return;

    if (EPI == EPI_GELU) {
        // the whole bias vector, once (the launcher insists on N <= 4096)
        for (int i = tid; i < N; i += 1024) aux[i] = bias[i];
    }
    __syncthreads();            // also drains the bias loads: from here on the MFMA waves count operand pieces only

    if (abl & 64)
        for (int i = (blockIdx.x >> 3) & 15; i > 0; --i) __builtin_amdgcn_s_sleep(32);
    if (w < 8) {
        // ------------------------------------------------------------------ MFMA waves: gemm_nt.hip's main loop
        if (abl & 16) __builtin_amdgcn_s_setprio(3);
        const int wm = w >> 2, wn = w & 3;
        const int l31 = lane & 31, hf = lane >> 5;
        const int drow = 16 * w + (lane >> 2);
        const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);
        const uint32_t a_voff = ((uint32_t)drow * (uint32_t)lda + dchunk * 8) * 2;
        const uint32_t w_voff = ((uint32_t)drow * (uint32_t)ldw + dchunk * 8) * 2;
        const uint32_t w_half = (uint32_t)(128 * ldw * 2);
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)lds + w * 1024;
        u32x4_t ars, wrs;
        ars[3] = wrs[3] = 0x00020000u;
        wrs[2] = __builtin_amdgcn_readfirstlane((uint32_t)(((long)(TNC - 1) * ldw + K) * 2));
#define GW_DESC(ti)                                                                                        \
    {                                                                                                      \
        const Tile t_ = tile_of((ti), xcd, loc, per_xcd, nt, M);                                           \
        const uint64_t ab_ = (uint64_t)(A + t_.m0 * lda), wb_ = (uint64_t)(W + (long)t_.n0 * ldw);         \
        ars[0] = __builtin_amdgcn_readfirstlane((uint32_t)ab_);                                            \
        ars[1] = __builtin_amdgcn_readfirstlane((uint32_t)(ab_ >> 32));                                    \
        ars[2] = __builtin_amdgcn_readfirstlane((uint32_t)(((long)(t_.mrows - 1) * lda + K) * 2));         \
        wrs[0] = __builtin_amdgcn_readfirstlane((uint32_t)wb_);                                            \
        wrs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(wb_ >> 32));                                    \
    }
#define GW_DMA(stage, step)                                                                                       \
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
        const int swz = (l31 >> 2) & 3;
        const int of0 = l31 * 64 + ((hf ^ swz) << 4), of1 = l31 * 64 + (((2 + hf) ^ swz) << 4);
        const int oa = (64 * wm) * 64, ow = TMR * 64 + (64 * wn) * 64;
#define GW_FRAG(p) __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p))
#define GW_COMPUTE(stage)                                                                     \
    {                                                                                         \
        const char* sb = lds + (stage) * STG;                                                 \
        const bf16x8 w00 = GW_FRAG(sb + ow + of0), w01 = GW_FRAG(sb + ow + 2048 + of0);       \
        const bf16x8 x00 = GW_FRAG(sb + oa + of0), x01 = GW_FRAG(sb + oa + 2048 + of0);       \
        const bf16x8 w10 = GW_FRAG(sb + ow + of1), w11 = GW_FRAG(sb + ow + 2048 + of1);       \
        const bf16x8 x10 = GW_FRAG(sb + oa + of1), x11 = GW_FRAG(sb + oa + 2048 + of1);       \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w00, x00, acc[0][0], 0, 0, 0);    \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w01, x00, acc[0][1], 0, 0, 0);    \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w00, x01, acc[1][0], 0, 0, 0);    \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w01, x01, acc[1][1], 0, 0, 0);    \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w10, x10, acc[0][0], 0, 0, 0);    \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w11, x10, acc[0][1], 0, 0, 0);    \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w10, x11, acc[1][0], 0, 0, 0);    \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w11, x11, acc[1][1], 0, 0, 0);    \
    }
        f32x16 acc[2][2];
        // Global step g = ti * nk + s.  Ring of 3 stages, DMA two steps ahead, ONE barrier per step (as in gemm_nt.hip);
        // the descriptors move on to the next tile when the step two ahead is its first.
        const long total = (long)ntile * nk;
        unsigned long long dbg_wait = 0, dbg_bar = 0, dbg_hand = 0;
        const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
        GW_DESC(0);
        if (!(abl & 1)) {
        GW_DMA(0, 0);
        GW_DMA(1, 1);
        }                                   // nk >= 2 (launcher)
        long g = 0;
        int st = 0;                                     // g % NSTAGE
        for (int ti = 0; ti < ntile; ++ti) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            for (int s = 0; s < nk; ++s, ++g) {
                unsigned long long ta = 0, tb = 0, tc = 0;
                if (abl & 128) ta = __builtin_amdgcn_s_memtime();
                if (g + 1 < total)
                    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (abl & 128) tb = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();
                if (abl & 128) { tc = __builtin_amdgcn_s_memtime(); dbg_wait += tb - ta; dbg_bar += tc - tb; }
                __builtin_amdgcn_sched_barrier(0);
                if (abl & 1) continue;
                if (g + 2 < total) {
                    int s2 = s + 2;
                    if (s2 >= nk) {
                        s2 -= nk;
                        if (s2 == 0) GW_DESC(ti + 1);
                    }
                    const int st2 = st == 0 ? 2 : st - 1;       // (g + 2) % 3
                    GW_DMA(st2, s2);
                }
                GW_COMPUTE(st);
                st = st == 2 ? 0 : st + 1;
            }
            // hand-over: every epilogue wave has finished reading the previous tile (its last phase ran in this tile's
            // last step at the latest), every MFMA of this tile is issued
            unsigned long long th0 = 0;
            if (abl & 128) th0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = 64 * wm + 32 * i + l31, c0 = 64 * wn + 32 * j;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16x4 p;
#pragma unroll
                        for (int u = 0; u < 4; ++u) p[u] = (__bf16)acc[i][j][4 * q + u];
                        *reinterpret_cast<uint2*>(tile + row * LDE + c0 + 8 * q + 4 * hf) = __builtin_bit_cast(uint2, p);
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (abl & 128) dbg_hand += __builtin_amdgcn_s_memtime() - th0;
        }
        if ((abl & 128) && part && w == 0 && lane == 0) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(part) + blockIdx.x * 8;
            d[0] = __builtin_amdgcn_s_memtime() - dbg_t0; d[1] = dbg_wait; d[2] = dbg_bar; d[3] = dbg_hand;
        }
        __builtin_amdgcn_s_barrier();                   // F: the last tile is in LDS
        if (EPI == EPI_MUL && part) __builtin_amdgcn_s_barrier();      // D: the epilogue waves' last column-sum exchange
#undef GW_DESC
#undef GW_DMA
#undef GW_FRAG
#undef GW_COMPUTE
        return;
    }

    // ---------------------------------------------------------------------- epilogue waves
    const int et = tid - 512;
    const int erow = et >> 5, ecol = (et & 31) * 8;
    if (seed_base) seed += *seed_base;   // device-resident offset: a captured hipGraph draws fresh masks per replay
    const RngKey rk = rng_key(seed);
    const GeluK gk = gelu_consts(keep_scale);
    const uint32_t thresh2 = thresh | (thresh << 16);

    // state of the chunk in flight between its phase A and its phase B
    float xv[8], ev[8];
    uint32_t keep[4];
    float b[8];
    float cs[8];
    uint4 gv0 = CWLT_U4Z, gv1 = CWLT_U4Z;
    u32x4_t qhold = {0u, 0u, 0u, 0u};
    long qrow = -1;                                     // row of the held gd chunk (< 0: none)
    Tile tp;                                            // the tile being written out (the one before the MFMA waves')
    tp.m0 = 0; tp.n0 = 0; tp.mrows = 0; tp.mt = 0;

    int qn0 = 0;
    auto flush_gd = [&]() {
        if (qrow >= 0) {
            u32x4_t* dst = reinterpret_cast<u32x4_t*>(G + qrow * ldg + qn0 + ecol);
            if (NT_STREAMS)
                __builtin_nontemporal_store(qhold, dst);
            else
                *dst = qhold;
            qrow = -1;
        }
    };
    // phase p of tile tp: chunk i = p >> 1 = rows erow + 16 i, columns ecol .. ecol + 7
    auto phase = [&](int p) {
        if (abl & 8) return;
        const int i = p >> 1;
        const int row = erow + 16 * i;
        if (EPI == EPI_GELU) {
            if ((p & 1) == 0) {
                // A: the previous chunk's gd leaves (one store per wave and step keeps the CU's store traffic even);
                // pre-activation, exponential, dropout lanes of this chunk
                flush_gd();
                const uint4 hv = *reinterpret_cast<const uint4*>(tile + row * LDE + ecol);
                float t[8];
                load8(reinterpret_cast<const bf16_t*>(&hv), t);
                const uint64_t pair0 = ((uint64_t)(tp.m0 + row) * (uint64_t)N + (uint64_t)(tp.n0 + ecol)) >> 1;
                const uint32_t x0 = ((uint32_t)pair0 & 0xffffffu) ^ rng_block_key(rk, pair0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xv[j] = t[j] + b[j];
                    ev[j] = (abl & 2) ? 0.f : gelu_expterm(xv[j]);
                }
                if (!(abl & 2))
#pragma unroll
                for (int j = 0; j < 4; ++j) keep[j] = keep_lanes16(rng_mix(x0 ^ (uint32_t)j, rk.k2), thresh2);
            } else {
                // B: polynomial, value and derivative, stores
                u32x4_t r, q;
                if (abl & 2) {
                    r[0] = __float_as_uint(xv[0]); r[1] = __float_as_uint(xv[1]); r[2] = __float_as_uint(xv[2]); r[3] = __float_as_uint(xv[3]);
                    q[0] = __float_as_uint(xv[4]); q[1] = __float_as_uint(xv[5]); q[2] = __float_as_uint(xv[6]); q[3] = __float_as_uint(xv[7]);
                } else
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    float c0, u0, c1, u1;
                    gelu_cdf(xv[j], ev[j], gk, c0, u0);
                    gelu_cdf(xv[j + 1], ev[j + 1], gk, c1, u1);
                    const float y0 = fmaf(fabsf(xv[j]), u0, gk.c0 * xv[j]);
                    const float y1 = fmaf(fabsf(xv[j + 1]), u1, gk.c0 * xv[j + 1]);
                    const float d0 = fmaf(ev[j] * xv[j], gk.pdfc, c0);
                    const float d1 = fmaf(ev[j + 1] * xv[j + 1], gk.pdfc, c1);
                    r[j >> 1] = f32x2_to_bf16x2(y0, y1) & keep[j >> 1];
                    q[j >> 1] = f32x2_to_bf16x2(d0, d1) & keep[j >> 1];
                }
                if ((abl & 4) && r[0] != 0x12345u) return;
                if (row < tp.mrows) {
                    // g is the next GEMM's operand: default policy; gd waits for the backward: streamed past the caches
                    *reinterpret_cast<u32x4_t*>(Cout + (tp.m0 + row) * ldc + tp.n0 + ecol) = r;
                    qhold = q;
                    qrow = tp.m0 + row;
                    qn0 = tp.n0;
                }
            }
        } else {
            // gd chunks are requested two chunks ahead (A of chunk i asks for chunk i + 1; A of chunk 0 for 0 and 1), so
            // that the wait in B leaves the previous chunk's store in flight
            const __amdgpu_buffer_rsrc_t gr =
                make_rsrc(G + tp.m0 * ldg + tp.n0, (uint32_t)(((long)(tp.mrows - 1) * ldg + TNC) * 2));   // rows past the end: zeros
            if ((p & 1) == 0) {
                if (i == 0) {
                    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(
                        gr, (int)(((uint32_t)row * (uint32_t)ldg + ecol) * 2), 0, NT_STREAMS ? 2 : 0);
                    gv0 = make_uint4(v[0], v[1], v[2], v[3]);
                }
                if (i < 7) {
                    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(
                        gr, (int)(((uint32_t)(row + 16) * (uint32_t)ldg + ecol) * 2), 0, NT_STREAMS ? 2 : 0);
                    if (i & 1)
                        gv0 = make_uint4(v[0], v[1], v[2], v[3]);
                    else
                        gv1 = make_uint4(v[0], v[1], v[2], v[3]);
                }
            } else {
                const uint4 hv = *reinterpret_cast<const uint4*>(tile + row * LDE + ecol);
                float t[8], gg[8];
                load8(reinterpret_cast<const bf16_t*>(&hv), t);
                const uint4 gsel = (i & 1) ? gv1 : gv0;
                load8(reinterpret_cast<const bf16_t*>(&gsel), gg);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    t[j] *= gg[j];               // rows past the end: gd read back as zero -> contributes nothing
                    cs[j] += t[j];
                }
                if (row < tp.mrows) {
                    u32x4_t r;
                    r[0] = f32x2_to_bf16x2(t[0], t[1]);
                    r[1] = f32x2_to_bf16x2(t[2], t[3]);
                    r[2] = f32x2_to_bf16x2(t[4], t[5]);
                    r[3] = f32x2_to_bf16x2(t[6], t[7]);
                    u32x4_t* dst = reinterpret_cast<u32x4_t*>(Cout + (tp.m0 + row) * ldc + tp.n0 + ecol);
                    if (NT_STREAMS)
                        __builtin_nontemporal_store(r, dst);
                    else
                        *dst = r;
                }
            }
        }
    };
    auto begin_tile = [&](int ti) {
        tp = tile_of(ti, xcd, loc, per_xcd, nt, M);
        if (EPI == EPI_GELU) {
            const float4 b0 = *reinterpret_cast<const float4*>(aux + tp.n0 + ecol);
            const float4 b1 = *reinterpret_cast<const float4*>(aux + tp.n0 + ecol + 4);
            b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) cs[j] = 0.f;
        }
    };
    // column sums of the finished tile: 16 threads share a column chunk (erow = 0..15)
    auto colsum_put = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) aux[erow * TNC + ecol + j] = cs[j];
    };
    auto colsum_get = [&]() {
        if (et < TNC) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += aux[r * TNC + et];
            part[tp.mt * N + tp.n0 + et] = s;
        }
    };

    unsigned long long edbg_bar = 0, edbg_work[2] = {0, 0};
    for (int ti = 0; ti < ntile; ++ti) {
        // while the MFMA waves compute tile ti, tile ti - 1 leaves: phases [ceil(16 s / nk), ceil(16 (s + 1) / nk)) in step s
        if (ti > 0) begin_tile(ti - 1);
        int p = 0;
        for (int s = 0; s < nk; ++s) {
            unsigned long long ta = 0, tb = 0;
            if (abl & 128) ta = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if (abl & 128) tb = __builtin_amdgcn_s_memtime();
            if (ti > 0)
                for (; p * nk < NPHASE * (s + 1); ++p) phase(p);        // p < ceil(16 (s + 1) / nk)
            if (abl & 128) { edbg_bar += tb - ta; edbg_work[s & 1] += __builtin_amdgcn_s_memtime() - tb; }
        }
        if (EPI == EPI_GELU) flush_gd();
        if (EPI == EPI_MUL && part && ti > 0) colsum_put();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // hand-over (the MFMA waves now overwrite the tile)
        if (EPI == EPI_MUL && part && ti > 0) colsum_get();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((abl & 128) && part && w == 8 && lane == 0) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(part) + blockIdx.x * 8;
        d[4] = edbg_bar; d[5] = edbg_work[0]; d[6] = edbg_work[1];
    }
    __builtin_amdgcn_s_barrier();                       // F
    begin_tile(ntile - 1);
    for (int p = 0; p < NPHASE; ++p) phase(p);
    if (EPI == EPI_GELU) flush_gd();
    if (EPI == EPI_MUL && part) {
        colsum_put();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // D
        colsum_get();
    }
}

}  // namespace gw
}  // namespace cwlt

namespace cwlt {

static int ws_grid() {
    // one workgroup per CU, a multiple of 8 (workgroups are dealt to the XCDs round-robin)
    static const int v = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8)
                cus = prop.multiProcessorCount / 8 * 8;
        }
        return cus;
    }();
    return v;
}

static int ws_abl() { static const int v = [] { const char* a = getenv("CWLT_GEMM_WS_ABLATE"); return a ? atoi(a) : 0; }(); return v; }

static float* ws_dbg() {
    static float* p = nullptr;
    if (!(ws_abl() & 128)) return nullptr;
    if (!p) { hipMalloc((void**)&p, 256 * 64); hipMemset(p, 0, 256 * 64); }
    return p;
}
extern "C" void cwlt_ws_debug_dump() {
    if (!ws_dbg()) return;
    unsigned long long h[256 * 8];
    hipDeviceSynchronize();
    hipMemcpy(h, ws_dbg(), sizeof(h), hipMemcpyDeviceToHost);
    double a[8] = {0};
    for (int b = 0; b < 256; ++b) for (int i = 0; i < 8; ++i) a[i] += (double)h[b * 8 + i] / 256.0;
    printf("ws debug (mean cycles per WG): mfma total %.0f  vmcnt-wait %.0f  barrier %.0f  handover %.0f | epi barrier %.0f  workA(even) %.0f  workB(odd) %.0f\n",
           a[0], a[1], a[2], a[3], a[4], a[5], a[6]);
}

bool gemm_ws_enabled() {
    static const bool v = [] { const char* e = getenv("CWLT_GEMM_WS"); return !(e && e[0] == '0'); }();
    return v;
}

// a (M, K), w (N, K), g / c (M, N) bf16; see gemm_nt.hip's entry points for the contracts (checked there)
int launch_gemm_ws_mul(const void* a, const void* w, const void* g, void* c, float* part, long M, int N, int K, long lda,
                       long ldw, long ldg, long ldc, bool nt_streams, hipStream_t st) {
    auto kfn = nt_streams ? gw::gemm_ws_kernel<gw::EPI_MUL, true> : gw::gemm_ws_kernel<gw::EPI_MUL, false>;
    hipLaunchKernelGGL(kfn, dim3((unsigned)ws_grid()), dim3(1024), 0, st, (const bf16_t*)a, (const bf16_t*)w,
                       const_cast<bf16_t*>((const bf16_t*)g), (bf16_t*)c, part, M, N, K, lda, ldw, ldg, ldc,
                       (const float*)nullptr, 0u, 1.0f, (uint64_t)0, (const uint64_t*)nullptr, ws_abl());
    return (int)hipGetLastError();
}

int launch_gemm_ws_gelu(const void* a, const void* w, const float* bias, void* g, void* gd, long M, int N, int K,
                        long lda, long ldw, float p, uint64_t seed, const uint64_t* seed_base, bool nt_streams,
                        hipStream_t st) {
    auto kfn = nt_streams ? gw::gemm_ws_kernel<gw::EPI_GELU, true> : gw::gemm_ws_kernel<gw::EPI_GELU, false>;
    hipLaunchKernelGGL(kfn, dim3((unsigned)ws_grid()), dim3(1024), 0, st, (const bf16_t*)a, (const bf16_t*)w,
                       (bf16_t*)gd, (bf16_t*)g, ws_dbg(), M, N, K, lda, ldw, (long)N, (long)N, bias,
                       drop_thresh(p), drop_scale(p), seed, seed_base, ws_abl());
    return (int)hipGetLastError();
}

}  // namespace cwlt
