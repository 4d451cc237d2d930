// Weight-gradient GEMM for the dense projections:  C[N1][N2] (f32) = sum_m A[m][n1] * B[m][n2]
// with A = upstream gradient (M, N1) and B = layer input (M, N2), both bf16 ROW-major, M = B*T token rows.
//
// Replaces the K = B*T "NT" GEMMs autograd runs for every nn.Linear weight of the encoder
// (fast_transformers AttentionLayer / TransformerEncoderLayer projections built at
// /root/reference/dqn_policy/model.py:128-137).  hipBLASLt runs these at ~545 TF with 128x128 tiles and is
// memory-bound on operand re-reads; here:
//   * 256 x 256 output tile per workgroup (16 waves as 4 x 4, each wave 64 x 64 = 4 MFMA 32x32x16 tiles),
//     so A is re-read N2/256 times and B N1/256 times only;
//   * the reduction runs over the ROW index of both operands, i.e. both MFMA operands are needed
//     transposed: tiles are staged row-major straight into LDS (LDS-DMA, 512-B rows, XOR-swizzled on the source
//     side) and fragments are fetched with ds_read_b64_tr_b16 -- conflict-free under that swizzle;
//   * the token dimension is split over blockIdx.z so that ~256-512 workgroups exist; every split writes
//     an f32 partial tile and a fixed-order reduce kernel sums them straight into the f32 gradient
//     buffer (deterministic, no atomics, no bf16 rounding of the gradient).
#include "cwlt_common.h"
#include <stdlib.h>

namespace cwlt {
namespace wg {

constexpr int TM = 256, TN = 256, BK = 32;
constexpr int NSTAGE = 4;                 // LDS ring: 4 x (32 x 256 A rows + 32 x 256 B rows) bf16 = 128 KiB
constexpr int ROW = 256;                  // LDS row = 256 bf16 = 512 B, unpadded (LDS-DMA writes 1 KiB = two whole rows)
constexpr int OPB = BK * ROW * 2;         // bytes of one operand stage (16 KiB)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ constexpr int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// Staging goes global -> LDS directly (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write -- the 16-byte
// LDS stores of the register-staged version cost ~13 LDS cycles per wave-instruction and, with the fragment reads,
// kept the LDS pipe busier than the MFMA pipe).  An LDS-DMA wave-instruction writes 1 KiB linearly, so rows are
// unpadded 512 B; bank conflicts of the transposed fragment reads (4 consecutive rows, same 64-byte column block =
// same banks at this stride) are avoided by an XOR swizzle applied on the SOURCE side: the 16-byte chunk c of row r
// is stored at chunk position c ^ ((r & 3) << 2), i.e. 64-byte block b of row r sits at block b ^ (r & 3).
//
// byte offset (inside one operand stage) of the transposed-fragment read of this lane: X[l&31][8h + j] = T[k0 + 8h + j]
// [c0 + (l&31)] for k0 = 0; k0 = 16 adds 16 rows (the swizzle depends on row & 3 only)
__device__ __forceinline__ int tfrag_off(int c0, int lane) {
    const int q = (lane >> 2) & 3, p = lane & 3;
    const int row = 8 * (lane >> 5) + q;
    const int chunk = (c0 >> 3) + 2 * ((lane >> 4) & 1) + (p >> 1);
    return (row * ROW + ((chunk ^ (q << 2)) << 3) + 4 * (p & 1)) * 2;
}
__device__ __forceinline__ bf16x8 tfrag_at(const char* base) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * ROW * 2));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// The kernel proper, as a function of the workgroup's index `id` among the ntile * S workgroups of ONE product: the plain
// kernel passes blockIdx.x, the grouped kernel (several products in one launch) the index inside the product it belongs to.
template <bool EDGE, bool ILV>
__device__ __forceinline__ void wgrad_body(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                           float* __restrict__ part, long M, int N1, int N2, long lda, long ldb,
                                           long mslice, const int id, const int S) {
    extern __shared__ __attribute__((aligned(16))) char lds[];      // NSTAGE x [A stage | B stage]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // 16 waves as 4 x 4, each 64 x 64
    const int wn1 = w >> 2, wn2 = w & 3;
    const int l31 = lane & 31, hf = lane >> 5;
    // XCD-aware work mapping (speed only): workgroup ids are dealt round-robin over the 8 XCDs, so id % 8
    // labels the XCD.  All tiles of one token slice read the same A / B rows; giving a slice's tiles to ONE
    // XCD lets those re-reads hit that XCD's L2 instead of HBM (used when the number of slices S is a multiple of 8).
    const int nt1 = (N1 + TM - 1) / TM, nt2 = (N2 + TN - 1) / TN;   // partial edge tiles allowed (N % 8 == 0)
    const int ntile = nt1 * nt2;
    int slice, tile;
    if (S & 7) {                                    // few token rows -> few slices: plain (slice, tile) order
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
    const bf16_t* Ab = A + (long)t1 * TM;
    const bf16_t* Bb = B + (long)t2 * TN;

    // the slice's rows [m0, m1) of both operands as buffer resources: rows past the slice end read back as zeros
    // (hardware range check), offsets are 32-bit and relative to the slice start
    // An edge tile narrower than 256 columns reads on into the next row (finite data whose products land in output
    // columns that are never stored); the descriptor ends at the operand's last element so nothing past it is read.
    const long nrow = m1 > m0 ? m1 - m0 : 0;
    const long a_lim = min((nrow - 1) * lda + TM, (M - m0 - 1) * lda + (N1 - t1 * TM));
    const long b_lim = min((nrow - 1) * ldb + TN, (M - m0 - 1) * ldb + (N2 - t2 * TN));
    // descriptors as four SGPRs each (built from wave-uniform values only; readfirstlane makes that provable)
    const uint64_t abase = (uint64_t)(Ab + m0 * lda), bbase = (uint64_t)(Bb + m0 * ldb);
    u32x4_t ars, brs;
    ars[0] = __builtin_amdgcn_readfirstlane((uint32_t)abase);
    ars[1] = __builtin_amdgcn_readfirstlane((uint32_t)(abase >> 32));
    ars[2] = __builtin_amdgcn_readfirstlane(nrow ? (uint32_t)(a_lim * 2) : 0u);
    ars[3] = 0x00020000u;
    brs[0] = __builtin_amdgcn_readfirstlane((uint32_t)bbase);
    brs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(bbase >> 32));
    brs[2] = __builtin_amdgcn_readfirstlane(nrow ? (uint32_t)(b_lim * 2) : 0u);
    brs[3] = 0x00020000u;

    // this wave's two DMA pieces per step: rows 2w, 2w + 1 of the A stage and of the B stage.  Lane l lands at
    // piece base + 16 l = (row 2w + (l >> 5), chunk position l & 31), which holds chunk (l & 31) ^ ((row & 3) << 2).
    const int drow = 2 * w + hf;
    const int dchunk = l31 ^ ((drow & 3) << 2);
    const uint32_t a_voff = ((uint32_t)drow * (uint32_t)lda + dchunk * 8) * 2;
    const uint32_t b_voff = ((uint32_t)drow * (uint32_t)ldb + dchunk * 8) * 2;
    const uint32_t a_step = (uint32_t)(BK * lda * 2), b_step = (uint32_t)(BK * ldb * 2);   // bytes per 32-row step
    // The two pieces are issued from inline asm: through the builtin, hipcc (ROCm 7.2) cannot tell which LDS bytes a
    // DMA writes and drains ALL of them (s_waitcnt vmcnt(0)) before the first fragment read of every step, which
    // serialises the ring.  Here the waits are counted by hand (vmcnt(4) below).  M0 carries the LDS address and
    // is compiler-reserved: saved and restored inside the statement; s_nop: SGPR write -> M0 / VMEM-read hazards.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)lds + w * 1024;      // this wave's piece inside a stage
#define WG_DMA(stage, step)                                                                                       \
    {                                                                                                             \
        unsigned keep;                                                                                            \
        const uint32_t la = lds0 + (uint32_t)(stage) * (2 * OPB);                                                 \
        const uint32_t sa = (uint32_t)(step) * a_step, sb_ = (uint32_t)(step) * b_step;                           \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"                                     \
                     "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"                                               \
                     "s_add_u32 m0, %3, 0x4000\n\ts_nop 0\n\t"                                                   \
                     "buffer_load_dwordx4 %5, %6, %7 offen lds\n\t"                                               \
                     "s_mov_b32 m0, %0"                                                                           \
                     : "=&s"(keep)                                                                                \
                     : "v"(a_voff), "s"(ars), "s"(la), "s"(sa), "v"(b_voff), "s"(brs), "s"(sb_)                   \
                     : "memory", "scc");                                                                          \
    }
    // one piece at a time (which = 0: A, 1: B), for issue between the step's MFMAs (see gemm_nt.hip, GN_COMPUTE_DMA)
#define WG_DMA1(stage, step, which)                                                                               \
    {                                                                                                             \
        unsigned keep;                                                                                            \
        const uint32_t la = lds0 + (uint32_t)(stage) * (2 * OPB) + ((which) ? 0x4000u : 0u);                      \
        if (which)                                                                                                \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"                              \
                         "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"                         \
                         : "=&s"(keep)                                                                            \
                         : "v"(b_voff), "s"(brs), "s"(la), "s"((uint32_t)(step) * b_step)                         \
                         : "memory", "scc");                                                                      \
        else                                                                                                      \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"                              \
                         "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"                         \
                         : "=&s"(keep)                                                                            \
                         : "v"(a_voff), "s"(ars), "s"(la), "s"((uint32_t)(step) * a_step)                         \
                         : "memory", "scc");                                                                      \
    }
    // fragment read offsets of this lane inside a stage (k0 = 0; the second k-step adds 16 rows)
    const int oa0 = tfrag_off(64 * wn1, lane), oa1 = tfrag_off(64 * wn1 + 32, lane);
    const int ob0 = OPB + tfrag_off(64 * wn2, lane), ob1 = OPB + tfrag_off(64 * wn2 + 32, lane);
    // all 16 fragment reads of the step are issued first (both k-steps, distinct registers), so the second k-step's
    // LDS latency hides behind the first one's MFMAs instead of being waited for after them
#define WG_COMPUTE(stage)                                                                  \
    {                                                                                      \
        const char* sb = lds + (stage) * (2 * OPB);                                        \
        constexpr int ko = 16 * ROW * 2;                                                   \
        const bf16x8 a0 = tfrag_at(sb + oa0), b0 = tfrag_at(sb + ob0);                     \
        const bf16x8 a1 = tfrag_at(sb + oa1), b1 = tfrag_at(sb + ob1);                     \
        const bf16x8 c0_ = tfrag_at(sb + oa0 + ko), d0_ = tfrag_at(sb + ob0 + ko);         \
        const bf16x8 c1_ = tfrag_at(sb + oa1 + ko), d1_ = tfrag_at(sb + ob1 + ko);         \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);   \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);   \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);   \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);   \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0_, d0_, acc[0][0], 0, 0, 0); \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0_, d1_, acc[0][1], 0, 0, 0); \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1_, d0_, acc[1][0], 0, 0, 0); \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1_, d1_, acc[1][1], 0, 0, 0); \
    }

#define WG_COMPUTE_DMA(stage, nstage, nstepi)                                              \
    {                                                                                      \
        const char* sb = lds + (stage) * (2 * OPB);                                        \
        constexpr int ko = 16 * ROW * 2;                                                   \
        const bf16x8 a0 = tfrag_at(sb + oa0), b0 = tfrag_at(sb + ob0);                     \
        const bf16x8 a1 = tfrag_at(sb + oa1), b1 = tfrag_at(sb + ob1);                     \
        const bf16x8 c0_ = tfrag_at(sb + oa0 + ko), d0_ = tfrag_at(sb + ob0 + ko);         \
        const bf16x8 c1_ = tfrag_at(sb + oa1 + ko), d1_ = tfrag_at(sb + ob1 + ko);         \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);   \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);   \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        WG_DMA1(nstage, nstepi, 0);                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);   \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);   \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0_, d0_, acc[0][0], 0, 0, 0); \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0_, d1_, acc[0][1], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        WG_DMA1(nstage, nstepi, 1);                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1_, d0_, acc[1][0], 0, 0, 0); \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1_, d1_, acc[1][1], 0, 0, 0); \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Ring of NSTAGE = 4 stages, DMA running 3 steps ahead.  Per step ONE barrier:
    //   wait until this wave's pieces of step s have landed (vmcnt leaves the 2 younger steps = 4 pieces in flight);
    //   barrier: every wave's pieces of step s have landed, and every wave has finished reading step s - 1;
    //   issue step s + 3 into the stage step s - 1 used;  compute step s.
    // Steps past the slice end load zeros (range check), so the tail needs no special case: the slice is padded to
    // a whole number of steps and the last three issued steps are never computed.
    const int nstep = (int)((mslice + BK - 1) / BK);
    WG_DMA(0, 0);
    WG_DMA(1, 1);
    WG_DMA(2, 2);
    for (int s = 0; s < nstep; ++s) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (ILV) {
            WG_COMPUTE_DMA(s & 3, (s + 3) & 3, s + 3);
        } else {
            WG_DMA((s + 3) & 3, s + 3);
            WG_COMPUTE(s & 3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // pieces still in flight target LDS: drain before exit
#undef WG_DMA
#undef WG_DMA1
#undef WG_COMPUTE_DMA
#undef WG_COMPUTE
    const int r0 = t1 * TM + 64 * wn1, c0 = t2 * TN + 64 * wn2 + l31;
    float* pb = part + ((long)slice * N1 + r0) * N2 + c0;
    if (!EDGE || (r0 + 64 <= N1 && t2 * TN + 64 * wn2 + 64 <= N2)) {          // interior wave tile: unguarded stores
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) pb[(long)(32 * i + acc_row(r, hf)) * N2 + 32 * j] = acc[i][j][r];
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r0 + 32 * i + acc_row(r, hf) < N1 && c0 + 32 * j < N2)
                        pb[(long)(32 * i + acc_row(r, hf)) * N2 + 32 * j] = acc[i][j][r];
    }
}

template <bool EDGE, bool ILV = true>
__global__ __launch_bounds__(1024) void wgrad_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                      float* __restrict__ part, long M, int N1, int N2, long lda,
                                                      long ldb, long mslice) {
    const int ntile = ((N1 + TM - 1) / TM) * ((N2 + TN - 1) / TN);
    wgrad_body<EDGE, ILV>(A, B, part, M, N1, N2, lda, ldb, mslice, (int)blockIdx.x,
                          (int)(gridDim.x / ntile));       // the launcher's grid is exactly ntile * S workgroups
}

// Up to four products over the SAME token rows in one launch (the four weight gradients of an encoder layer at few token
// rows: 20-80 workgroups each, 240 together): workgroup ids [start[p], start[p + 1]) belong to product p.  Widths are
// multiples of 256 (no edge tiles).
struct WgGroup {
    const bf16_t* a[4];
    const bf16_t* b[4];
    float* part[4];
    float* out[4];
    int n1[4], n2[4], S[4];
    long lda[4], ldb[4], mslice[4];
    int start[5];               // workgroup ranges of the products; start[count] = grid size
    long rstart[5];             // float4 ranges of the reduce launch
    int count;
};
template <bool ILV>
__global__ __launch_bounds__(1024) void wgrad_group_kernel(const WgGroup g, long M) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < g.count && (int)blockIdx.x >= g.start[i]) p = i;
    wgrad_body<false, ILV>(g.a[p], g.b[p], g.part[p], M, g.n1[p], g.n2[p], g.lda[p], g.ldb[p], g.mslice[p],
                           (int)blockIdx.x - g.start[p], g.S[p]);
}
// the reduce of every product of a group: thread t of the launch owns float4 t - rstart[p] of product p
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(const WgGroup g, int accumulate) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    int p = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < g.count && t >= g.rstart[i]) p = i;
    const long n = (long)g.n1[p] * g.n2[p];
    const long i = (t - g.rstart[p]) * 4;
    if (t >= g.rstart[g.count] || i >= n) return;
    const float* part = g.part[p];
    float4 a = load4(part + i);
    for (int s = 1; s < g.S[p]; ++s) {
        const float4 v = load4(part + (long)s * n + i);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    float* out = g.out[p];
    if (accumulate) {
        const float4 o = load4(out + i);
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    }
    store4(out + i, a);
}

// out[e] (+)= sum_s part[s * n + e], 4 floats per thread, fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           int S, long n, int accumulate) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float4 a = load4(part + i);
    for (int s = 1; s < S; ++s) {
        const float4 t = load4(part + (long)s * n + i);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    if (accumulate) {
        const float4 o = load4(out + i);
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    }
    store4(out + i, a);
}

}  // namespace wg
}  // namespace cwlt

extern "C" {

/* number of token-dimension splits (= partial tiles per output tile) the kernel will use */
int cwlt_wgrad_splits(int64_t M, int N1, int N2) {
    if (M <= 0 || N1 <= 0 || N2 <= 0) return 1;
    const long tiles = (long)((N1 + 255) / 256) * ((N2 + 255) / 256);
    long s = tiles > 0 ? 256 / tiles : 8;       // about one 16-wave workgroup per CU ...
    s = s / 8 * 8;                              // ... in multiples of 8 (one slice group per XCD)
    if (s < 8) s = 8;
    if (s > 128) s = 128;
    // few token rows (the RL updates: 1 500): a slice below ~256 rows is mostly padding, and every extra slice is
    // another N1 x N2 f32 partial tile for the reduce kernel to read
    const long by_rows = M / 256 > 0 ? M / 256 : 1;
    if (s > by_rows) s = by_rows;
    return (int)s;
}

/* out (N1, N2) f32 dense (+)= A^T B;  a (M, N1), b (M, N2) bf16 row-major with row strides lda, ldb;
 * N1, N2 multiples of 8 (edge tiles of the 256 x 256 tiling may be partial); part: cwlt_wgrad_splits(M, N1, N2) * N1 * N2 floats. */
int cwlt_wgrad_bf16(const void* a, const void* b, float* part, float* out, int64_t M, int N1, int N2, int64_t lda,
                    int64_t ldb, int accumulate, void* stream) {
    using namespace cwlt;
    if (!a || !b || !part || !out || M <= 0) return CWLT_ERR_ARG;
    if (N1 <= 0 || N2 <= 0 || (N1 & 7) || (N2 & 7) || (lda & 7) || (ldb & 7) || lda < N1 || ldb < N2)
        return CWLT_ERR_ARG;
    const int S = cwlt_wgrad_splits(M, N1, N2);
    if (S <= 0) return CWLT_ERR_ARG;
    long mslice = (M + S - 1) / S;
    mslice = (mslice + wg::BK - 1) / wg::BK * wg::BK;                   // whole 32-row steps
    hipStream_t st = (hipStream_t)stream;
    // CWLT_WGRAD_V2=1 (both widths multiples of 256, slices of at least 4 K-tiles): the 8-wave form on gemm_bf16.hip's main
    // loop (wgrad2.hip).  Measured equal to this file's 16-wave kernel -- 945-958 / 918-924 / 284-287 / 807-811 us against
    // 970-985 / 919-922 / 288-292 / 803-806 on the four layer shapes at R = 524 288 (profiles/r04_wgrad2.txt): both run at
    // the 1.15-1.2 PFLOP/s the chip holds under this load (1.68 GHz), so it stays a switch.
    static const bool v2 = [] { const char* e = getenv("CWLT_WGRAD_V2"); return e && e[0] == '1'; }();
    if (v2 && !(N1 & 255) && !(N2 & 255)) {
        const long ms2 = (mslice + 63) / 64 * 64;
        if (ms2 >= 256 && (int64_t)ms2 * (lda > ldb ? lda : ldb) * 2 < (1ll << 31)) {
            int e = launch_wgrad2(a, b, part, (long)M, N1, N2, (long)lda, (long)ldb, ms2, S, st);
            if (e) return e;
            const long n = (long)N1 * N2;
            hipLaunchKernelGGL(wg::wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, part, out,
                               S, n, accumulate);
            return (int)hipGetLastError();
        }
    }
    const bool edge = (N1 & 255) || (N2 & 255);
    constexpr int lds_bytes = wg::NSTAGE * 2 * wg::OPB;                 // 128 KiB: above the 64 KiB default limit
    // the opt-in is per device: remember which devices have it (a process may launch on several)
    static unsigned long long lds_set = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= 64 || !((lds_set >> dev) & 1ull)) {
        // all four instantiations (ILV defaults to true: <true> is <true, true>) -- CWLT_WGRAD_ILV=0 launches the other two
        const void* kfns[4] = {(const void*)wg::wgrad_kernel<true, true>, (const void*)wg::wgrad_kernel<false, true>,
                               (const void*)wg::wgrad_kernel<true, false>, (const void*)wg::wgrad_kernel<false, false>};
        int e = 0;
        for (int i = 0; i < 4 && !e; ++i)
            e = (int)hipFuncSetAttribute(kfns[i], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e) return e;
        if (dev < 64) lds_set |= 1ull << dev;
    }
    // the step's two pieces issued between its MFMAs instead of both behind the barrier (gemm_nt.hip, GN_COMPUTE_DMA):
    // 1 009 / 956 / 291 / 847 -> 981 / 952 / 284 / 822 us on the four layer shapes, same box; CWLT_WGRAD_ILV=0: block issue
    static const bool ilv = [] { const char* e = getenv("CWLT_WGRAD_ILV"); return !(e && e[0] == '0'); }();
    auto kfn = ilv ? (edge ? wg::wgrad_kernel<true, true> : wg::wgrad_kernel<false, true>)
                   : (edge ? wg::wgrad_kernel<true, false> : wg::wgrad_kernel<false, false>);
    hipLaunchKernelGGL(kfn,
                       dim3(((N1 + 255) / 256) * ((N2 + 255) / 256) * S), dim3(1024), lds_bytes, st, (const bf16_t*)a,
                       (const bf16_t*)b, part, (long)M, N1, N2, (long)lda, (long)ldb, mslice);
    int e = (int)hipGetLastError();
    if (e) return e;
    const long n = (long)N1 * N2;
    hipLaunchKernelGGL(wg::wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, part, out, S,
                       n, accumulate);
    return (int)hipGetLastError();
}

/* count (1..4) products out_p (n1_p, n2_p) f32 (+)= A_p^T B_p over the SAME M token rows in ONE launch + one reduce launch
 * (see include/cwlt.h).  All arrays are HOST arrays of `count` entries; widths multiples of 256; part_p:
 * cwlt_wgrad_splits(M, n1_p, n2_p) * n1_p * n2_p floats each, distinct buffers. */
int cwlt_wgrad_bf16_group(const void* const* a, const void* const* b, float* const* part, float* const* out, const int* n1,
                          const int* n2, const int64_t* lda, const int64_t* ldb, int count, int64_t M, int accumulate,
                          void* stream) {
    using namespace cwlt;
    if (count < 1 || count > 4 || !a || !b || !part || !out || !n1 || !n2 || !lda || !ldb || M <= 0) return CWLT_ERR_ARG;
    wg::WgGroup g;
    g.count = count;
    g.start[0] = 0;
    g.rstart[0] = 0;
    for (int p = 0; p < 4; ++p) {
        const int q = p < count ? p : 0;                 // unused slots repeat product 0 (never selected)
        if (!a[q] || !b[q] || !part[q] || !out[q]) return CWLT_ERR_ARG;
        if (n1[q] <= 0 || n2[q] <= 0 || (n1[q] & 255) || (n2[q] & 255) || (lda[q] & 7) || (ldb[q] & 7) || lda[q] < n1[q] ||
            ldb[q] < n2[q])
            return CWLT_ERR_ARG;
        g.a[p] = (const bf16_t*)a[q];
        g.b[p] = (const bf16_t*)b[q];
        g.part[p] = part[q];
        g.out[p] = out[q];
        g.n1[p] = n1[q];
        g.n2[p] = n2[q];
        g.lda[p] = (long)lda[q];
        g.ldb[p] = (long)ldb[q];
        g.S[p] = cwlt_wgrad_splits(M, n1[q], n2[q]);
        // every product cuts the rows as its own launch would (cwlt_wgrad_bf16): whole 32-row steps
        long ms = (M + g.S[p] - 1) / g.S[p];
        g.mslice[p] = (ms + wg::BK - 1) / wg::BK * wg::BK;
        if (p < count) {
            g.start[p + 1] = g.start[p] + (n1[q] / 256) * (n2[q] / 256) * g.S[p];
            g.rstart[p + 1] = g.rstart[p] + ((long)n1[q] * n2[q] / 4 + 255) / 256 * 256;
        }
    }
    for (int p = count; p < 4; ++p) {
        g.start[p + 1] = g.start[count];
        g.rstart[p + 1] = g.rstart[count];
    }
    constexpr int lds_bytes = wg::NSTAGE * 2 * wg::OPB;
    static unsigned long long lds_set = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= 64 || !((lds_set >> dev) & 1ull)) {
        int e = (int)hipFuncSetAttribute((const void*)wg::wgrad_group_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         lds_bytes);
        if (!e)
            e = (int)hipFuncSetAttribute((const void*)wg::wgrad_group_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e) return e;
        if (dev < 64) lds_set |= 1ull << dev;
    }
    static const bool ilv = [] { const char* e = getenv("CWLT_WGRAD_ILV"); return !(e && e[0] == '0'); }();
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ilv ? wg::wgrad_group_kernel<true> : wg::wgrad_group_kernel<false>, dim3((unsigned)g.start[count]),
                       dim3(1024), lds_bytes, st, g, (long)M);
    int e = (int)hipGetLastError();
    if (e) return e;
    hipLaunchKernelGGL(wg::wgrad_reduce_group_kernel, dim3((unsigned)(g.rstart[count] / 256)), dim3(256), 0, st, g, accumulate);
    return (int)hipGetLastError();
}

}  // extern "C"
