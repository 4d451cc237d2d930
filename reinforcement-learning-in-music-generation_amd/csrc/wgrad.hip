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
//     transposed: tiles are staged row-major (16-B coalesced loads, 576-B LDS rows) and fragments are
//     fetched with ds_read_b64_tr_b16 -- conflict-free at this row stride;
//   * the token dimension is split over blockIdx.z so that ~256-512 workgroups exist; every split writes
//     an f32 partial tile and a fixed-order reduce kernel sums them straight into the f32 gradient
//     buffer (deterministic, no atomics, no bf16 rounding of the gradient).
#include "cwlt_common.h"

namespace cwlt {
namespace wg {

constexpr int TM = 256, TN = 256, BK = 32;
constexpr int LDW = 288;  // LDS row stride in bf16 (576 B = 144 banks = 16 mod 64)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ constexpr int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// X[l&31][8h + j] = T[k0 + 8h + j][c0 + (l&31)], T row-major with stride LDW
__device__ __forceinline__ bf16x8 tfrag(const bf16_t* t, int k0, int c0, int lane) {
    const int q = (lane >> 2) & 3, p = lane & 3;
    const bf16_t* base = t + (k0 + 8 * (lane >> 5) + q) * LDW + c0 + 16 * ((lane >> 4) & 1) + 4 * p;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * LDW));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <bool EDGE>
__global__ __launch_bounds__(1024) void wgrad_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                      float* __restrict__ part, long M, int N1, int N2, long lda,
                                                      long ldb, long mslice) {
    __shared__ __attribute__((aligned(16))) bf16_t As[2][BK * LDW];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[2][BK * LDW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // 16 waves as 4 x 4, each 64 x 64
    const int wn1 = w >> 2, wn2 = w & 3;
    const int l31 = lane & 31, hf = lane >> 5;
    // XCD-aware work mapping (speed only): workgroup ids are dealt round-robin over the 8 XCDs, so id % 8
    // labels the XCD.  All tiles of one token slice read the same A / B rows; giving a slice's tiles to ONE
    // XCD lets those re-reads hit that XCD's L2 instead of HBM (used when the number of slices S is a multiple of 8).
    const int nt1 = (N1 + TM - 1) / TM, nt2 = (N2 + TN - 1) / TN;   // partial edge tiles allowed (N % 8 == 0)
    const int ntile = nt1 * nt2;
    const int id = blockIdx.x;
    const int S = gridDim.x / ntile;                // the launcher's grid is exactly ntile * S workgroups
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

    const int srow = tid >> 5, scol = (tid & 31) * 8;   // one 16-B slot of the 32 x 256 stage per thread
    uint4 ra0, rb0, ra1, rb1;   // two register stages: global loads run two steps ahead

    // the slice's rows [m0, m1) of both operands as buffer resources: rows past the slice end read back as zeros
    // (hardware range check), offsets are 32-bit and relative to the slice start
    // An edge tile narrower than 256 columns reads on into the next row (finite data whose products land in output
    // columns that are never stored); the descriptor ends at the operand's last element so nothing past it is read.
    const long nrow = m1 > m0 ? m1 - m0 : 0;
    const long a_lim = min((nrow - 1) * lda + TM, (M - m0 - 1) * lda + (N1 - t1 * TM));
    const long b_lim = min((nrow - 1) * ldb + TN, (M - m0 - 1) * ldb + (N2 - t2 * TN));
    const __amdgpu_buffer_rsrc_t ar = make_rsrc(Ab + m0 * lda, nrow ? (uint32_t)(a_lim * 2) : 0u);
    const __amdgpu_buffer_rsrc_t br = make_rsrc(Bb + m0 * ldb, nrow ? (uint32_t)(b_lim * 2) : 0u);
#define WG_LOAD(RA, RB, ms)                                                                \
    {                                                                                      \
        const uint32_t row = (uint32_t)((ms) - m0) + srow;                                 \
        RA = buf_load16(ar, (row * (uint32_t)lda + scol) * 2);                             \
        RB = buf_load16(br, (row * (uint32_t)ldb + scol) * 2);                             \
    }
#define WG_STAGE(RA, RB, buf)                                                              \
    {                                                                                      \
        *reinterpret_cast<uint4*>(&As[buf][srow * LDW + scol]) = RA;                       \
        *reinterpret_cast<uint4*>(&Bs[buf][srow * LDW + scol]) = RB;                       \
    }
#define WG_COMPUTE(buf)                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                               \
        bf16x8 a[2], b[2];                                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) a[i] = tfrag(As[buf], 16 * ks, 64 * wn1 + 32 * i, lane); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) b[j] = tfrag(Bs[buf], 16 * ks, 64 * wn2 + 32 * j, lane); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                      \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0); \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: step 0 -> LDS[0]; step 1 in register stage 0
    WG_LOAD(ra0, rb0, m0);
    WG_STAGE(ra0, rb0, 0);
    WG_LOAD(ra0, rb0, m0 + BK);
    __syncthreads();
    // steady state, unrolled by two so both register stages keep static names (a register rotation would
    // make every iteration wait for its NEWEST loads); slices are padded to an even number of steps
    // (rows past the slice end load zeros).
    //   even step: compute LDS[0]; loads for step+2 -> stage 1; stage 0 (step+1) -> LDS[1]
    //   odd  step: compute LDS[1]; loads for step+2 -> stage 0; stage 1 (step+1) -> LDS[0]
    for (long ms = m0; ms < m1; ms += 2 * BK) {
        WG_LOAD(ra1, rb1, ms + 2 * BK);
        WG_COMPUTE(0);
        WG_STAGE(ra0, rb0, 1);
        __syncthreads();
        WG_LOAD(ra0, rb0, ms + 3 * BK);
        WG_COMPUTE(1);
        WG_STAGE(ra1, rb1, 0);
        __syncthreads();
    }
#undef WG_LOAD
#undef WG_STAGE
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
    mslice = (mslice + 2 * wg::BK - 1) / (2 * wg::BK) * (2 * wg::BK);   // even number of BK steps
    hipStream_t st = (hipStream_t)stream;
    const bool edge = (N1 & 255) || (N2 & 255);
    hipLaunchKernelGGL(edge ? wg::wgrad_kernel<true> : wg::wgrad_kernel<false>, dim3(((N1 + 255) / 256) * ((N2 + 255) / 256) * S), dim3(1024), 0, st, (const bf16_t*)a,
                       (const bf16_t*)b, part, (long)M, N1, N2, (long)lda, (long)ldb, mslice);
    int e = (int)hipGetLastError();
    if (e) return e;
    const long n = (long)N1 * N2;
    hipLaunchKernelGGL(wg::wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, part, out, S,
                       n, accumulate);
    return (int)hipGetLastError();
}

}  // extern "C"
