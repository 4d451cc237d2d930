// Causal linear attention, bf16-storage throughput path for gfx950 (forward + backward).
//
// Same mathematics and C-ABI as cla.hip (which remains the f32 / parity path); this file is what
// CWLT_BF16 tensors run through.  Replaces fast_transformers' CausalLinearAttention + causal_dot_product
// (reference call sites: dqn_policy/model.py:128-137,231-232).
//
// Why a second implementation: with bf16 activations the f32 MFMA (64 cycles per 32x32x2) made the
// scan compute-bound at ~4x its HBM time.  Here every product is v_mfma_f32_32x32x16_bf16 (16x the
// rate).  Accuracy is kept at the level of the bf16 output rounding:
//   * running states (64x64 KV state and the reverse states) accumulate in f32 MFMA accumulators for
//     the whole sequence and are fed back as operands rounded to bf16 (optionally split hi + lo,
//     CWLT_STATE_LO in cwlt_mfma_bf16.h -- measured to make no difference to the error);
//   * the intra-chunk score tile is rounded to bf16 once and the SAME rounded tile feeds both the
//     numerator and the normaliser, so each output row stays an exactly normalised combination;
//   * forward / dQ outputs leave through an f32 LDS tile and are rounded to bf16 once.
//
// Layout: workgroup = 4 waves = one (n, h) stream, chunk = 64 tokens; wave (wi, wj) owns the 32x32
// tile (wi, wj) of every 64x64 product.  Operand tiles live in LDS ROW-MAJOR only, bf16 [64][72]
// (144-B rows: conflict-free 16-B row-type fragment reads); operands that must be contracted over
// the token index are fetched transposed by ds_read_b64_tr_b16.  Every product is evaluated in the
// orientation whose accumulator layout (rows on registers) lets the tile be written 8-16 B per lane.
// The normaliser and all rank-1 terms (phi(q).ksum, dden*ksum, r1) ride on the MFMA pipe as a 65th
// "ones" value column built from constant register fragments -- no cross-lane reductions.
#include "cwlt_common.h"
#include "cwlt_mfma_bf16.h"

namespace cwlt {
namespace b16 {

// elu(x)+1 with the hardware exp (v_exp_f32, ~1 ulp): far below bf16 resolution, ~12 VALU
// instructions per element cheaper than expf -- this path is issue-bound, not HBM-bound.
__device__ __forceinline__ float phi(float x) { return x > 0.f ? x + 1.f : __expf(x); }
__device__ __forceinline__ float dphi(float x) { return x > 0.f ? 1.f : __expf(x); }
constexpr int FIN_FLOATS = 65 * 64;   // per stream: the forward's final state, see cla_fwd_bf16_kernel
// ------------------------------------------------------------------------------------------------
// forward.  wave (wi, wj): score tile (i-half wi, j-half wj), numerator tile (i-half wi, m-half wj) and
// half of the normaliser:  den_i = sum_j A~[i][j] * 1 + phi(q_i) . ksum,  ksum = state of the ones column.
// ------------------------------------------------------------------------------------------------
// Few streams (N * H below the CU count): the sequence is cut into P segments of `cps` chunks, one workgroup per
// (stream, segment).  STATE_ONLY workgroups first reduce their segment to its state increment (sum of phi(k)^T v
// and of phi(k)); seg_prefix_kernel turns the increments into the state each segment starts from; the scan proper
// then runs every segment from that state.  P == 1: one workgroup per stream, no state traffic (the default).
// State tiles travel in accumulator-register order ([tile][register][lane] f32), so loading one is 16 coalesced loads.
// Workgroup id -> (sequence, head) stream.  Workgroup ids are dealt round-robin over the 8 XCDs, and a token row's H
// heads are H adjacent 128-byte pieces of one contiguous row: with the plain order the 8 heads of a sequence run on 8
// different XCDs, each pulling its 128 bytes out of every row at its own time.  Dealing whole sequences to XCDs (all
// heads of sequence n are neighbours in ONE XCD's dispatch order, so they walk the same rows at about the same time)
// lets the row's DRAM page and L2 lines be shared.  N % 8 != 0: plain order.  (Measured on one box, B = 512: forward
// 511 -> 498 us, one-sweep backward within the run-to-run noise.)
__device__ __forceinline__ int stream_of_block(int b, int N, int H) {
    if (N & 7) return b;
    const int x = b & 7, j = b >> 3;
    return ((j / H) * 8 + x) * H + j % H;
}

__device__ __forceinline__ f32x16 load_tile(const float* t, int lane) {
    f32x16 x;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = t[r * 64 + lane];
    return x;
}
__device__ __forceinline__ void store_tile(float* t, int lane, const f32x16& x) {
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r * 64 + lane] = x[r];
}

template <bool STATE_ONLY>
__global__ __launch_bounds__(256) void cla_fwd_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, bf16_t* __restrict__ o,
                                                           float* __restrict__ zinv, int H, int L, long ldq, long ldk,
                                                           long ldv, long ldo, float eps, int P, int cps,
                                                           const float* __restrict__ pre, float* __restrict__ part,
                                                           float* __restrict__ fin) {
    __shared__ __attribute__((aligned(16))) bf16_t qs[C * LD];   // phi(q)  [i][e]
    __shared__ __attribute__((aligned(16))) bf16_t ks[C * LD];   // phi(k)  [j][e]
    __shared__ __attribute__((aligned(16))) bf16_t vs[C * LD];   // v       [j][m]
    __shared__ __attribute__((aligned(16))) bf16_t as_[C * LD];  // masked scores [i][j]
    __shared__ __attribute__((aligned(16))) float os[C * LDO];   // un-normalised output tile [i][m]
    __shared__ float dens[2][C];                                 // normaliser halves

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = w >> 1, wj = w & 1;
    const int l31 = lane & 31, hf = lane >> 5;
    const int seg = blockIdx.x % P;                             // segment
    const int sid = P == 1 ? stream_of_block(blockIdx.x, gridDim.x / H, H) : blockIdx.x / P;   // stream
    const int n = sid / H, h = sid % H;
    const bf16_t* qb = q + ((long)n * L) * ldq + h * D;
    const bf16_t* kb = k + ((long)n * L) * ldk + h * D;
    const bf16_t* vb = v + ((long)n * L) * ldv + h * D;
    bf16_t* ob = o + ((long)n * L) * ldo + h * D;
    float* zb = zinv + ((long)n * L) * H + h;

    const int srow = tid >> 3, scol = (tid & 7) * 8;
    const int nch = (L + C - 1) / C;
    const int cbeg = seg * cps, cend = min(nch, cbeg + cps);
    if (cbeg >= nch || (STATE_ONLY && seg == P - 1)) return;   // nothing follows the last segment
    uint4 rq[2], rk[2], rv[2];
    const bf16x8 ones0 = ones_if(l31 == 0);   // A operand: row 0 of the ones block, all k

    // this stream's q / k / v rows as buffer resources: rows >= L read back as zeros (hardware range check)
    const __amdgpu_buffer_rsrc_t qr = make_rsrc(qb, (uint32_t)(((long)(L - 1) * ldq + D) * 2));
    const __amdgpu_buffer_rsrc_t kr = make_rsrc(kb, (uint32_t)(((long)(L - 1) * ldk + D) * 2));
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(vb, (uint32_t)(((long)(L - 1) * ldv + D) * 2));
#define CLA_LOAD(c)                                                                      \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                   \
        const uint32_t row = (uint32_t)(c) * C + srow + 32 * it;                         \
        if (!STATE_ONLY) rq[it] = buf_load16(qr, (row * (uint32_t)ldq + scol) * 2);      \
        rk[it] = buf_load16(kr, (row * (uint32_t)ldk + scol) * 2);                       \
        rv[it] = buf_load16(vr, (row * (uint32_t)ldv + scol) * 2);                       \
    }
#define CLA_STORE(c)                                                                     \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                   \
        const int row = srow + 32 * it;                                                  \
        const long grow = (long)(c) * C + row;                                           \
        if (grow < L) {                                                                  \
            const float z = 1.0f / (dens[0][row] + dens[1][row] + eps);                  \
            const float4 a = *reinterpret_cast<const float4*>(os + row * LDO + scol);    \
            const float4 b = *reinterpret_cast<const float4*>(os + row * LDO + scol + 4);\
            const float x[8] = {a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z}; \
            *reinterpret_cast<uint4*>(ob + grow * ldo + scol) = pack8(x);                \
            if ((tid & 7) == 0) zb[grow * H] = z;                                        \
        }                                                                                \
    }

    CLA_LOAD(cbeg);
    f32x16 S0 = zero16(), S1 = zero16();  // S[e-half t][m-half wj]: rows e on regs, cols m on lanes
    f32x16 Sa = zero16();                 // ones-column state, e-half wj: Sa[e][0] = ksum[32wj + e]
    if (!STATE_ONLY && pre && seg > 0) {  // the state this segment starts from (both wi waves hold the same copy)
        const float* t = pre + (((long)sid * P + seg) * 2 + wj) * (3 * 1024);
        S0 = load_tile(t, lane);
        S1 = load_tile(t + 1024, lane);
        Sa = load_tile(t + 2048, lane);
    }

    for (int c = cbeg; c < cend; ++c) {
        if (!STATE_ONLY && c > cbeg) { CLA_STORE(c - 1); }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = srow + 32 * it;
            const bool ok = c * C + row < L;
            float x[8];
            if (!STATE_ONLY) {
                unpack8(rq[it], x);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = ok ? phi(x[j]) : 0.f;
                put_row(qs, row, scol, pack8(x));
            }
            unpack8(rk[it], x);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = ok ? phi(x[j]) : 0.f;
            put_row(ks, row, scol, pack8(x));
            put_row(vs, row, scol, rv[it]);
        }
        __syncthreads();
        if (c + 1 < cend) { CLA_LOAD(c + 1); }

        // scores as A^T (rows j on regs, cols i on lanes) -> as_[i][j], masked j <= i
        if (!STATE_ONLY) {
        if (!(wi == 0 && wj == 1)) {
            const f32x16 AT = prod_rows(zero16(), ks, 32 * wj + l31, qs, 32 * wi + l31, 0, 4, hf);
            put_acc_T(as_, 32 * wi + l31, 32 * wj, AT, hf, 0, wi == wj ? l31 : 64, 0.f);
        }
        __syncthreads();

        {
            // O^T tile (rows m on regs, cols i on lanes) = v^T A~^T + S^T phi(q)^T ; ones row alongside
            f32x16 O = zero16(), Oa = zero16();
            const int nks = wi == 0 ? 2 : 4;
#pragma unroll 2
            for (int s = 0; s < nks; ++s) {
                const bf16x8 b = row8(as_, 32 * wi + l31, 16 * s + 8 * hf);
                O = mfma(tfrag8(vs, 16 * s, 32 * wj, lane), b, O);
                if (wj == 0) Oa = mfma(ones0, b, Oa);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x16& S = t == 0 ? S0 : S1;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 hi, lo;
                    acc_frag(S, s, hi, lo);
                    const bf16x8 b = perm8(qs, 32 * wi + l31, 32 * t + 16 * s, hf);
                    O = mfma_hl(hi, lo, b, O);
                    if (t == wj) {
                        acc_frag(Sa, s, hi, lo);
                        Oa = mfma_hl(hi, lo, b, Oa);
                    }
                }
            }
            put_acc_T_f32(os, 32 * wi + l31, 32 * wj, O, hf);
            if (hf == 0) dens[wj][32 * wi + l31] = Oa[0];   // row 0 of the ones block
        }
        }   // !STATE_ONLY
        // states: S_t[e][m] += sum_j phi(k)[j][32t+e] v[j][32wj+m] ;  Sa[e][0] += sum_j phi(k)[j][32wj+e]
        // (STATE_ONLY: the two wi waves of a column half split the chunk's four k-steps; their sums are added later)
#pragma unroll 2
        for (int s = STATE_ONLY ? 2 * wi : 0; s < (STATE_ONLY ? 2 * wi + 2 : 4); ++s) {
            const bf16x8 b = tfrag8(vs, 16 * s, 32 * wj, lane);
            const bf16x8 a0 = tfrag8(ks, 16 * s, 0, lane);
            const bf16x8 a1 = tfrag8(ks, 16 * s, 32, lane);
            S0 = mfma(a0, b, S0);
            S1 = mfma(a1, b, S1);
            Sa = mfma(wj == 0 ? a0 : a1, ones_if(l31 == 0), Sa);
        }
        __syncthreads();
    }
    if (STATE_ONLY) {
        float* t = part + (((long)sid * P + seg) * 4 + w) * (3 * 1024);
        store_tile(t, lane, S0);
        store_tile(t + 1024, lane, S1);
        store_tile(t + 2048, lane, Sa);
        return;
    }
    CLA_STORE(cend - 1);
#undef CLA_LOAD
#undef CLA_STORE
    if (fin) {
        // P == 1: the state after the last token, for the one-sweep backward (cla_bwd_sweep_bf16_kernel), as
        // fin[stream][m][e] = S[e][m] (64 x 64 f32, row-major) followed by ksum[e] (64 f32); through the os tile so the
        // global writes are whole rows
        __syncthreads();
        if (wi == 0) {
            put_acc_T_f32(os, 32 * wj + l31, 0, S0, hf);
            put_acc_T_f32(os, 32 * wj + l31, 32, S1, hf);
            if (l31 == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dens[0][32 * wj + acc_row(r, hf)] = Sa[r];
            }
        }
        __syncthreads();
        float* f = fin + (long)sid * FIN_FLOATS;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = (tid >> 4) + 16 * it, col = (tid & 15) * 4;
            *reinterpret_cast<float4*>(f + row * 64 + col) = *reinterpret_cast<const float4*>(os + row * LDO + col);
        }
        if (tid < 64) f[64 * 64 + tid] = dens[0][tid];
    }
}

// part (streams, P, 4 waves, NT tiles, 1024) f32 -> pre (streams, P, 2 column halves, NT tiles, 1024): tiles
// [0, nfwd) get the EXCLUSIVE PREFIX over the segments of (wave wj + wave 2 + wj), tiles [nfwd, NT) the exclusive SUFFIX.
__global__ __launch_bounds__(256) void seg_prefix_kernel(const float* __restrict__ part, float* __restrict__ pre,
                                                         long total, int P, int NT, int nfwd) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;    // over (stream, wj, tile, element)
    if (i >= total) return;
    const int e = (int)(i & 1023);
    const int tile = (int)((i >> 10) % NT);
    const int wj = (int)((i >> 10) / NT) & 1;
    const long sid = (i >> 10) / NT / 2;
    const long pstride = (long)4 * NT * 1024, ostride = (long)2 * NT * 1024;
    const float* a = part + sid * P * pstride + ((long)wj * NT + tile) * 1024 + e;          // wave (0, wj)
    const float* b = a + (long)2 * NT * 1024;                                               // wave (1, wj)
    float* o = pre + sid * P * ostride + ((long)wj * NT + tile) * 1024 + e;
    float run = 0.f;
    if (tile < nfwd) {
        for (int p = 0; p < P; ++p) {
            o[p * ostride] = run;
            if (p < P - 1) run += a[p * pstride] + b[p * pstride];
        }
    } else {
        for (int p = P - 1; p >= 0; --p) {
            o[p * ostride] = run;
            if (p > 0) run += a[p * pstride] + b[p * pstride];
        }
    }
}

// Common staging of the backward kernels for one (row, 8-column) slot:
//   g = dout * z  (bf16),  dden = -(dout . out) * z  (row dot over the 8 threads of the row)
__device__ __forceinline__ uint4 stage_g(uint4 rdo, uint4 ro, float z, float& dden) {
    float a[8], b[8];
    unpack8(rdo, a);
    unpack8(ro, b);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        dot = fmaf(a[j], b[j], dot);
        a[j] *= z;
    }
    // sum over the 8 lanes that share a row: DPP moves (quad xor 1, xor 2, half-row mirror) instead of three
    // ds_bpermute round trips through the LDS pipe
    dot += dpp_move<0xB1>(dot);
    dot += dpp_move<0x4E>(dot);
    dot += dpp_move<0x141>(dot);
    dden = -dot * z;
    return pack8(a);
}

// Per-stream column sums of a stored output: thread (row slot, 8-column slot) holds bsum[8]; reduce over
// the 32 threads that share a column slot and write 64 floats.  `scratch` = any >= 256-float LDS area that
// is no longer needed (called after the chunk loop).
__device__ __forceinline__ void colsum_store(float (&bsum)[8], float* scratch, float* dst, int tid, int lane, int w) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = bsum[j];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if ((lane >> 3) == 0) scratch[w * 64 + (lane & 7) * 8 + j] = v;
    }
    __syncthreads();
    if (tid < 64) dst[tid] = (scratch[tid] + scratch[64 + tid]) + (scratch[128 + tid] + scratch[192 + tid]);
}

// ------------------------------------------------------------------------------------------------
// backward, dQ (forward scan).  W_ij = g_i . v_j + dden_i (j <= i)
//   dqf_i = sum_{j<=i} W_ij kf_j = (W kf)_i + S_prev g_i + dden_i ksum_prev ;  dQ = dqf * phi'(Q)
// wave (wi, wj): W tile (i-half wi, j-half wj); dq tile (i-half wi, e-half wj).
// ------------------------------------------------------------------------------------------------
// HAS_DDEN: dden (N, L, H) f32 was written by the reverse-scan kernel (launched first); the `out` stream -- needed
// only to rebuild it -- is then not read at all (one of the six 16-byte streams of this kernel).
template <bool HAS_DDEN>
__global__ __launch_bounds__(256) void cla_bwd_dq_bf16_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, const float* __restrict__ zinv,
    const float* __restrict__ dden_in, bf16_t* __restrict__ dq, float* __restrict__ csum, int H, int L, long ldq,
    long ldk, long ldv, long ldo, long lddo, long lddq, int P, int cps, const float* __restrict__ pre) {
    __shared__ __attribute__((aligned(16))) bf16_t gs[C * LD];   // g       [i][m]
    __shared__ __attribute__((aligned(16))) bf16_t vs[C * LD];   // v       [j][m]
    __shared__ __attribute__((aligned(16))) bf16_t ks[C * LD];   // phi(k)  [j][e]
    __shared__ __attribute__((aligned(16))) bf16_t ws[C * LD];   // masked W [i][j]
    __shared__ __attribute__((aligned(16))) float os[C * LDO];   // dqf tile [i][e]
    __shared__ float dd[C];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = w >> 1, wj = w & 1;
    const int l31 = lane & 31, hf = lane >> 5;
    const int sid = blockIdx.x / P, seg = blockIdx.x % P;      // stream, segment (P == 1: whole sequences)
    const int n = sid / H, h = sid % H;
    const bf16_t* qb = q + ((long)n * L) * ldq + h * D;
    const bf16_t* kb = k + ((long)n * L) * ldk + h * D;
    const bf16_t* vb = v + ((long)n * L) * ldv + h * D;
    const bf16_t* ob = out + ((long)n * L) * ldo + h * D;
    const bf16_t* gb = dout + ((long)n * L) * lddo + h * D;
    const float* zb = zinv + ((long)n * L) * H + h;
    const float* ddb = HAS_DDEN ? dden_in + ((long)n * L) * H + h : zb;
    bf16_t* dqb = dq + ((long)n * L) * lddq + h * D;

    const int srow = tid >> 3, scol = (tid & 7) * 8;
    const int nch = (L + C - 1) / C;
    const int cbeg = seg * cps, cend = min(nch, cbeg + cps);
    if (cbeg >= nch) return;
    uint4 rq[2], rk[2], rv[2], rg[2], ro[2], rqp[2];
    float rz[2], rdd[2];

    // this stream's rows as buffer resources: rows >= L read back as zeros (hardware range check, no branches)
    const __amdgpu_buffer_rsrc_t qr = make_rsrc(qb, (uint32_t)(((long)(L - 1) * ldq + D) * 2));
    const __amdgpu_buffer_rsrc_t kr = make_rsrc(kb, (uint32_t)(((long)(L - 1) * ldk + D) * 2));
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(vb, (uint32_t)(((long)(L - 1) * ldv + D) * 2));
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(gb, (uint32_t)(((long)(L - 1) * lddo + D) * 2));
    const __amdgpu_buffer_rsrc_t orr = make_rsrc(ob, (uint32_t)(((long)(L - 1) * ldo + D) * 2));
    const __amdgpu_buffer_rsrc_t zr = make_rsrc(zb, (uint32_t)(((long)(L - 1) * H + 1) * 4));
    const __amdgpu_buffer_rsrc_t ddr = make_rsrc(ddb, (uint32_t)(((long)(L - 1) * H + 1) * 4));
#define CLA_LOAD(c)                                                                      \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                   \
        const uint32_t row = (uint32_t)(c) * C + srow + 32 * it;                         \
        rq[it] = buf_load16(qr, (row * (uint32_t)ldq + scol) * 2);                       \
        rk[it] = buf_load16(kr, (row * (uint32_t)ldk + scol) * 2);                       \
        rv[it] = buf_load16(vr, (row * (uint32_t)ldv + scol) * 2);                       \
        rg[it] = buf_load16(gr, (row * (uint32_t)lddo + scol) * 2);                      \
        rz[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zr, (int)(row * (uint32_t)H * 4), 0, 0)); \
        if (HAS_DDEN)                                                                    \
            rdd[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ddr, (int)(row * (uint32_t)H * 4), 0, 0)); \
        else                                                                             \
            ro[it] = buf_load16(orr, (row * (uint32_t)ldo + scol) * 2);                  \
    }
#define CLA_STORE(c)                                                                     \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                   \
        const int row = srow + 32 * it;                                                  \
        const long grow = (long)(c) * C + row;                                           \
        if (grow < L) {                                                                  \
            const float4 a = *reinterpret_cast<const float4*>(os + row * LDO + scol);    \
            const float4 b = *reinterpret_cast<const float4*>(os + row * LDO + scol + 4);\
            float x[8];                                                                  \
            unpack8(rqp[it], x);                                                         \
            x[0] = a.x * dphi(x[0]); x[1] = a.y * dphi(x[1]); x[2] = a.z * dphi(x[2]);   \
            x[3] = a.w * dphi(x[3]); x[4] = b.x * dphi(x[4]); x[5] = b.y * dphi(x[5]);   \
            x[6] = b.z * dphi(x[6]); x[7] = b.w * dphi(x[7]);                            \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) bsum[j] += x[j];               \
            *reinterpret_cast<uint4*>(dqb + grow * lddq + scol) = pack8(x);              \
        }                                                                                \
    }

    float bsum[8];   // column sums of dQ over this stream (bias gradient of the query projection)
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    CLA_LOAD(cbeg);
    f32x16 T0 = zero16(), T1 = zero16();  // ST_t[m][e] = S[32wj+e][32t+m]: rows m on regs, cols e on lanes
    f32x16 Ta = zero16();                 // ones row: Ta[0][e] = ksum[32wj + e]
    if (pre && seg > 0) {                 // prefix state of this segment: tiles 0..2 of the backward's 8
        const float* t = pre + (((long)sid * P + seg) * 2 + wj) * (8 * 1024);
        T0 = load_tile(t, lane);
        T1 = load_tile(t + 1024, lane);
        Ta = load_tile(t + 2048, lane);
    }
    const bf16x8 ones0 = ones_if(l31 == 0);

    for (int c = cbeg; c < cend; ++c) {
        if (c > cbeg) { CLA_STORE(c - 1); }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = srow + 32 * it;
            const bool ok = c * C + row < L;
            float dden;
            uint4 gp;
            if (HAS_DDEN) {
                float a[8];
                unpack8(rg[it], a);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] *= rz[it];
                gp = pack8(a);
                dden = rdd[it];
            } else {
                gp = stage_g(rg[it], ro[it], rz[it], dden);
            }
            if ((tid & 7) == 0) dd[row] = dden;
            put_row(gs, row, scol, gp);
            put_row(vs, row, scol, rv[it]);
            float x[8];
            unpack8(rk[it], x);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = ok ? phi(x[j]) : 0.f;
            put_row(ks, row, scol, pack8(x));
            rqp[it] = rq[it];
        }
        __syncthreads();
        if (c + 1 < cend) { CLA_LOAD(c + 1); }

        const float dden_i = dd[32 * wi + l31];
        // W^T (rows j on regs, cols i on lanes) = v g^T  (+ dden_i) -> ws[i][j], masked j <= i
        if (!(wi == 0 && wj == 1)) {
            const f32x16 WT = prod_rows(zero16(), vs, 32 * wj + l31, gs, 32 * wi + l31, 0, 4, hf);
            put_acc_T(ws, 32 * wi + l31, 32 * wj, WT, hf, 0, wi == wj ? l31 : 64, dden_i);
        }
        __syncthreads();

        {
            // dqf^T tile (rows e on regs, cols i on lanes) = kf^T W^T + ST^T g^T + ksum (x) dden
            f32x16 Q = zero16();
            const int nks = wi == 0 ? 2 : 4;
#pragma unroll 2
            for (int s = 0; s < nks; ++s)
                Q = mfma(tfrag8(ks, 16 * s, 32 * wj, lane), row8(ws, 32 * wi + l31, 16 * s + 8 * hf), Q);
            Q = prod_accA(Q, T0, T1, gs, 32 * wi + l31, hf);
            {
                bf16x8 hi, lo;
                acc_frag(Ta, 0, hi, lo);
                const bf16x8 b = first_if(hf == 0, dden_i);
                Q = mfma_hl(hi, lo, b, Q);
            }
            put_acc_T_f32(os, 32 * wi + l31, 32 * wj, Q, hf);
        }
        // ST_t[m][e] += sum_j v[j][32t+m] kf[j][32wj+e] ;  Ta[0][e] += sum_j kf[j][32wj+e]
#pragma unroll 2
        for (int s = 0; s < 4; ++s) {
            const bf16x8 b = tfrag8(ks, 16 * s, 32 * wj, lane);
            T0 = mfma(tfrag8(vs, 16 * s, 0, lane), b, T0);
            T1 = mfma(tfrag8(vs, 16 * s, 32, lane), b, T1);
            Ta = mfma(ones0, b, Ta);
        }
        __syncthreads();
    }
    CLA_STORE(cend - 1);
#undef CLA_LOAD
#undef CLA_STORE
    if (csum) colsum_store(bsum, os, csum + (((long)n * P + seg) * H + h) * D, tid, lane, w);
}

// ------------------------------------------------------------------------------------------------
// backward, dK and dV (reverse scan).
//   dkf_j = sum_{i>=j} W_ij qf_i = (W^T qf)_j + R_next v_j + r1_next ;  dK = dkf * phi'(K),  phi' = min(phi, 1)
//   dv_j  = sum_{i>=j} A_ij g_i  = (A^T g)_j + R_next^T kf_j
//   R[e][m] = sum_{i later} qf_i[e] g_i[m],  r1[e] = sum_{i later} qf_i[e] dden_i
// wave (wi, wj): W and A tiles (i-half wi, j-half wj); dk tile (j-half wi, e-half wj); dv tile (j-half wi, m-half wj).
// ------------------------------------------------------------------------------------------------
// STATE_ONLY (few streams, see the forward kernel): the segment's increments of ALL backward states, 8 tiles per
// wave -- [0..2] the dq scan's ST_0, ST_1, ones row (from k, v), [3..7] this kernel's RT_0, RT_1, RTa, R2_0, R2_1.
template <bool STATE_ONLY>
__global__ __launch_bounds__(256, 2) void cla_bwd_dkdv_bf16_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, const float* __restrict__ zinv,
    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, float* __restrict__ csum_k, float* __restrict__ csum_v,
    float* __restrict__ dden_out, int H, int L, long ldq, long ldk, long ldv, long ldo, long lddo, long lddk,
    long lddv, int P, int cps, const float* __restrict__ pre, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) bf16_t qs[C * LD];   // phi(q)  [i][e]
    __shared__ __attribute__((aligned(16))) bf16_t ks[C * LD];   // phi(k)  [j][e]
    __shared__ __attribute__((aligned(16))) bf16_t vs[C * LD];   // v       [j][m]
    __shared__ __attribute__((aligned(16))) bf16_t gs[C * LD];   // g       [i][m]
    __shared__ __attribute__((aligned(16))) bf16_t wt[C * LD];   // masked W^T [j][i]
    __shared__ __attribute__((aligned(16))) bf16_t at[C * LD];   // masked A^T [j][i]
    __shared__ __attribute__((aligned(16))) bf16_t ok_[C * LD];  // dkf tile [j][e] (bf16)
    __shared__ __attribute__((aligned(16))) bf16_t ov[C * LD];   // dv tile  [j][m] (bf16)
    __shared__ float dd[C];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = w >> 1, wj = w & 1;
    const int l31 = lane & 31, hf = lane >> 5;
    const int sid = blockIdx.x / P, seg = blockIdx.x % P;      // stream, segment (P == 1: whole sequences)
    const int n = sid / H, h = sid % H;
    const bf16_t* qb = q + ((long)n * L) * ldq + h * D;
    const bf16_t* kb = k + ((long)n * L) * ldk + h * D;
    const bf16_t* vb = v + ((long)n * L) * ldv + h * D;
    const bf16_t* ob = out + ((long)n * L) * ldo + h * D;
    const bf16_t* gb = dout + ((long)n * L) * lddo + h * D;
    const float* zb = zinv + ((long)n * L) * H + h;
    bf16_t* dkb = dk + ((long)n * L) * lddk + h * D;
    bf16_t* dvb = dv + ((long)n * L) * lddv + h * D;
    float* ddb = dden_out ? dden_out + ((long)n * L) * H + h : nullptr;   // dden for the dq kernel (it then skips `out`)

    const int srow = tid >> 3, scol = (tid & 7) * 8;
    const int nch = (L + C - 1) / C;
    const int cbeg = seg * cps, cend = min(nch, cbeg + cps);
    if (cbeg >= nch) return;
    uint4 rq[2], rk[2], rv[2], rg[2], ro[2];
    float rz[2];

    // this stream's rows as buffer resources: rows >= L read back as zeros (hardware range check, no branches)
    const __amdgpu_buffer_rsrc_t qr = make_rsrc(qb, (uint32_t)(((long)(L - 1) * ldq + D) * 2));
    const __amdgpu_buffer_rsrc_t kr = make_rsrc(kb, (uint32_t)(((long)(L - 1) * ldk + D) * 2));
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(vb, (uint32_t)(((long)(L - 1) * ldv + D) * 2));
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(gb, (uint32_t)(((long)(L - 1) * lddo + D) * 2));
    const __amdgpu_buffer_rsrc_t orr = make_rsrc(ob, (uint32_t)(((long)(L - 1) * ldo + D) * 2));
    const __amdgpu_buffer_rsrc_t zr = make_rsrc(zb, (uint32_t)(((long)(L - 1) * H + 1) * 4));
#define CLA_LOAD(c)                                                                      \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                   \
        const uint32_t row = (uint32_t)(c) * C + srow + 32 * it;                         \
        rq[it] = buf_load16(qr, (row * (uint32_t)ldq + scol) * 2);                       \
        rk[it] = buf_load16(kr, (row * (uint32_t)ldk + scol) * 2);                       \
        rv[it] = buf_load16(vr, (row * (uint32_t)ldv + scol) * 2);                       \
        rg[it] = buf_load16(gr, (row * (uint32_t)lddo + scol) * 2);                      \
        ro[it] = buf_load16(orr, (row * (uint32_t)ldo + scol) * 2);                      \
        rz[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zr, (int)(row * (uint32_t)H * 4), 0, 0)); \
    }
    // store slot (row, scol): each thread reads exactly the ks slot it re-stages next, so no barrier is needed
#define CLA_STORE(c)                                                                     \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                   \
        const int row = srow + 32 * it;                                                  \
        const long grow = (long)(c) * C + row;                                           \
        if (grow < L) {                                                                  \
            float x[8], f[8];                                                            \
            unpack8(*reinterpret_cast<const uint4*>(ok_ + row * LD + scol), x);          \
            unpack8(*reinterpret_cast<const uint4*>(ks + row * LD + scol), f);           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) x[j] *= fminf(f[j], 1.0f);     \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) bsk[j] += x[j];                \
            *reinterpret_cast<uint4*>(dkb + grow * lddk + scol) = pack8(x);              \
            const uint4 vv = *reinterpret_cast<const uint4*>(ov + row * LD + scol);      \
            unpack8(vv, f);                                                              \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) bsv[j] += f[j];                \
            *reinterpret_cast<uint4*>(dvb + grow * lddv + scol) = vv;                    \
        }                                                                                \
    }

    float bsk[8], bsv[8];   // column sums of dK, dV over this stream (bias gradients of the key / value projections)
#pragma unroll
    for (int j = 0; j < 8; ++j) bsk[j] = bsv[j] = 0.f;

    CLA_LOAD(cend - 1);
    f32x16 RT0 = zero16(), RT1 = zero16();  // RT_t[m][e] = R[32wj+e][32t+m]: rows m on regs, cols e on lanes
    f32x16 RTa = zero16();                  // RTa[0][e] = r1[32wj + e]
    f32x16 R20 = zero16(), R21 = zero16();  // R2_t[e][m] = R[32t+e][32wj+m]: rows e on regs, cols m on lanes
    f32x16 T0 = zero16(), T1 = zero16(), Ta = zero16();   // STATE_ONLY: the dq scan's states (its kernel's layout)
    const bf16x8 ones0 = ones_if(l31 == 0);
    if (!STATE_ONLY && pre && seg < P - 1) {              // suffix state of this segment: tiles 3..7
        const float* t = pre + (((long)sid * P + seg) * 2 + wj) * (8 * 1024) + 3 * 1024;
        RT0 = load_tile(t, lane);
        RT1 = load_tile(t + 1024, lane);
        RTa = load_tile(t + 2048, lane);
        R20 = load_tile(t + 3072, lane);
        R21 = load_tile(t + 4096, lane);
    }

    for (int c = cend - 1; c >= cbeg; --c) {
        if (!STATE_ONLY && c < cend - 1) { CLA_STORE(c + 1); }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = srow + 32 * it;
            const bool ok = c * C + row < L;
            float dden;
            const uint4 gp = stage_g(rg[it], ro[it], rz[it], dden);
            if ((tid & 7) == 0) {
                dd[row] = dden;
                if (!STATE_ONLY && ddb && ok) ddb[((long)c * C + row) * H] = dden;
            }
            put_row(gs, row, scol, gp);
            put_row(vs, row, scol, rv[it]);
            float x[8];
            unpack8(rk[it], x);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = ok ? phi(x[j]) : 0.f;
            put_row(ks, row, scol, pack8(x));
            unpack8(rq[it], x);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = ok ? phi(x[j]) : 0.f;
            put_row(qs, row, scol, pack8(x));
        }
        __syncthreads();
        if (c > cbeg) { CLA_LOAD(c - 1); }

        if (!STATE_ONLY) {
        // W and A tiles (rows i on regs, cols j on lanes), kept where i >= j, written transposed: wt[j][i], at[j][i]
        if (!(wi == 0 && wj == 1)) {
            f32x16 W = prod_rows(zero16(), gs, 32 * wi + l31, vs, 32 * wj + l31, 0, 4, hf);
            // + dden_i * 1 (augmentation k-step: A = [dden_i at k=0], B = [1 at k=0])
            W = mfma(first_if(hf == 0, dd[32 * wi + l31]), first_if(hf == 0, 1.0f), W);
            put_acc_T(wt, 32 * wj + l31, 32 * wi, W, hf, wi == wj ? l31 : 0, 64, 0.f);
            const f32x16 A = prod_rows(zero16(), qs, 32 * wi + l31, ks, 32 * wj + l31, 0, 4, hf);
            put_acc_T(at, 32 * wj + l31, 32 * wi, A, hf, wi == wj ? l31 : 0, 64, 0.f);
        }
        __syncthreads();

        {
            // contraction over i >= j: for j-half 1 only i-half 1 contributes
            const int s0 = wi == 1 ? 2 : 0;
            // dkf^T tile (rows e on regs, cols j on lanes) = qf^T W + RT^T v^T + r1 (x) 1
            f32x16 K = zero16();
#pragma unroll 2
            for (int s = s0; s < 4; ++s)
                K = mfma(tfrag8(qs, 16 * s, 32 * wj, lane), row8(wt, 32 * wi + l31, 16 * s + 8 * hf), K);
            K = prod_accA(K, RT0, RT1, vs, 32 * wi + l31, hf);
            {
                bf16x8 hi, lo;
                acc_frag(RTa, 0, hi, lo);
                const bf16x8 b = first_if(hf == 0, 1.0f);
                K = mfma_hl(hi, lo, b, K);
            }
            put_acc_T(ok_, 32 * wi + l31, 32 * wj, K, hf, 0, 64, 0.f);
            // dv^T tile (rows m on regs, cols j on lanes) = g^T A + R2^T kf^T
            f32x16 V = zero16();
#pragma unroll 2
            for (int s = s0; s < 4; ++s)
                V = mfma(tfrag8(gs, 16 * s, 32 * wj, lane), row8(at, 32 * wi + l31, 16 * s + 8 * hf), V);
            V = prod_accA(V, R20, R21, ks, 32 * wi + l31, hf);
            put_acc_T(ov, 32 * wi + l31, 32 * wj, V, hf, 0, 64, 0.f);
        }
        }   // !STATE_ONLY
        // RT_t[m][e] += sum_i g[i][32t+m] qf[i][32wj+e] ; RTa[0][e] += sum_i dden_i qf[i][32wj+e]
        // R2_t[e][m] += sum_i qf[i][32t+e] g[i][32wj+m]
        // (STATE_ONLY: the two wi waves of a column half split the chunk's four k-steps)
#pragma unroll 2
        for (int s = STATE_ONLY ? 2 * wi : 0; s < (STATE_ONLY ? 2 * wi + 2 : 4); ++s) {
            const bf16x8 g0 = tfrag8(gs, 16 * s, 0, lane), g1 = tfrag8(gs, 16 * s, 32, lane);
            const bf16x8 q0 = tfrag8(qs, 16 * s, 0, lane), q1 = tfrag8(qs, 16 * s, 32, lane);
            const bf16x8 bq = wj == 0 ? q0 : q1;
            const bf16x8 bg = wj == 0 ? g0 : g1;
            RT0 = mfma(g0, bq, RT0);
            RT1 = mfma(g1, bq, RT1);
            R20 = mfma(q0, bg, R20);
            R21 = mfma(q1, bg, R21);
            // A operand row 0 = dden over this k-step's 8 tokens (lane l31 == 0 only), hi + lo
            bf16x8 dh, dl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = l31 == 0 ? dd[16 * s + 8 * hf + j] : 0.f;
                const __bf16 hh = (__bf16)x;
                dh[j] = hh;
                dl[j] = (__bf16)(x - (float)hh);
            }
            RTa = mfma(dh, bq, RTa);
            RTa = mfma(dl, bq, RTa);
            if (STATE_ONLY) {
                // the dq scan's states over the same tokens: ST_t[m][e] += sum_j v[j][32t+m] kf[j][32wj+e] ; ones row
                const bf16x8 bk = tfrag8(ks, 16 * s, 32 * wj, lane);
                T0 = mfma(tfrag8(vs, 16 * s, 0, lane), bk, T0);
                T1 = mfma(tfrag8(vs, 16 * s, 32, lane), bk, T1);
                Ta = mfma(ones0, bk, Ta);
            }
        }
        __syncthreads();
    }
    if (STATE_ONLY) {
        float* t = part + (((long)sid * P + seg) * 4 + w) * (8 * 1024);
        store_tile(t, lane, T0);
        store_tile(t + 1024, lane, T1);
        store_tile(t + 2048, lane, Ta);
        store_tile(t + 3072, lane, RT0);
        store_tile(t + 4096, lane, RT1);
        store_tile(t + 5120, lane, RTa);
        store_tile(t + 6144, lane, R20);
        store_tile(t + 7168, lane, R21);
        return;
    }
    CLA_STORE(cbeg);
#undef CLA_LOAD
#undef CLA_STORE
    if (csum_k) {
        float* scratch = reinterpret_cast<float*>(wt);   // score tiles are dead after the loop
        colsum_store(bsk, scratch, csum_k + (((long)n * P + seg) * H + h) * D, tid, lane, w);
        colsum_store(bsv, scratch, csum_v + (((long)n * P + seg) * H + h) * D, tid, lane, w);
    }
}

// ------------------------------------------------------------------------------------------------
// backward in ONE reverse sweep (whole sequences, P == 1): dQ, dK and dV from a single pass over q, k, v, out, dout.
//   8 waves.  Waves 0-3 (the "KV" group) are the reverse scan above (dK, dV, states R).  Waves 4-7 (the "Q" group) are
//   the dQ scan run BACKWARDS: its state S_prev (the sum over EARLIER tokens) is the forward pass's final state `fin`
//   minus the contribution of every chunk the sweep has passed, including the current one.  The subtraction runs in the
//   f32 accumulators with the very bf16 operands the forward added, so what is left differs from the true prefix by
//   f32 rounding only (~2^-24 of the running total per chunk, against the 2^-9 of the bf16 operands); chunk 0 starts
//   from an exact zero.
//   Both groups work from the same staged LDS tiles: every input is read from HBM once (5 streams in, 3 out = 8 units
//   against the 12 of the two-kernel schedule) and the W score tile is computed once.  Input tiles are double-buffered
//   in LDS and two chunks are in flight in registers, so one workgroup per CU keeps as many bytes in flight as the two
//   4-wave workgroups of the split kernels did.
//   dden rides in the two padding columns 64 / 65 of the g tile (bf16 hi + lo), so r1 = sum_i dden_i qf_i is one more
//   transposed-fragment MFMA per k-step instead of a per-wave hi/lo fragment build on the VALU.
// ------------------------------------------------------------------------------------------------
struct ChunkRegs {
    uint4 q, k, v, g, o;
    float z;
};

// min(f, 1) for f >= 0 (f = a staged phi value): one v_med3_f32, no NaN canonicalisation in front
__device__ __forceinline__ float min1(float f) { return __builtin_amdgcn_fmed3f(f, 0.0f, 1.0f); }

// accumulator tile (rows rr on regs, cols on lanes) times min(f, 1) of the SAME positions of tile `f_t` (phi' of the
// staged phi values: phi' = min(phi, 1)) -> x[xrow][c0 + rr], bf16, rounded once
__device__ __forceinline__ void put_acc_T_dphi(bf16_t* x, const bf16_t* f_t, int xrow, int c0, const f32x16& acc, int hf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint2 fr = *reinterpret_cast<const uint2*>(f_t + xrow * LD + c0 + 8 * g + 4 * hf);
        const float f0 = __uint_as_float(fr.x << 16), f1 = __uint_as_float(fr.x & 0xffff0000u);
        const float f2 = __uint_as_float(fr.y << 16), f3 = __uint_as_float(fr.y & 0xffff0000u);
        bf16x4 p;
        p[0] = (__bf16)(acc[4 * g] * min1(f0));
        p[1] = (__bf16)(acc[4 * g + 1] * min1(f1));
        p[2] = (__bf16)(acc[4 * g + 2] * min1(f2));
        p[3] = (__bf16)(acc[4 * g + 3] * min1(f3));
        *reinterpret_cast<uint2*>(x + xrow * LD + c0 + 8 * g + 4 * hf) = __builtin_bit_cast(uint2, p);
    }
}
// accumulator tile -> x[xrow][c0 + rr], bf16, all of it (no mask, nothing added)
__device__ __forceinline__ void put_acc_T_all(bf16_t* x, int xrow, int c0, const f32x16& acc, int hf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 p;
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = (__bf16)acc[4 * g + u];
        *reinterpret_cast<uint2*>(x + xrow * LD + c0 + 8 * g + 4 * hf) = __builtin_bit_cast(uint2, p);
    }
}
// accumulator tile -> x[xrow][c0 + rr], bf16, kept where rr >= lo (the diagonal score tiles)
__device__ __forceinline__ void put_acc_T_ge(bf16_t* x, int xrow, int c0, const f32x16& acc, int hf, int lo) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 p;
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = (__bf16)(8 * g + 4 * hf + u >= lo ? acc[4 * g + u] : 0.f);
        *reinterpret_cast<uint2*>(x + xrow * LD + c0 + 8 * g + 4 * hf) = __builtin_bit_cast(uint2, p);
    }
}
// sum over four k-steps of a_t[arow][k] * b_t[brow][k], chain started from the MFMA's zero C operand
__device__ __forceinline__ f32x16 prod_rows4(const bf16_t* a_t, int arow, const bf16_t* b_t, int brow, int hf) {
    f32x16 acc = mfma(row8(a_t, arow, 8 * hf), row8(b_t, brow, 8 * hf), zero16());
#pragma unroll
    for (int s = 1; s < 4; ++s) acc = mfma(row8(a_t, arow, 16 * s + 8 * hf), row8(b_t, brow, 16 * s + 8 * hf), acc);
    return acc;
}
__device__ __forceinline__ bf16x8 negate8(bf16x8 x) {
    uint4 u = __builtin_bit_cast(uint4, x);
    u.x ^= 0x80008000u; u.y ^= 0x80008000u; u.z ^= 0x80008000u; u.w ^= 0x80008000u;
    return __builtin_bit_cast(bf16x8, u);
}

__global__ __launch_bounds__(512, 1) void cla_bwd_sweep_bf16_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, const float* __restrict__ zinv,
    const float* __restrict__ fin, bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
    float* __restrict__ csum_q, float* __restrict__ csum_k, float* __restrict__ csum_v, int H, int L, long ldq,
    long ldk, long ldv, long ldo, long lddo, long lddq, long lddk, long lddv) {
    constexpr int TILE = C * LD;
    __shared__ __attribute__((aligned(16))) bf16_t in_[2 * 4 * TILE];   // [buffer][phi(q) | phi(k) | v | g]
    __shared__ __attribute__((aligned(16))) bf16_t er[2 * TILE];        // [buffer] R[e][m]: reverse state behind a chunk
    __shared__ __attribute__((aligned(16))) bf16_t es[TILE];            // S[e][m]: forward state in front of the chunk
    __shared__ __attribute__((aligned(16))) bf16_t wt[TILE];            // masked W^T [j][i]
    __shared__ __attribute__((aligned(16))) bf16_t at[TILE];            // masked A^T [j][i]
    __shared__ __attribute__((aligned(16))) bf16_t ok_[TILE];           // dK tile [j][e]
    __shared__ __attribute__((aligned(16))) bf16_t ov[TILE];            // dV tile [j][m]
    __shared__ __attribute__((aligned(16))) bf16_t oq[TILE];            // dQ tile [i][e]
    __shared__ float dd[2][C];
    __shared__ float r1s[2][C];                                         // r1[e] behind a chunk (see er)
    __shared__ float ksm[C];                                            // ksum[e] in front of the chunk (see es)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool kvg = w8 < 4;                       // KV group / Q group
    const int w = w8 & 3, wi = w >> 1, wj = w & 1;
    const int l31 = lane & 31, hf = lane >> 5;
    const int sid = stream_of_block(blockIdx.x, gridDim.x / H, H);
    const int n = sid / H, h = sid % H;
    const bf16_t* qb = q + ((long)n * L) * ldq + h * D;
    const bf16_t* kb = k + ((long)n * L) * ldk + h * D;
    const bf16_t* vb = v + ((long)n * L) * ldv + h * D;
    const bf16_t* ob = out + ((long)n * L) * ldo + h * D;
    const bf16_t* gb = dout + ((long)n * L) * lddo + h * D;
    const float* zb = zinv + ((long)n * L) * H + h;
    bf16_t* dqb = dq + ((long)n * L) * lddq + h * D;
    bf16_t* dkb = dk + ((long)n * L) * lddk + h * D;
    bf16_t* dvb = dv + ((long)n * L) * lddv + h * D;

    const int srow = tid >> 3, scol = (tid & 7) * 8;   // staging: one 16-byte slot of every stream per thread and chunk
    const int orow = (tid & 255) >> 3;                 // output: rows orow, orow + 32 of dK and dV (KV group) / dQ (Q group)
    const int nch = (L + C - 1) / C;

    const __amdgpu_buffer_rsrc_t qr = make_rsrc(qb, (uint32_t)(((long)(L - 1) * ldq + D) * 2));
    const __amdgpu_buffer_rsrc_t kr = make_rsrc(kb, (uint32_t)(((long)(L - 1) * ldk + D) * 2));
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(vb, (uint32_t)(((long)(L - 1) * ldv + D) * 2));
    const __amdgpu_buffer_rsrc_t gr = make_rsrc(gb, (uint32_t)(((long)(L - 1) * lddo + D) * 2));
    const __amdgpu_buffer_rsrc_t orr = make_rsrc(ob, (uint32_t)(((long)(L - 1) * ldo + D) * 2));
    const __amdgpu_buffer_rsrc_t zr = make_rsrc(zb, (uint32_t)(((long)(L - 1) * H + 1) * 4));

    auto load = [&](ChunkRegs& R, int c) __attribute__((always_inline)) {
        const uint32_t row = (uint32_t)c * C + srow;
        R.q = buf_load16(qr, (row * (uint32_t)ldq + scol) * 2);
        R.k = buf_load16(kr, (row * (uint32_t)ldk + scol) * 2);
        R.v = buf_load16(vr, (row * (uint32_t)ldv + scol) * 2);
        R.g = buf_load16(gr, (row * (uint32_t)lddo + scol) * 2);
        R.o = buf_load16(orr, (row * (uint32_t)ldo + scol) * 2);
        R.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zr, (int)(row * (uint32_t)H * 4), 0, 0));
    };
    // registers of chunk c -> LDS buffer b
    auto stage = [&](const ChunkRegs& R, int c, int b) __attribute__((always_inline)) {
        bf16_t* t = in_ + b * 4 * TILE;
        float dden;
        const uint4 gp = stage_g(R.g, R.o, R.z, dden);
        put_row(t + 3 * TILE, srow, scol, gp);
        if ((tid & 7) == 0) {
            dd[b][srow] = dden;
            const __bf16 hh = (__bf16)dden;
            const __bf16 ll = (__bf16)(dden - (float)hh);
            const uint32_t hl = (uint32_t)__builtin_bit_cast(unsigned short, hh) |
                                ((uint32_t)__builtin_bit_cast(unsigned short, ll) << 16);
            put_row(t + 3 * TILE, srow, 64, make_uint4(hl, 0u, 0u, 0u));     // columns 64..71: dden hi, lo, zeros
        }
        put_row(t + 2 * TILE, srow, scol, R.v);
        float x[8], y[8];
        unpack8(R.k, x);
        unpack8(R.q, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = phi(x[j]);
            y[j] = phi(y[j]);
        }
        if ((c + 1) * C > L && c * C + srow >= L) {      // rows past the end of a ragged last chunk: phi(0) = 1 is not 0
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = y[j] = 0.f;
        }
        put_row(t + 1 * TILE, srow, scol, pack8(x));
        put_row(t, srow, scol, pack8(y));
    };

    // column sums over this stream (bias gradients of the projections): KV group bs0 = dK, bs1 = dV; Q group bs0 = dQ
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    f32x8 bs0 = 0.f, bs1 = 0.f;
    auto widen = [](uint4 r) __attribute__((always_inline)) {
        f32x8 x;
        x[0] = __uint_as_float(r.x << 16); x[1] = __uint_as_float(r.x & 0xffff0000u);
        x[2] = __uint_as_float(r.y << 16); x[3] = __uint_as_float(r.y & 0xffff0000u);
        x[4] = __uint_as_float(r.z << 16); x[5] = __uint_as_float(r.z & 0xffff0000u);
        x[6] = __uint_as_float(r.w << 16); x[7] = __uint_as_float(r.w & 0xffff0000u);
        return x;
    };
    // finished output tiles of chunk c -> global
    const bf16_t* t0 = kvg ? ok_ : oq;
    bf16_t* g0 = kvg ? dkb : dqb;
    const long l0 = kvg ? lddk : lddq;
    auto store = [&](int c, f32x8& b0, f32x8& b1) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = orow + 32 * it;
            const long grow = (long)c * C + row;
            if (grow < L) {
                const uint4 a = *reinterpret_cast<const uint4*>(t0 + row * LD + scol);
                b0 += widen(a);
                *reinterpret_cast<uint4*>(g0 + grow * l0 + scol) = a;
                if (kvg) {
                    const uint4 b = *reinterpret_cast<const uint4*>(ov + row * LD + scol);
                    b1 += widen(b);
                    *reinterpret_cast<uint4*>(dvb + grow * lddv + scol) = b;
                }
            }
        }
    };

    // running states, one 32 x 32 tile per wave (wi = m-half, wj = e-half), f32 for the whole sequence:
    //   KV group: st[m][e] = R[32wj+e][32wi+m], R = sum over LATER tokens of qf g^T; waves wi == 0 also sa rows 0 / 1 =
    //             the hi / lo parts of r1 = sum of dden_i qf_i.
    //   Q group:  st[m][e] = S[32wj+e][32wi+m], S = sum over EARLIER tokens of kf v^T; waves wi == 0 also sa row 0 = ksum.
    // The products read them from LDS (er / es, bf16; r1s / ksm), where every wave finds all four tiles.
    f32x16 st = zero16(), sa = zero16();
    if (!kvg) {
        const float* f = fin + (long)sid * FIN_FLOATS;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = f[(32 * wi + acc_row(r, hf)) * 64 + 32 * wj + l31];
        if (wi == 0) sa[0] = hf == 0 ? f[64 * 64 + 32 * wj + l31] : 0.f;
    } else {
        put_acc_T_all(er, 32 * wj + l31, 32 * wi, st, hf);     // R = 0 behind the last chunk
        if (wi == 0 && hf == 0) r1s[0][32 * wj + l31] = 0.f;
    }
    const bf16x8 ones0 = ones_if(l31 == 0);
    const bf16x8 one0 = first_if(hf == 0, 1.0f);

    // b: the buffer chunk c was staged in = the parity of its distance from the last chunk (a literal at both call sites)
    auto iter = [&](int c, ChunkRegs& R, const int b) __attribute__((always_inline)) {
        if (c < nch - 1) store(c + 1, bs0, bs1);
        if (c >= 1) {
            stage(R, c - 1, 1 - b);
            if (c >= 3) load(R, c - 3);
        }
        const bf16_t* qs = in_ + b * 4 * TILE;
        const bf16_t* ks = qs + TILE;
        const bf16_t* vs = qs + 2 * TILE;
        const bf16_t* gs = qs + 3 * TILE;
        const float dden_i = dd[b][32 * wi + l31];
        __builtin_amdgcn_sched_barrier(0);      // keep the fragment reads of a product next to it: registers are short
        // ---- phase 1: score tiles (rows i on regs, cols j on lanes; kept where i >= j, written transposed) and states
        if (kvg) {
            if (!(wi == 0 && wj == 1)) {
                f32x16 W = prod_rows4(gs, 32 * wi + l31, vs, 32 * wj + l31, hf);
                W = mfma(first_if(hf == 0, dden_i), one0, W);                        // + dden_i
                if (wi == wj) put_acc_T_ge(wt, 32 * wj + l31, 32 * wi, W, hf, l31);
                else put_acc_T_all(wt, 32 * wj + l31, 32 * wi, W, hf);
            }
            __builtin_amdgcn_sched_barrier(0);
            // st[m][e] += sum_i g[i][32wi+m] qf[i][32wj+e] ;  sa[0 / 1][e] += sum_i dden_i (hi / lo) qf[i][32wj+e]
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 bq = tfrag8(qs, 16 * s, 32 * wj, lane);
                st = mfma(tfrag8(gs, 16 * s, 32 * wi, lane), bq, st);
                if (wi == 0) sa = mfma(tfrag8(gs, 16 * s, 64, lane), bq, sa);        // g columns 64, 65 = dden hi, lo
            }
            if (c > 0) {                                  // R and r1 behind chunk c - 1, for the next iteration
                put_acc_T_all(er + (1 - b) * TILE, 32 * wj + l31, 32 * wi, st, hf);
                if (wi == 0 && hf == 0) r1s[1 - b][32 * wj + l31] = sa[0] + sa[1];
            }
        } else {
            if (!(wi == 0 && wj == 1)) {
                const f32x16 A = prod_rows4(qs, 32 * wi + l31, ks, 32 * wj + l31, hf);
                if (wi == wj) put_acc_T_ge(at, 32 * wj + l31, 32 * wi, A, hf, l31);
                else put_acc_T_all(at, 32 * wj + l31, 32 * wi, A, hf);
            }
            __builtin_amdgcn_sched_barrier(0);
            // leave this chunk's own tokens out of the prefix state:  st[m][e] -= sum_j v[j][32wi+m] kf[j][32wj+e]
            if (c == 0) {
                st = zero16();
                sa = zero16();
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 nb = negate8(tfrag8(ks, 16 * s, 32 * wj, lane));
                    st = mfma(tfrag8(vs, 16 * s, 32 * wi, lane), nb, st);
                    if (wi == 0) sa = mfma(ones0, nb, sa);
                }
            }
            put_acc_T_all(es, 32 * wj + l31, 32 * wi, st, hf);
            if (wi == 0 && hf == 0) ksm[32 * wj + l31] = sa[0];
        }
        __syncthreads();
        // ---- phase 2: the products
        if (kvg) {
            const bf16_t* rs = er + b * TILE;            // R behind this chunk
            // dkf^T tile (rows e on regs, cols j on lanes) = R v^T + qf^T W + r1 (x) 1 ; j-half wi: over i >= j only
            f32x16 K = prod_rows4(rs, 32 * wj + l31, vs, 32 * wi + l31, hf);
            K = mfma(first_if(hf == 0, r1s[b][32 * wj + l31]), one0, K);
            // dv^T tile (rows m on regs, cols j on lanes) = R^T kf^T + g^T A
            f32x16 V = mfma(tfrag8(rs, 0, 32 * wj, lane), row8(ks, 32 * wi + l31, 8 * hf), zero16());
#pragma unroll
            for (int s = 1; s < 4; ++s)
                V = mfma(tfrag8(rs, 16 * s, 32 * wj, lane), row8(ks, 32 * wi + l31, 16 * s + 8 * hf), V);
            __builtin_amdgcn_sched_barrier(0);
            if (wi == 0) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    K = mfma(tfrag8(qs, 16 * s, 32 * wj, lane), row8(wt, 32 * wi + l31, 16 * s + 8 * hf), K);
                    V = mfma(tfrag8(gs, 16 * s, 32 * wj, lane), row8(at, 32 * wi + l31, 16 * s + 8 * hf), V);
                }
            }
#pragma unroll
            for (int s = 2; s < 4; ++s) {
                K = mfma(tfrag8(qs, 16 * s, 32 * wj, lane), row8(wt, 32 * wi + l31, 16 * s + 8 * hf), K);
                V = mfma(tfrag8(gs, 16 * s, 32 * wj, lane), row8(at, 32 * wi + l31, 16 * s + 8 * hf), V);
            }
            __builtin_amdgcn_sched_barrier(0);
            put_acc_T_dphi(ok_, ks, 32 * wi + l31, 32 * wj, K, hf);
            put_acc_T_all(ov, 32 * wi + l31, 32 * wj, V, hf);
        } else {
            // dqf^T tile (rows e on regs, cols i on lanes) = S g^T + kf^T W^T + ksum (x) dden ; i-half wi: over j <= i only
            f32x16 Q = prod_rows4(es, 32 * wj + l31, gs, 32 * wi + l31, hf);
            Q = mfma(first_if(hf == 0, ksm[32 * wj + l31]), first_if(hf == 0, dden_i), Q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 2; ++s)
                Q = mfma(tfrag8(ks, 16 * s, 32 * wj, lane), tfrag8(wt, 16 * s, 32 * wi, lane), Q);
            if (wi == 1) {
#pragma unroll
                for (int s = 2; s < 4; ++s)
                    Q = mfma(tfrag8(ks, 16 * s, 32 * wj, lane), tfrag8(wt, 16 * s, 32 * wi, lane), Q);
            }
            put_acc_T_dphi(oq, qs, 32 * wi + l31, 32 * wj, Q, hf);
        }
        __syncthreads();
    };

    ChunkRegs RA, RB;
    load(RA, nch - 1);
    if (nch > 1) load(RB, nch - 2);
    stage(RA, nch - 1, 0);
    if (nch > 2) load(RA, nch - 3);
    __syncthreads();
    for (int c = nch - 1; c >= 0; c -= 2) {
        iter(c, RB, 0);
        if (c >= 1) iter(c - 1, RA, 1);
    }
    store(0, bs0, bs1);

    // column sums over the stream: each wave covers 16 of the rows, four waves per group
    if (csum_q) {
        float* scratch = reinterpret_cast<float*>(wt);     // 3 quantities x 4 waves x 64 columns; score tiles are dead
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = bs0[j], y = bs1[j];
            x += __shfl_xor(x, 8, 64);
            y += __shfl_xor(y, 8, 64);
            x += __shfl_xor(x, 16, 64);
            y += __shfl_xor(y, 16, 64);
            x += __shfl_xor(x, 32, 64);
            y += __shfl_xor(y, 32, 64);
            if ((lane >> 3) == 0) {
                scratch[((kvg ? 1 : 0) * 4 + w) * 64 + (lane & 7) * 8 + j] = x;
                if (kvg) scratch[(2 * 4 + w) * 64 + (lane & 7) * 8 + j] = y;
            }
        }
        __syncthreads();
        if (tid < 192) {
            const int a = tid >> 6, e = tid & 63;
            const float s = (scratch[(a * 4 + 0) * 64 + e] + scratch[(a * 4 + 1) * 64 + e]) +
                            (scratch[(a * 4 + 2) * 64 + e] + scratch[(a * 4 + 3) * 64 + e]);
            float* dst = a == 0 ? csum_q : a == 1 ? csum_k : csum_v;
            dst[(long)sid * D + e] = s;
        }
    }
}

}  // namespace b16

// Segment count for few-stream launches: 1 (one workgroup per stream) once N * H fills the chip; otherwise enough
// segments for ~2 workgroups per CU, at least two chunks each, at most 16.
int scan_segments(int N, int H, int L) {
    const long streams = (long)N * H;
    const int nch = (L + b16::C - 1) / b16::C;
    if (streams >= 256 || nch < 4) return 1;
    long want = (512 + streams - 1) / streams;
    if (want > 16) want = 16;
    if (want > nch / 2) want = nch / 2;
    if (want < 2) return 1;
    const int cps = (nch + (int)want - 1) / (int)want;
    return (nch + cps - 1) / cps;                      // no empty segment
}
static int seg_cps(int L, int P) { return (((L + b16::C - 1) / b16::C) + P - 1) / P; }

long scan_seg_floats(int N, int H, int P, int backward) {
    return P > 1 ? (long)N * H * P * 6 * (backward ? 8 : 3) * 1024 : 0;
}

long scan_final_state_floats(int N, int H) { return (long)N * H * b16::FIN_FLOATS; }

int launch_cla_fwd_bf16(const void* q, const void* k, const void* v, void* out, float* zinv, int N, int H, int L,
                        long ldq, long ldk, long ldv, long ldo, float eps, int P, float* ws, float* fin,
                        hipStream_t st) {
    const int nch = (L + b16::C - 1) / b16::C;
    if (P <= 1) {
        hipLaunchKernelGGL(b16::cla_fwd_bf16_kernel<false>, dim3(N * H), dim3(256), 0, st, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, zinv, H, L, ldq, ldk, ldv, ldo, eps, 1, nch,
                           (const float*)nullptr, (float*)nullptr, fin);
        return (int)hipGetLastError();
    }
    const int cps = seg_cps(L, P);
    const long NS = (long)N * H;
    float* part = ws;
    float* pre = ws + NS * P * 4 * 3 * 1024;
    hipLaunchKernelGGL(b16::cla_fwd_bf16_kernel<true>, dim3(NS * P), dim3(256), 0, st, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, zinv, H, L, ldq, ldk, ldv, ldo, eps, P, cps,
                       (const float*)nullptr, part, (float*)nullptr);
    const long total = NS * 2 * 3 * 1024;
    hipLaunchKernelGGL(b16::seg_prefix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, pre, total,
                       P, 3, 3);
    hipLaunchKernelGGL(b16::cla_fwd_bf16_kernel<false>, dim3(NS * P), dim3(256), 0, st, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, zinv, H, L, ldq, ldk, ldv, ldo, eps, P, cps,
                       (const float*)pre, (float*)nullptr, (float*)nullptr);
    return (int)hipGetLastError();
}

// whole-sequence backward in one sweep; `fin` = the final state launch_cla_fwd_bf16 wrote for these q, k, v
int launch_cla_bwd_sweep_bf16(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                              const void* dout, const float* fin, void* dq, void* dk, void* dv, float* csum_q,
                              float* csum_k, float* csum_v, int N, int H, int L, long ldq, long ldk, long ldv, long ldo,
                              long lddo, long lddq, long lddk, long lddv, hipStream_t st) {
    hipLaunchKernelGGL(b16::cla_bwd_sweep_bf16_kernel, dim3(N * H), dim3(512), 0, st, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)out, (const bf16_t*)dout, zinv, fin,
                       (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, csum_q, csum_k, csum_v, H, L, ldq, ldk, ldv, ldo, lddo,
                       lddq, lddk, lddv);
    return (int)hipGetLastError();
}

// P > 1: `ws` must be the workspace launch_cla_bwd_dkdv_bf16 has just filled (it computes the prefix states of both scans)
int launch_cla_bwd_dq_bf16(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                           const void* dout, const float* dden, void* dq, float* csum, int N, int H, int L, long ldq,
                           long ldk, long ldv, long ldo, long lddo, long lddq, int P, float* ws, hipStream_t st) {
    const int nch = (L + b16::C - 1) / b16::C;
    const long NS = (long)N * H;
    const int cps = P > 1 ? seg_cps(L, P) : nch;
    const float* pre = P > 1 ? ws + NS * P * 4 * 8 * 1024 : nullptr;
    if (P < 1) P = 1;
    if (dden)
        hipLaunchKernelGGL(b16::cla_bwd_dq_bf16_kernel<true>, dim3(NS * P), dim3(256), 0, st, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)out, (const bf16_t*)dout, zinv, dden,
                           (bf16_t*)dq, csum, H, L, ldq, ldk, ldv, ldo, lddo, lddq, P, cps, pre);
    else
        hipLaunchKernelGGL(b16::cla_bwd_dq_bf16_kernel<false>, dim3(NS * P), dim3(256), 0, st, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)out, (const bf16_t*)dout, zinv, dden,
                           (bf16_t*)dq, csum, H, L, ldq, ldk, ldv, ldo, lddo, lddq, P, cps, pre);
    return (int)hipGetLastError();
}

int launch_cla_bwd_dkdv_bf16(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                             const void* dout, void* dk, void* dv, float* csum_k, float* csum_v, float* dden_out,
                             int N, int H, int L, long ldq, long ldk, long ldv, long ldo, long lddo, long lddk,
                             long lddv, int P, float* ws, hipStream_t st) {
    const int nch = (L + b16::C - 1) / b16::C;
    const long NS = (long)N * H;
    if (P <= 1) {
        hipLaunchKernelGGL(b16::cla_bwd_dkdv_bf16_kernel<false>, dim3(NS), dim3(256), 0, st, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)out, (const bf16_t*)dout, zinv,
                           (bf16_t*)dk, (bf16_t*)dv, csum_k, csum_v, dden_out, H, L, ldq, ldk, ldv, ldo, lddo, lddk,
                           lddv, 1, nch, (const float*)nullptr, (float*)nullptr);
        return (int)hipGetLastError();
    }
    const int cps = seg_cps(L, P);
    float* part = ws;
    float* pre = ws + NS * P * 4 * 8 * 1024;
    hipLaunchKernelGGL(b16::cla_bwd_dkdv_bf16_kernel<true>, dim3(NS * P), dim3(256), 0, st, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)out, (const bf16_t*)dout, zinv, (bf16_t*)dk,
                       (bf16_t*)dv, (float*)nullptr, (float*)nullptr, (float*)nullptr, H, L, ldq, ldk, ldv, ldo, lddo,
                       lddk, lddv, P, cps, (const float*)nullptr, part);
    const long total = NS * 2 * 8 * 1024;
    hipLaunchKernelGGL(b16::seg_prefix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, pre, total,
                       P, 8, 3);
    hipLaunchKernelGGL(b16::cla_bwd_dkdv_bf16_kernel<false>, dim3(NS * P), dim3(256), 0, st, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)out, (const bf16_t*)dout, zinv, (bf16_t*)dk,
                       (bf16_t*)dv, csum_k, csum_v, dden_out, H, L, ldq, ldk, ldv, ldo, lddo, lddk, lddv, P, cps,
                       (const float*)pre, (float*)nullptr);
    return (int)hipGetLastError();
}

}  // namespace cwlt
