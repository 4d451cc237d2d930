// Causal linear attention (CLA) for gfx950 -- forward and backward.
//
// Replaces pytorch-fast-transformers==0.4.0 `causal_dot_product` (+ the PyTorch ops around it in
// `CausalLinearAttention.forward`: elu(x)+1 feature map, K.cumsum normaliser, V*Z) as reached from
// /root/reference/dqn_policy/model.py:128-137,231-232 and ppo_policy/model.py:129-138,313-321.
//
//   qf = phi(Q), kf = phi(K), phi(x) = elu(x)+1
//   den_l = qf_l . sum_{j<=l} kf_j + eps
//   out_l = (qf_l . sum_{j<=l} kf_j (x) v_j) / den_l
//
// Layout (MI355X-first, no permutes): Q, K, V, out are read/written in the projection GEMM's own
// (N, L, H, 64) layout -- token row stride `ld*` elements, head h at column h*64 -- so the kernels
// sit directly between the QKV GEMM and the out-projection GEMM with no copies.
//
// Algorithm: chunked scan, chunk = 32 tokens.  Per chunk the intra-chunk part is a masked 32x32
// (qf kf^T) product and the inter-chunk part goes through the running 64x64 state, which lives in
// MFMA accumulator registers for the whole sequence (never touches LDS or HBM).  All products are
// v_mfma_f32_32x32x2_f32 (exact f32 fma chains), operands staged through LDS tiles with an odd row
// stride (65 floats) so both row-type and column-type ds_read_b32 operand fetches are conflict-free.
//
// Work split: one workgroup per (n, h) stream.  Forward: 2 waves, wave w owns value columns
// [32w, 32w+32).  Backward: `dq` kernel (forward scan, 2 waves = the two halves of the query
// feature dim) and `dkdv` kernel (reverse scan, 4 waves = dk halves + dv halves).  Every wave runs
// exactly 112 MFMAs per chunk, so the waves of a workgroup stay in step between barriers.
#include "cwlt_common.h"

namespace cwlt {

constexpr int D = 64;    // head dim: E = M = 64 (reference: d_model 512 / 8 heads, config.py:11-15)
constexpr int C = 32;    // tokens per chunk
constexpr int LDT = 65;  // LDS row stride of the 32x64 operand tiles
constexpr int LDA = 33;  // LDS row stride of the 32x32 intra-chunk score tiles

// elu(x)+1 exactly as the reference evaluates it: (exp(x)-1)+1 on the negative side.
__device__ __forceinline__ float phi(float x) { return x > 0.f ? x + 1.f : (expf(x) - 1.f) + 1.f; }
__device__ __forceinline__ float dphi(float x) { return x > 0.f ? 1.f : expf(x); }

// accumulator register r of lane-half hf -> row of the 32x32 tile (column = lane & 31)
__device__ __forceinline__ constexpr int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// X[i][j] = sum_e a_t[i][e] * b_t[j][e]   (both tiles row-major 32x64, stride LDT) -> 32 MFMAs
__device__ __forceinline__ f32x16 prod_rows(const float* a_t, const float* b_t, int l31, int hf) {
    f32x16 acc = zero16();
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
        float a = a_t[l31 * LDT + 2 * s + hf];
        float b = b_t[l31 * LDT + 2 * s + hf];
        acc = mfma(a, b, acc);
    }
    return acc;
}

// acc[i][n] += sum_j x[i][j] * t[j][col0+n]   (x: 32x32 stride LDA, row-type read) -> 16 MFMAs
__device__ __forceinline__ f32x16 prod_x_t(f32x16 acc, const float* x, const float* t, int col0, int l31, int hf) {
#pragma unroll 8
    for (int s = 0; s < 16; ++s) {
        float a = x[l31 * LDA + 2 * s + hf];
        float b = t[(2 * s + hf) * LDT + col0 + l31];
        acc = mfma(a, b, acc);
    }
    return acc;
}

// acc[j][n] += sum_i x[i][j] * t[i][col0+n]   (x read column-type = x^T) -> 16 MFMAs
__device__ __forceinline__ f32x16 prod_xT_t(f32x16 acc, const float* x, const float* t, int col0, int l31, int hf) {
#pragma unroll 8
    for (int s = 0; s < 16; ++s) {
        float a = x[(2 * s + hf) * LDA + l31];
        float b = t[(2 * s + hf) * LDT + col0 + l31];
        acc = mfma(a, b, acc);
    }
    return acc;
}

// acc[i][n] += sum_{k in 32t..32t+31} a_t[i][k] * S[k][n], S held in accumulator layout (rows on regs)
__device__ __forceinline__ f32x16 prod_state(f32x16 acc, const float* a_t, int k0, const f32x16& S, int l31, int hf) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a = a_t[l31 * LDT + k0 + acc_row(r, hf)];
        acc = mfma(a, S[r], acc);
    }
    return acc;
}

// S0[p][n] += sum_j a_t[j][p] * b_t[j][col0+n],  S1[p][n] += sum_j a_t[j][32+p] * b_t[j][col0+n]  -> 32 MFMAs
__device__ __forceinline__ void update_state(f32x16& S0, f32x16& S1, const float* a_t, const float* b_t, int col0,
                                             int l31, int hf) {
#pragma unroll 8
    for (int s = 0; s < 16; ++s) {
        float b = b_t[(2 * s + hf) * LDT + col0 + l31];
        float a0 = a_t[(2 * s + hf) * LDT + l31];
        float a1 = a_t[(2 * s + hf) * LDT + 32 + l31];
        S0 = mfma(a0, b, S0);
        S1 = mfma(a1, b, S1);
    }
}

__device__ __forceinline__ void write4(float* t, int row, int col, float4 x) {
    float* p = t + row * LDT + col;
    p[0] = x.x; p[1] = x.y; p[2] = x.z; p[3] = x.w;
}
__device__ __forceinline__ float4 phi4(float4 x, bool valid) {
    if (!valid) return make_float4(0.f, 0.f, 0.f, 0.f);
    return make_float4(phi(x.x), phi(x.y), phi(x.z), phi(x.w));
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void cla_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                      const T* __restrict__ v, T* __restrict__ o,
                                                      float* __restrict__ zinv, int H, int L, long ldq, long ldk,
                                                      long ldv, long ldo, float eps) {
    __shared__ float qs[C * LDT];
    __shared__ float ks[C * LDT];
    __shared__ float vs[C * LDT];
    __shared__ float as[2][C * LDA];
    __shared__ float ksum[2][D];
    __shared__ float zs[2][C];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    const int n = blockIdx.x / H, h = blockIdx.x % H;
    const T* qb = q + ((long)n * L) * ldq + h * D;
    const T* kb = k + ((long)n * L) * ldk + h * D;
    const T* vb = v + ((long)n * L) * ldv + h * D;
    T* ob = o + ((long)n * L) * ldo + h * D;
    float* zb = zinv + ((long)n * L) * H + h;

    const int srow = tid >> 4, scol = (tid & 15) * 4;
    const int nch = (L + C - 1) / C;
    const float4 f4z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rq[4], rk[4], rv[4];

#define CLA_FWD_LOAD(c)                                         \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {          \
        const long row = (long)(c) * C + srow + 8 * it;         \
        const bool ok = row < L;                                \
        rq[it] = ok ? load4(qb + row * ldq + scol) : f4z;       \
        rk[it] = ok ? load4(kb + row * ldk + scol) : f4z;       \
        rv[it] = ok ? load4(vb + row * ldv + scol) : f4z;       \
    }

    CLA_FWD_LOAD(0);
    if (tid < D) ksum[0][tid] = 0.f;
    f32x16 S0 = zero16(), S1 = zero16();  // S[e][m], e-half 0/1, m in this wave's half

    for (int c = 0; c < nch; ++c) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it;
            const bool ok = c * C + row < L;
            write4(qs, row, scol, phi4(rq[it], ok));
            write4(ks, row, scol, phi4(rk[it], ok));
            write4(vs, row, scol, rv[it]);
        }
        __syncthreads();
        if (c + 1 < nch) { CLA_FWD_LOAD(c + 1); }

        // intra-chunk scores A = qf kf^T, causal-masked
        f32x16 A = prod_rows(qs, ks, l31, hf);
        float* aw = as[w];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = acc_row(r, hf);
            aw[i * LDA + l31] = (l31 <= i) ? A[r] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();

        // normaliser: rowsum(A) + qf . ksum_prev  (lane = row l31; the two lane halves split the sums)
        float den = 0.f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) den += aw[l31 * LDA + hf * 16 + jj];
        const float* kp = ksum[c & 1];
#pragma unroll 8
        for (int e = 0; e < 32; ++e) den = fmaf(qs[l31 * LDT + hf * 32 + e], kp[hf * 32 + e], den);
        den += __shfl_xor(den, 32, 64);
        const float z = 1.0f / (den + eps);
        if (hf == 0) {
            zs[w][l31] = z;
            if (w == 0 && c * C + l31 < L) zb[((long)c * C + l31) * H] = z;
        }
        // running key sum for the next chunk (wave w owns e in [32w, 32w+32))
        {
            float ksn = 0.f;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) ksn += ks[(hf * 16 + jj) * LDT + 32 * w + l31];
            ksn += __shfl_xor(ksn, 32, 64);
            if (hf == 0) ksum[(c + 1) & 1][32 * w + l31] = kp[32 * w + l31] + ksn;
        }
        __builtin_amdgcn_wave_barrier();

        // numerator: A v + qf S_prev
        f32x16 O = zero16();
        O = prod_x_t(O, aw, vs, 32 * w, l31, hf);
        O = prod_state(O, qs, 0, S0, l31, hf);
        O = prod_state(O, qs, 32, S1, l31, hf);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = acc_row(r, hf);
            const long row = (long)c * C + i;
            if (row < L) store1(ob + row * ldo + 32 * w + l31, O[r] * zs[w][i]);
        }
        // state: S += kf^T v
        update_state(S0, S1, ks, vs, 32 * w, l31, hf);
        __syncthreads();
    }
#undef CLA_FWD_LOAD
}

// ------------------------------------------------------------------------------------------------
// backward, dQ: forward scan.  wave w owns query-feature columns [32w, 32w+32).
//   g_i = dout_i * z_i ; dden_i = -(dout_i . out_i) z_i ; W_ij = g_i . v_j + dden_i (j <= i)
//   dqf_i = sum_{j<=i} W_ij kf_j  =  (W kf)_i + g_i S_prev^T + dden_i ksum_prev ;  dQ = dqf * phi'(Q)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void cla_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ v, const T* __restrict__ out,
                                                         const T* __restrict__ dout, const float* __restrict__ zinv,
                                                         T* __restrict__ dq, int H, int L, long ldq, long ldk,
                                                         long ldv, long ldo, long lddo, long lddq) {
    __shared__ float gs[C * LDT];
    __shared__ float vs[C * LDT];
    __shared__ float ks[C * LDT];
    __shared__ float ws[2][C * LDA];
    __shared__ float dd[C];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    const int n = blockIdx.x / H, h = blockIdx.x % H;
    const T* qb = q + ((long)n * L) * ldq + h * D;
    const T* kb = k + ((long)n * L) * ldk + h * D;
    const T* vb = v + ((long)n * L) * ldv + h * D;
    const T* ob = out + ((long)n * L) * ldo + h * D;
    const T* gb = dout + ((long)n * L) * lddo + h * D;
    const float* zb = zinv + ((long)n * L) * H + h;
    T* dqb = dq + ((long)n * L) * lddq + h * D;

    const int srow = tid >> 4, scol = (tid & 15) * 4;
    const int nch = (L + C - 1) / C;
    const float4 f4z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rk[4], rv[4], rg[4], ro[4];
    float rz[4];

#define CLA_DQ_LOAD(c)                                          \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {          \
        const long row = (long)(c) * C + srow + 8 * it;         \
        const bool ok = row < L;                                \
        rk[it] = ok ? load4(kb + row * ldk + scol) : f4z;       \
        rv[it] = ok ? load4(vb + row * ldv + scol) : f4z;       \
        rg[it] = ok ? load4(gb + row * lddo + scol) : f4z;      \
        ro[it] = ok ? load4(ob + row * ldo + scol) : f4z;       \
        rz[it] = ok ? zb[row * H] : 0.f;                        \
    }

    CLA_DQ_LOAD(0);
    f32x16 ST0 = zero16(), ST1 = zero16();  // ST_t[m'][e'] = S[32w+e'][32t+m']
    float zp = 0.f;                         // ksum_prev[32w + l31]

    for (int c = 0; c < nch; ++c) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = srow + 8 * it;
            const bool ok = c * C + row < L;
            const float z = rz[it];
            float dot = rg[it].x * ro[it].x + rg[it].y * ro[it].y + rg[it].z * ro[it].z + rg[it].w * ro[it].w;
            dot += __shfl_xor(dot, 1, 64);
            dot += __shfl_xor(dot, 2, 64);
            dot += __shfl_xor(dot, 4, 64);
            dot += __shfl_xor(dot, 8, 64);
            if ((tid & 15) == 0) dd[row] = -dot * z;
            write4(gs, row, scol, make_float4(rg[it].x * z, rg[it].y * z, rg[it].z * z, rg[it].w * z));
            write4(vs, row, scol, rv[it]);
            write4(ks, row, scol, phi4(rk[it], ok));
        }
        __syncthreads();
        // raw Q in accumulator layout (for phi'), issued early so it lands under the MFMAs
        float qraw[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = (long)c * C + acc_row(r, hf);
            qraw[r] = row < L ? load1(qb + row * ldq + 32 * w + l31) : 0.f;
        }
        if (c + 1 < nch) { CLA_DQ_LOAD(c + 1); }

        f32x16 W = prod_rows(gs, vs, l31, hf);
        float* xw = ws[w];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = acc_row(r, hf);
            xw[i * LDA + l31] = (l31 <= i) ? W[r] + dd[i] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();

        f32x16 DQ = zero16();
        DQ = prod_x_t(DQ, xw, ks, 32 * w, l31, hf);
        DQ = prod_state(DQ, gs, 0, ST0, l31, hf);
        DQ = prod_state(DQ, gs, 32, ST1, l31, hf);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = acc_row(r, hf);
            const long row = (long)c * C + i;
            const float val = fmaf(dd[i], zp, DQ[r]) * dphi(qraw[r]);
            if (row < L) store1(dqb + row * lddq + 32 * w + l31, val);
        }
        // ST_t += v[:, t-half]^T kf[:, w-half]
        update_state(ST0, ST1, vs, ks, 32 * w, l31, hf);
        {
            float s = 0.f;
#pragma unroll 8
            for (int j = 0; j < C; ++j) s += ks[j * LDT + 32 * w + l31];
            zp += s;
        }
        __syncthreads();
    }
#undef CLA_DQ_LOAD
}

// ------------------------------------------------------------------------------------------------
// backward, dK and dV: reverse scan.  waves 0,1: dK feature halves; waves 2,3: dV value halves.
//   dkf_j = sum_{i>=j} W_ij qf_i = (W^T qf)_j + v_j R_next^T + r1_next ;  dK = dkf * phi'(K)
//   dv_j  = sum_{i>=j} A_ij g_i  = (A^T g)_j + kf_j R_next
//   R = sum_{i later} qf_i (x) g_i,  r1 = sum_{i later} qf_i dden_i
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cla_bwd_dkdv_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                           const T* __restrict__ v, const T* __restrict__ out,
                                                           const T* __restrict__ dout, const float* __restrict__ zinv,
                                                           T* __restrict__ dk, T* __restrict__ dv, int H, int L,
                                                           long ldq, long ldk, long ldv, long ldo, long lddo,
                                                           long lddk, long lddv) {
    __shared__ float qs[C * LDT];
    __shared__ float ks[C * LDT];
    __shared__ float vs[C * LDT];
    __shared__ float gs[C * LDT];
    __shared__ float xs[4][C * LDA];
    __shared__ float dd[C];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    const int n = blockIdx.x / H, h = blockIdx.x % H;
    const T* qb = q + ((long)n * L) * ldq + h * D;
    const T* kb = k + ((long)n * L) * ldk + h * D;
    const T* vb = v + ((long)n * L) * ldv + h * D;
    const T* ob = out + ((long)n * L) * ldo + h * D;
    const T* gb = dout + ((long)n * L) * lddo + h * D;
    const float* zb = zinv + ((long)n * L) * H + h;
    T* dkb = dk + ((long)n * L) * lddk + h * D;
    T* dvb = dv + ((long)n * L) * lddv + h * D;

    const int srow = tid >> 4, scol = (tid & 15) * 4;  // srow 0..15
    const int nch = (L + C - 1) / C;
    const float4 f4z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rq[2], rk[2], rv[2], rg[2], ro[2];
    float rz[2];

#define CLA_KV_LOAD(c)                                          \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {          \
        const long row = (long)(c) * C + srow + 16 * it;        \
        const bool ok = row < L;                                \
        rq[it] = ok ? load4(qb + row * ldq + scol) : f4z;       \
        rk[it] = ok ? load4(kb + row * ldk + scol) : f4z;       \
        rv[it] = ok ? load4(vb + row * ldv + scol) : f4z;       \
        rg[it] = ok ? load4(gb + row * lddo + scol) : f4z;      \
        ro[it] = ok ? load4(ob + row * ldo + scol) : f4z;       \
        rz[it] = ok ? zb[row * H] : 0.f;                        \
    }

    CLA_KV_LOAD(nch - 1);
    f32x16 R0 = zero16(), R1 = zero16();
    float r1 = 0.f;
    const int half = w & 1;

    for (int c = nch - 1; c >= 0; --c) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = srow + 16 * it;
            const bool ok = c * C + row < L;
            const float z = rz[it];
            float dot = rg[it].x * ro[it].x + rg[it].y * ro[it].y + rg[it].z * ro[it].z + rg[it].w * ro[it].w;
            dot += __shfl_xor(dot, 1, 64);
            dot += __shfl_xor(dot, 2, 64);
            dot += __shfl_xor(dot, 4, 64);
            dot += __shfl_xor(dot, 8, 64);
            if ((tid & 15) == 0) dd[row] = -dot * z;
            write4(gs, row, scol, make_float4(rg[it].x * z, rg[it].y * z, rg[it].z * z, rg[it].w * z));
            write4(vs, row, scol, rv[it]);
            write4(ks, row, scol, phi4(rk[it], ok));
            write4(qs, row, scol, phi4(rq[it], ok));
        }
        __syncthreads();
        float kraw[16];
        if (w < 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = (long)c * C + acc_row(r, hf);
                kraw[r] = row < L ? load1(kb + row * ldk + 32 * half + l31) : 0.f;
            }
        }
        if (c > 0) { CLA_KV_LOAD(c - 1); }

        float* xw = xs[w];
        if (w < 2) {
            // ---- dK, feature half `half`; R_t[m'][e'] = R[32*half+e'][32t+m'] ----
            f32x16 W = prod_rows(gs, vs, l31, hf);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = acc_row(r, hf);
                xw[i * LDA + l31] = (l31 <= i) ? W[r] + dd[i] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            f32x16 DK = zero16();
            DK = prod_xT_t(DK, xw, qs, 32 * half, l31, hf);
            DK = prod_state(DK, vs, 0, R0, l31, hf);
            DK = prod_state(DK, vs, 32, R1, l31, hf);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = (long)c * C + acc_row(r, hf);
                const float val = (DK[r] + r1) * dphi(kraw[r]);
                if (row < L) store1(dkb + row * lddk + 32 * half + l31, val);
            }
            update_state(R0, R1, gs, qs, 32 * half, l31, hf);
            float s = 0.f;
#pragma unroll 8
            for (int i = 0; i < C; ++i) s = fmaf(qs[i * LDT + 32 * half + l31], dd[i], s);
            r1 += s;
        } else {
            // ---- dV, value half `half`; R_t[e'][m'] = R[32t+e'][32*half+m'] ----
            f32x16 A = prod_rows(qs, ks, l31, hf);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = acc_row(r, hf);
                xw[i * LDA + l31] = (l31 <= i) ? A[r] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            f32x16 DV = zero16();
            DV = prod_xT_t(DV, xw, gs, 32 * half, l31, hf);
            DV = prod_state(DV, ks, 0, R0, l31, hf);
            DV = prod_state(DV, ks, 32, R1, l31, hf);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = (long)c * C + acc_row(r, hf);
                if (row < L) store1(dvb + row * lddv + 32 * half + l31, DV[r]);
            }
            update_state(R0, R1, qs, gs, 32 * half, l31, hf);
        }
        __syncthreads();
    }
#undef CLA_KV_LOAD
}

template <typename T>
static int launch_fwd(const void* q, const void* k, const void* v, void* out, float* zinv, int N, int H, int L,
                      long ldq, long ldk, long ldv, long ldo, float eps, hipStream_t st) {
    hipLaunchKernelGGL((cla_fwd_kernel<T>), dim3(N * H), dim3(128), 0, st, (const T*)q, (const T*)k, (const T*)v,
                       (T*)out, zinv, H, L, ldq, ldk, ldv, ldo, eps);
    return (int)hipGetLastError();
}

template <typename T>
static int launch_bwd_dkdv(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                           const void* dout, void* dk, void* dv, int N, int H, int L, long ldq, long ldk, long ldv,
                           long ldo, long lddo, long lddk, long lddv, hipStream_t st) {
    hipLaunchKernelGGL((cla_bwd_dkdv_kernel<T>), dim3(N * H), dim3(256), 0, st, (const T*)q, (const T*)k,
                       (const T*)v, (const T*)out, (const T*)dout, zinv, (T*)dk, (T*)dv, H, L, ldq, ldk, ldv, ldo,
                       lddo, lddk, lddv);
    return (int)hipGetLastError();
}

template <typename T>
static int launch_bwd_dq(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                         const void* dout, void* dq, int N, int H, int L, long ldq, long ldk, long ldv, long ldo,
                         long lddo, long lddq, hipStream_t st) {
    hipLaunchKernelGGL((cla_bwd_dq_kernel<T>), dim3(N * H), dim3(128), 0, st, (const T*)q, (const T*)k, (const T*)v,
                       (const T*)out, (const T*)dout, zinv, (T*)dq, H, L, ldq, ldk, ldv, ldo, lddo, lddq);
    return (int)hipGetLastError();
}

static bool bad_ld(long ld, int H) { return ld < (long)H * D || (ld & 3) != 0; }

}  // namespace cwlt

extern "C" {

/* A sequence of nch = ceil(L / 64) chunks cut into `segments` pieces of cps = ceil(nch / segments) whole chunks: every
 * piece must own at least one chunk.  A count such as 4 for 9 chunks (cps 3: the fourth piece is empty) would leave that
 * piece's state increment unwritten and the prefix / suffix pass would add workspace garbage to its neighbours. */
static int cwlt_segments_ok(int L, int segments) {
    const int nch = (L + 63) / 64;
    if (segments < 1 || segments > nch) return 0;
    const int cps = (nch + segments - 1) / segments;
    return (long)(segments - 1) * cps < nch;
}

int cwlt_scan_segments(int N, int H, int L, int dtype) {
    if (dtype != CWLT_BF16 || N <= 0 || H <= 0 || L <= 0) return 1;
    return cwlt::scan_segments(N, H, L);
}

int64_t cwlt_scan_seg_floats(int N, int H, int segments, int backward) {
    if (N <= 0 || H <= 0) return 0;
    return cwlt::scan_seg_floats(N, H, segments, backward);
}

int64_t cwlt_scan_final_state_floats(int N, int H) {
    if (N <= 0 || H <= 0) return 0;
    return cwlt::scan_final_state_floats(N, H);
}

int cwlt_causal_linear_fwd(const void* q, const void* k, const void* v, void* out, float* zinv, int N, int H,
                           int L, int head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float eps,
                           int segments, float* seg_ws, float* final_state, int dtype, void* stream) {
    using namespace cwlt;
    if (N < 0 || H <= 0 || L < 0 || head_dim != D) return CWLT_ERR_ARG;
    if (N == 0 || L == 0) return CWLT_OK;                     // empty batch / sequence: nothing to do
    if (!q || !k || !v || !out || !zinv) return CWLT_ERR_ARG;
    if (bad_ld(ldq, H) || bad_ld(ldk, H) || bad_ld(ldv, H) || bad_ld(ldo, H)) return CWLT_ERR_ARG;
    if (N == 0 || L == 0) return CWLT_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool fast = dtype == CWLT_BF16 && ((ldq | ldk | ldv | ldo) & 7) == 0;
    if (segments < 1 || (segments > 1 && (!fast || !seg_ws || !cwlt_segments_ok(L, segments)))) return CWLT_ERR_ARG;
    if (final_state && (!fast || segments != 1)) return CWLT_ERR_ARG;   // the bf16 whole-sequence kernel writes it
    if (dtype == CWLT_F32) return launch_fwd<float>(q, k, v, out, zinv, N, H, L, ldq, ldk, ldv, ldo, eps, st);
    if (dtype == CWLT_BF16) {
        if (fast)
            return launch_cla_fwd_bf16(q, k, v, out, zinv, N, H, L, ldq, ldk, ldv, ldo, eps, segments, seg_ws,
                                       final_state, st);
        return launch_fwd<bf16_t>(q, k, v, out, zinv, N, H, L, ldq, ldk, ldv, ldo, eps, st);
    }
    return CWLT_ERR_DTYPE;
}

int cwlt_causal_linear_bwd_dkdv(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                                const void* dout, void* dk, void* dv, float* colsum_k, float* colsum_v, float* dden,
                                int N, int H, int L, int head_dim,
                                int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddk,
                                int64_t lddv, int segments, float* seg_ws, int dtype, void* stream) {
    using namespace cwlt;
    if (N < 0 || H <= 0 || L < 0 || head_dim != D) return CWLT_ERR_ARG;
    if (N == 0 || L == 0) return CWLT_OK;
    if (!q || !k || !v || !out || !zinv || !dout || !dk || !dv) return CWLT_ERR_ARG;
    if (bad_ld(ldq, H) || bad_ld(ldk, H) || bad_ld(ldv, H) || bad_ld(ldo, H) || bad_ld(lddo, H) ||
        bad_ld(lddk, H) || bad_ld(lddv, H))
        return CWLT_ERR_ARG;
    if (N == 0 || L == 0) return CWLT_OK;
    if ((colsum_k == nullptr) != (colsum_v == nullptr)) return CWLT_ERR_ARG;
    const bool fast = dtype == CWLT_BF16 && ((ldq | ldk | ldv | ldo | lddo | lddk | lddv) & 7) == 0;
    if (colsum_k && !fast) return CWLT_ERR_ARG;      // fused column sums exist in the bf16 kernels only
    if (dden && !fast) return CWLT_ERR_ARG;          // the dden hand-over too
    if (segments < 1 || (segments > 1 && (!fast || !seg_ws || !cwlt_segments_ok(L, segments)))) return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CWLT_F32)
        return launch_bwd_dkdv<float>(q, k, v, out, zinv, dout, dk, dv, N, H, L, ldq, ldk, ldv, ldo, lddo, lddk, lddv,
                                      st);
    if (dtype == CWLT_BF16) {
        if (fast)
            return launch_cla_bwd_dkdv_bf16(q, k, v, out, zinv, dout, dk, dv, colsum_k, colsum_v, dden, N, H, L, ldq,
                                            ldk, ldv, ldo, lddo, lddk, lddv, segments, seg_ws, st);
        return launch_bwd_dkdv<bf16_t>(q, k, v, out, zinv, dout, dk, dv, N, H, L, ldq, ldk, ldv, ldo, lddo, lddk,
                                       lddv, st);
    }
    return CWLT_ERR_DTYPE;
}

int cwlt_causal_linear_bwd_dq(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                              const void* dout, void* dq, float* colsum_q, const float* dden, int N, int H, int L,
                              int head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                              int64_t lddq, int segments, const float* seg_ws, int dtype, void* stream) {
    using namespace cwlt;
    if (N < 0 || H <= 0 || L < 0 || head_dim != D) return CWLT_ERR_ARG;
    if (N == 0 || L == 0) return CWLT_OK;
    if (!q || !k || !v || (!out && !dden) || !zinv || !dout || !dq) return CWLT_ERR_ARG;
    if (bad_ld(ldq, H) || bad_ld(ldk, H) || bad_ld(ldv, H) || bad_ld(ldo, H) || bad_ld(lddo, H) || bad_ld(lddq, H))
        return CWLT_ERR_ARG;
    if (N == 0 || L == 0) return CWLT_OK;
    const bool fast = dtype == CWLT_BF16 && ((ldq | ldk | ldv | ldo | lddo | lddq) & 7) == 0;
    if (colsum_q && !fast) return CWLT_ERR_ARG;      // fused column sums exist in the bf16 kernels only
    if (dden && !fast) return CWLT_ERR_ARG;
    if (segments < 1 || (segments > 1 && (!fast || !seg_ws || !cwlt_segments_ok(L, segments)))) return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CWLT_F32)
        return launch_bwd_dq<float>(q, k, v, out, zinv, dout, dq, N, H, L, ldq, ldk, ldv, ldo, lddo, lddq, st);
    if (dtype == CWLT_BF16) {
        if (fast)
            return launch_cla_bwd_dq_bf16(q, k, v, out, zinv, dout, dden, dq, colsum_q, N, H, L, ldq, ldk, ldv, ldo,
                                          lddo, lddq, segments, const_cast<float*>(seg_ws), st);
        return launch_bwd_dq<bf16_t>(q, k, v, out, zinv, dout, dq, N, H, L, ldq, ldk, ldv, ldo, lddo, lddq, st);
    }
    return CWLT_ERR_DTYPE;
}

/* the whole backward in one reverse sweep (bf16, whole sequences): every input stream is read once */
int cwlt_causal_linear_bwd_sweep(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                                 const void* dout, const float* final_state, void* dq, void* dk, void* dv,
                                 float* colsum_q, float* colsum_k, float* colsum_v, int N, int H, int L, int head_dim,
                                 int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq,
                                 int64_t lddk, int64_t lddv, int dtype, void* stream) {
    using namespace cwlt;
    if (N < 0 || H <= 0 || L < 0 || head_dim != D) return CWLT_ERR_ARG;
    if (N == 0 || L == 0) return CWLT_OK;
    if (!q || !k || !v || !out || !zinv || !dout || !final_state || !dq || !dk || !dv) return CWLT_ERR_ARG;
    if (bad_ld(ldq, H) || bad_ld(ldk, H) || bad_ld(ldv, H) || bad_ld(ldo, H) || bad_ld(lddo, H) ||
        bad_ld(lddq, H) || bad_ld(lddk, H) || bad_ld(lddv, H))
        return CWLT_ERR_ARG;
    if ((colsum_q == nullptr) != (colsum_k == nullptr) || (colsum_q == nullptr) != (colsum_v == nullptr))
        return CWLT_ERR_ARG;
    if (dtype != CWLT_BF16) return CWLT_ERR_DTYPE;
    if (((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) & 7) != 0) return CWLT_ERR_ARG;
    return launch_cla_bwd_sweep_bf16(q, k, v, out, zinv, dout, final_state, dq, dk, dv, colsum_q, colsum_k, colsum_v, N,
                                     H, L, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, (hipStream_t)stream);
}

/* both halves of the backward: the heavier reverse scan first, the dq scan back-fills its tail */
int cwlt_causal_linear_bwd(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                           const void* dout, void* dq, void* dk, void* dv, int N, int H, int L, int head_dim,
                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq,
                           int64_t lddk, int64_t lddv, int dtype, void* stream) {
    int e = cwlt_causal_linear_bwd_dkdv(q, k, v, out, zinv, dout, dk, dv, nullptr, nullptr, nullptr, N, H, L, head_dim,
                                        ldq, ldk, ldv, ldo, lddo, lddk, lddv, 1, nullptr, dtype, stream);
    if (e) return e;
    return cwlt_causal_linear_bwd_dq(q, k, v, out, zinv, dout, dq, nullptr, nullptr, N, H, L, head_dim, ldq, ldk, ldv,
                                     ldo, lddo, lddq, 1, nullptr, dtype, stream);
}

}  // extern "C"
