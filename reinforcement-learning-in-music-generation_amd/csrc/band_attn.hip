// Sliding-window (banded) softmax attention, forward -- the attention body of the AIRL discriminator.
//
// Replaces HF `LongformerSelfAttention.forward` (transformers, pure-PyTorch chunked matmuls:
// _sliding_chunks_query_key_matmul -> softmax(fp32) -> _sliding_chunks_matmul_attn_probs_value) as the
// reference reaches it through `LongformerModel(inputs_embeds=..., attention_mask=...)`:
// /root/reference/dqn_policy/AIRL_model.py:78-90,117 (window 50 -> 25 each side, L = 50) and
// ppo_policy/model.py:440-451,470 (window 512 -> 256 each side; the reference pads 50 tokens to 512).
//
//   score_ij = (q_i / sqrt(64)) . k_j   for |i - j| <= w, j < L, key j not masked;  -inf otherwise
//   p_i = softmax_j(score_ij) in f32;  masked QUERY rows give a zero output row (HF masked_fill)
//   out_i = sum_j dropout(p_ij) v_j
// No padding to a multiple of the window is needed: padded keys are masked out in the reference, so
// real tokens see exactly the keys inside the band; the 10x padded work of A16 simply is not done.
//
// q, k, v, out: (B, L, H, 64) row-strided like the scan kernels (column slices of a fused QKV buffer).
// One workgroup = 64 queries of one (b, h); key tiles of 64 inside the band; online softmax; f32 FMA
// from LDS (the problem is tiny: L = 50..1024, w = 25..256; HBM/latency-bound, not MFMA-shaped).
#include "cwlt_common.h"
#include "cwlt_mfma_bf16.h"

namespace cwlt {

constexpr int BD = 64;   // head dim
constexpr int BT = 64;   // query / key tile
constexpr int BLD = 65;  // LDS row stride

template <typename T>
__global__ __launch_bounds__(256) void band_attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ v, const float* __restrict__ mask,
                                                            T* __restrict__ out, float* __restrict__ lse, int H, int L,
                                                            int w, long ldq, long ldk, long ldv, long ldo, float scale,
                                                            uint32_t thresh, float keep_scale, uint64_t seed,
        const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += *seed_base;   // device-resident offset: lets a captured hipGraph draw fresh masks per replay
    __shared__ float qs[BT * BLD];
    __shared__ float ks[BT * BLD];
    __shared__ float vs[BT * BLD];
    __shared__ float ps[BT * BLD];
    __shared__ float kvalid[BT];

    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;        // 4x4 block: rows 4ti.., cols 4tj..
    const int qt = blockIdx.x;                     // query tile
    const int bh = blockIdx.y;
    const int b = bh / H, h = bh % H;
    const int q0 = qt * BT;
    const T* qb = q + ((long)b * L) * ldq + h * BD;
    const T* kb = k + ((long)b * L) * ldk + h * BD;
    const T* vb = v + ((long)b * L) * ldv + h * BD;
    T* ob = out + ((long)b * L) * ldo + h * BD;
    const float* mb = mask ? mask + (long)b * L : nullptr;

    // stage Q tile (scaled)
    for (int e = tid; e < BT * (BD / 4); e += 256) {
        const int r = e >> 4, c4 = (e & 15) * 4;
        float4 x = make_float4(0, 0, 0, 0);
        if (q0 + r < L) x = load4(qb + (long)(q0 + r) * ldq + c4);
        float* p = qs + r * BLD + c4;
        p[0] = x.x * scale; p[1] = x.y * scale; p[2] = x.z * scale; p[3] = x.w * scale;
    }
    float m_run[4], l_run[4], o[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        m_run[a] = -INFINITY;
        l_run[a] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[a][c] = 0.f;
    }
    int kt0 = (q0 - w) / BT;
    if (q0 - w < 0) kt0 = 0;
    int kt1 = (q0 + BT - 1 + w) / BT;
    const int ktmax = (L - 1) / BT;
    if (kt1 > ktmax) kt1 = ktmax;

    for (int kt = kt0; kt <= kt1; ++kt) {
        const int k0 = kt * BT;
        __syncthreads();
        for (int e = tid; e < BT * (BD / 4); e += 256) {
            const int r = e >> 4, c4 = (e & 15) * 4;
            float4 x = make_float4(0, 0, 0, 0), y = x;
            if (k0 + r < L) {
                x = load4(kb + (long)(k0 + r) * ldk + c4);
                y = load4(vb + (long)(k0 + r) * ldv + c4);
            }
            float* p = ks + r * BLD + c4;
            p[0] = x.x; p[1] = x.y; p[2] = x.z; p[3] = x.w;
            float* pv = vs + r * BLD + c4;
            pv[0] = y.x; pv[1] = y.y; pv[2] = y.z; pv[3] = y.w;
        }
        if (tid < BT) kvalid[tid] = (k0 + tid < L && (!mb || mb[k0 + tid] != 0.f)) ? 1.f : 0.f;
        __syncthreads();

        float s[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) s[a][c] = 0.f;
#pragma unroll 8
        for (int d = 0; d < BD; ++d) {
            float qa[4], kc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) qa[a] = qs[(4 * ti + a) * BLD + d];
#pragma unroll
            for (int c = 0; c < 4; ++c) kc[c] = ks[(4 * tj + c) * BLD + d];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) s[a][c] = fmaf(qa[a], kc[c], s[a][c]);
        }
        // band + key mask, online softmax over the 16 threads sharing a row
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = q0 + 4 * ti + a;
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = k0 + 4 * tj + c;
                const int dlt = i - j;
                const bool ok = dlt <= w && dlt >= -w && kvalid[4 * tj + c] != 0.f;
                s[a][c] = ok ? s[a][c] : -INFINITY;
                mx = fmaxf(mx, s[a][c]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
            const float mnew = fmaxf(m_run[a], mx);
            const float alpha = (m_run[a] == -INFINITY) ? 0.f : expf(m_run[a] - mnew);
            float rs = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float p = (s[a][c] == -INFINITY) ? 0.f : expf(s[a][c] - mnew);
                rs += p;
                float pd = p;
                if (thresh) {
                    // dropout on the attention probabilities: keyed by (b, h, i, j)
                    const uint64_t idx = (((uint64_t)bh * L + (uint64_t)i) * L) + (uint64_t)(k0 + 4 * tj + c);
                    pd = dropout_keep(seed, idx, thresh) ? p * keep_scale : 0.f;
                }
                ps[(4 * ti + a) * BLD + 4 * tj + c] = pd;
            }
            rs += __shfl_xor(rs, 1, 64);
            rs += __shfl_xor(rs, 2, 64);
            rs += __shfl_xor(rs, 4, 64);
            rs += __shfl_xor(rs, 8, 64);
            l_run[a] = l_run[a] * alpha + rs;
            m_run[a] = mnew;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[a][c] *= alpha;
        }
        __syncthreads();
        // O[i][d] += sum_j P[i][j] V[j][d]; this thread: rows 4ti.., dims 4tj..
#pragma unroll 8
        for (int j = 0; j < BT; ++j) {
            float pa[4], vc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) pa[a] = ps[(4 * ti + a) * BLD + j];
#pragma unroll
            for (int c = 0; c < 4; ++c) vc[c] = vs[j * BLD + 4 * tj + c];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) o[a][c] = fmaf(pa[a], vc[c], o[a][c]);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = q0 + 4 * ti + a;
        if (i >= L) continue;
        const bool qok = !mb || mb[i] != 0.f;
        const float inv = (qok && l_run[a] > 0.f) ? 1.0f / l_run[a] : 0.f;
        store4(ob + (long)i * ldo + 4 * tj, make_float4(o[a][0] * inv, o[a][1] * inv, o[a][2] * inv, o[a][3] * inv));
        // log-sum-exp of the row (for the backward); +inf marks rows whose output is forced to zero
        if (lse && tj == 0) lse[((long)b * H + h) * L + i] = (qok && l_run[a] > 0.f) ? m_run[a] + logf(l_run[a]) : INFINITY;
    }
}

// ------------------------------------------------------------------------------------------------
// forward, bf16 storage: the same algorithm on v_mfma_f32_32x32x16_bf16 (the f32-FMA kernel above is the
// parity path and stays the f32 implementation).  Scoring a replay buffer runs this on thousands of
// 50-token windows per environment step (dqn_policy/AIRL.py:69-91), where the FMA version was the largest
// single kernel of the loop.  4 waves (wi, wj); as in the scan kernels every product is taken in the
// orientation that leaves rows on registers and the QUERY index on lanes:
//   S^T tile (rows j, cols i) = K Q^T  -> the softmax statistics of query i are lane-local (16 registers +
//   one cross-half shuffle + a 2-wave exchange through LDS);  probabilities are rounded to bf16 once (as HF
//   does when the model runs in bf16) into ps[i][j];  O^T tile (rows m, cols i) += V^T P^T with V fetched
//   transposed by ds_read_b64_tr_b16, so rescaling by alpha_i is a per-lane scalar multiply.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_attn_fwd_bf16_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const float* __restrict__ mask, bf16_t* __restrict__ out, float* __restrict__ lse, int H, int L, int w, long ldq,
    long ldk, long ldv, long ldo, float scale, uint32_t thresh, float keep_scale, uint64_t seed,
    const uint64_t* __restrict__ seed_base) {
    using namespace b16;
    __shared__ __attribute__((aligned(16))) bf16_t qs[C * LD];   // q * scale [i][d]; reused as the output tile
    __shared__ __attribute__((aligned(16))) bf16_t ks[C * LD];   // k [j][d]
    __shared__ __attribute__((aligned(16))) bf16_t vs[C * LD];   // v [j][m]
    __shared__ __attribute__((aligned(16))) bf16_t ps[C * LD];   // dropout(p) [i][j]
    __shared__ float pmax[2][C], psum[2][C], kvalid[C];
    if (seed_base) seed += *seed_base;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wv >> 1, wj = wv & 1;
    const int l31 = lane & 31, hf = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int q0 = blockIdx.x * C;
    const bf16_t* qb = q + ((long)b * L) * ldq + h * D;
    const bf16_t* kb = k + ((long)b * L) * ldk + h * D;
    const bf16_t* vb = v + ((long)b * L) * ldv + h * D;
    bf16_t* ob = out + ((long)b * L) * ldo + h * D;
    const float* mb = mask ? mask + (long)b * L : nullptr;
    const int srow = tid >> 3, scol = (tid & 7) * 8;

#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = srow + 32 * it;
        float x[8];
        unpack8(q0 + row < L ? *reinterpret_cast<const uint4*>(qb + (long)(q0 + row) * ldq + scol) : CWLT_U4Z, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= scale;
        put_row(qs, row, scol, pack8(x));
    }
    // k / v rows of this (b, h) as buffer resources: rows >= L read back as zeros (hardware range check)
    const __amdgpu_buffer_rsrc_t kr = make_rsrc(kb, (uint32_t)(((long)(L - 1) * ldk + D) * 2));
    const __amdgpu_buffer_rsrc_t vr = make_rsrc(vb, (uint32_t)(((long)(L - 1) * ldv + D) * 2));
    const int il = 32 * wi + l31;          // this lane's query row inside the tile
    const int iq = q0 + il;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 O = zero16();                   // O^T: rows m = 32 wj + acc_row(r, hf), col i = il

    int kt0 = (q0 - w) / C;
    if (q0 - w < 0) kt0 = 0;
    int kt1 = (q0 + C - 1 + w) / C;
    const int ktmax = (L - 1) / C;
    if (kt1 > ktmax) kt1 = ktmax;

    for (int kt = kt0; kt <= kt1; ++kt) {
        const int k0 = kt * C;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = srow + 32 * it;
            put_row(ks, row, scol, buf_load16(kr, ((uint32_t)(k0 + row) * (uint32_t)ldk + scol) * 2));
            put_row(vs, row, scol, buf_load16(vr, ((uint32_t)(k0 + row) * (uint32_t)ldv + scol) * 2));
        }
        if (tid < C) kvalid[tid] = (k0 + tid < L && (!mb || mb[k0 + tid] != 0.f)) ? 1.f : 0.f;
        __syncthreads();

        f32x16 AT = prod_rows(zero16(), ks, 32 * wj + l31, qs, il, 0, 4, hf);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jl = 32 * wj + acc_row(r, hf);
            const int dlt = iq - (k0 + jl);
            const bool ok = dlt <= w && dlt >= -w && kvalid[jl] != 0.f;
            AT[r] = ok ? AT[r] : -INFINITY;
            mx = fmaxf(mx, AT[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (hf == 0) pmax[wj][il] = mx;
        __syncthreads();
        const float mnew = fmaxf(m_run, fmaxf(pmax[0][il], pmax[1][il]));
        const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - mnew);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = (AT[r] == -INFINITY) ? 0.f : __expf(AT[r] - mnew);
            rs += p;
            if (thresh) {
                const uint64_t idx = (((uint64_t)bh * L + (uint64_t)iq) * L) + (uint64_t)(k0 + 32 * wj + acc_row(r, hf));
                p = dropout_keep(seed, idx, thresh) ? p * keep_scale : 0.f;
            }
            AT[r] = p;
        }
        rs += __shfl_xor(rs, 32, 64);
        if (hf == 0) psum[wj][il] = rs;
        put_acc_T(ps, il, 32 * wj, AT, hf, 0, 64, 0.f);
        __syncthreads();
        l_run = l_run * alpha + (psum[0][il] + psum[1][il]);
        m_run = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) O[r] *= alpha;
#pragma unroll
        for (int s = 0; s < 4; ++s) O = mfma(tfrag8(vs, 16 * s, 32 * wj, lane), row8(ps, il, 16 * s + 8 * hf), O);
    }
    const bool qok = iq < L && (!mb || mb[iq] != 0.f);
    const float inv = (qok && l_run > 0.f) ? 1.0f / l_run : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) O[r] *= inv;
    __syncthreads();                                   // all waves are past their last read of qs
    put_acc_T(qs, il, 32 * wj, O, hf, 0, 64, 0.f);     // output tile [i][m], bf16
    if (lse && wj == 0 && hf == 0 && iq < L)
        lse[((long)b * H + h) * L + iq] = (qok && l_run > 0.f) ? m_run + logf(l_run) : INFINITY;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = srow + 32 * it;
        if (q0 + row < L)
            *reinterpret_cast<uint4*>(ob + (long)(q0 + row) * ldo + scol) = *reinterpret_cast<const uint4*>(qs + row * LD + scol);
    }
}

// ------------------------------------------------------------------------------------------------
// backward.  P_ij = exp(s_ij - lse_i) inside the band (0 for masked keys / masked queries), Pd = dropout(P),
//   O = Pd V;  dV_j = sum_i Pd_ij dO_i;  dP_ij = keep_ij * (dO_i . V_j);  delta_i = dO_i . O_i
//   dS_ij = P_ij (dP_ij - delta_i);  dQ_i = scale * sum_j dS_ij K_j;  dK_j = scale * sum_i dS_ij Q_i   (q unscaled)
// Two kernels, both recompute the 64x64 score tile from LDS-staged tiles with f32 FMAs:
//   dq kernel: one workgroup per 64 queries, loops key tiles; dkdv kernel: one workgroup per 64 keys, loops query tiles.
// ------------------------------------------------------------------------------------------------
template <typename T, bool FOR_KV>
__global__ __launch_bounds__(256) void band_attn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ v, const float* __restrict__ mask,
                                                            const T* __restrict__ out, const T* __restrict__ dout,
                                                            const float* __restrict__ lse, T* __restrict__ g1,
                                                            T* __restrict__ g2, int H, int L, int w, long ldq, long ldk,
                                                            long ldv, long ldo, long lddo, long ldg1, long ldg2,
                                                            float scale, uint32_t thresh, float keep_scale,
                                                            uint64_t seed,
        const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += *seed_base;   // device-resident offset: lets a captured hipGraph draw fresh masks per replay
    // "own" tile = the 64 rows this workgroup produces gradients for (queries for dq, keys for dk/dv);
    // "other" tile = the rows it loops over.
    __shared__ float a_s[BT * BLD];   // own tile, first operand  (dq: q*scale   | dkdv: k)
    __shared__ float b_s[BT * BLD];   // other tile, first operand (dq: k        | dkdv: q*scale)
    __shared__ float c_s[BT * BLD];   // dq: v (other)           | dkdv: v (own)
    __shared__ float d_s[BT * BLD];   // dq: dO (own)            | dkdv: dO (other)
    __shared__ float p_s[BT * BLD];   // dS (dq) / Pd then dS (dkdv), [own][other]
    __shared__ float r_lse[BT], r_del[BT], r_ok[BT];   // per QUERY row of the current query tile

    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int own0 = blockIdx.x * BT;
    const T* qb = q + ((long)b * L) * ldq + h * BD;
    const T* kb = k + ((long)b * L) * ldk + h * BD;
    const T* vb = v + ((long)b * L) * ldv + h * BD;
    const T* ob = out + ((long)b * L) * ldo + h * BD;
    const T* gb = dout + ((long)b * L) * lddo + h * BD;
    const float* mb = mask ? mask + (long)b * L : nullptr;
    const float* lb = lse + ((long)b * H + h) * L;

    auto stage = [&](float* dst, const T* src, long ld, int r0, float mul) {
        for (int e = tid; e < BT * (BD / 4); e += 256) {
            const int r = e >> 4, c4 = (e & 15) * 4;
            float4 x = make_float4(0, 0, 0, 0);
            if (r0 + r < L) x = load4(src + (long)(r0 + r) * ld + c4);
            float* p = dst + r * BLD + c4;
            p[0] = x.x * mul; p[1] = x.y * mul; p[2] = x.z * mul; p[3] = x.w * mul;
        }
    };
    // per-query-row statistics of query tile starting at q0: lse, delta = dO . O, validity
    auto row_stats = [&](int q0) {
        for (int r = tid >> 2; r < BT; r += 64) {
            const int i = q0 + r;
            float d = 0.f;
            if (i < L) {
                for (int c = (tid & 3) * 16; c < (tid & 3) * 16 + 16; c += 4) {
                    const float4 x = load4(gb + (long)i * lddo + c), y = load4(ob + (long)i * ldo + c);
                    d += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
                }
            }
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            if ((tid & 3) == 0) {
                const float ls = i < L ? lb[i] : INFINITY;
                r_lse[r] = ls;
                r_del[r] = d;
                r_ok[r] = (i < L && ls != INFINITY) ? 1.f : 0.f;
            }
        }
    };

    float acc1[4][4], acc2[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc1[a][c] = acc2[a][c] = 0.f;

    if (FOR_KV) {
        stage(a_s, kb, ldk, own0, 1.f);
        stage(c_s, vb, ldv, own0, 1.f);
    } else {
        stage(a_s, qb, ldq, own0, scale);
        stage(d_s, gb, lddo, own0, 1.f);
        row_stats(own0);
    }
    int t0 = (own0 - w) / BT;
    if (own0 - w < 0) t0 = 0;
    int t1 = (own0 + BT - 1 + w) / BT;
    const int tmax = (L - 1) / BT;
    if (t1 > tmax) t1 = tmax;

    for (int t = t0; t <= t1; ++t) {
        const int oth0 = t * BT;
        __syncthreads();
        if (FOR_KV) {
            stage(b_s, qb, ldq, oth0, scale);
            stage(d_s, gb, lddo, oth0, 1.f);
            row_stats(oth0);
        } else {
            stage(b_s, kb, ldk, oth0, 1.f);
            stage(c_s, vb, ldv, oth0, 1.f);
        }
        __syncthreads();
        // this thread's 4x4 block of [own][other]: s = a_own . b_other ; dp = (dO . V) taken query-major
        float sc[4][4], dp[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) sc[a][c] = dp[a][c] = 0.f;
#pragma unroll 8
        for (int d = 0; d < BD; ++d) {
            float xa[4], xb[4], ya[4], yb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                xa[a] = a_s[(4 * ti + a) * BLD + d];
                ya[a] = FOR_KV ? c_s[(4 * ti + a) * BLD + d] : d_s[(4 * ti + a) * BLD + d];   // own: v | dO
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                xb[c] = b_s[(4 * tj + c) * BLD + d];
                yb[c] = FOR_KV ? d_s[(4 * tj + c) * BLD + d] : c_s[(4 * tj + c) * BLD + d];   // other: dO | v
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    sc[a][c] = fmaf(xa[a], xb[c], sc[a][c]);
                    dp[a][c] = fmaf(ya[a], yb[c], dp[a][c]);
                }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ro = 4 * ti + a, rt = 4 * tj + c;            // row in own tile / other tile
                const int qi = FOR_KV ? oth0 + rt : own0 + ro;         // query index, key index
                const int kj = FOR_KV ? own0 + ro : oth0 + rt;
                const int qr = FOR_KV ? rt : ro;                       // query row inside its tile (stats)
                const int dlt = qi - kj;
                const bool ok = qi < L && kj < L && dlt <= w && dlt >= -w && r_ok[qr] != 0.f && (!mb || mb[kj] != 0.f);
                float p = ok ? expf(sc[a][c] - r_lse[qr]) : 0.f;
                float keep = 1.f;
                if (thresh && ok) {
                    const uint64_t idx = (((uint64_t)bh * L + (uint64_t)qi) * L) + (uint64_t)kj;
                    keep = dropout_keep(seed, idx, thresh) ? keep_scale : 0.f;
                }
                const float ds = p * (dp[a][c] * keep - r_del[qr]);
                if (FOR_KV) {
                    sc[a][c] = p * keep;    // Pd[own key][query]
                    dp[a][c] = ds;          // dS[own key][query]
                } else {
                    p_s[ro * BLD + rt] = ds;
                }
            }
        if (FOR_KV) {
            // dV[own] += Pd . dO(other) ; dK[own] += dS . (q*scale)(other) -- two passes through p_s
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                __syncthreads();
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) p_s[(4 * ti + a) * BLD + 4 * tj + c] = pass == 0 ? sc[a][c] : dp[a][c];
                __syncthreads();
                const float* src = pass == 0 ? d_s : b_s;
#pragma unroll 8
                for (int j = 0; j < BT; ++j) {
                    float pa[4], vc[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) pa[a] = p_s[(4 * ti + a) * BLD + j];
#pragma unroll
                    for (int c = 0; c < 4; ++c) vc[c] = src[j * BLD + 4 * tj + c];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (pass == 0) acc2[a][c] = fmaf(pa[a], vc[c], acc2[a][c]);
                            else acc1[a][c] = fmaf(pa[a], vc[c], acc1[a][c]);
                        }
                }
            }
        } else {
            __syncthreads();
            // dQ[own] += dS . K(other)
#pragma unroll 8
            for (int j = 0; j < BT; ++j) {
                float pa[4], vc[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) pa[a] = p_s[(4 * ti + a) * BLD + j];
#pragma unroll
                for (int c = 0; c < 4; ++c) vc[c] = b_s[j * BLD + 4 * tj + c];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc1[a][c] = fmaf(pa[a], vc[c], acc1[a][c]);
            }
        }
    }
    // dq kernel: g1 = dQ = scale * acc1 (a_s already carried one factor `scale` on q; K unscaled -> multiply once more? no:
    // s = (q*scale).k, dS is w.r.t. s, so dq = scale * dS K and dk = dS^T (q*scale) -- b_s holds q*scale already)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = own0 + 4 * ti + a;
        if (i >= L) continue;
        const float m = FOR_KV ? 1.f : scale;
        store4(g1 + ((long)b * L + i) * ldg1 + h * BD + 4 * tj,
               make_float4(acc1[a][0] * m, acc1[a][1] * m, acc1[a][2] * m, acc1[a][3] * m));
        if (FOR_KV)
            store4(g2 + ((long)b * L + i) * ldg2 + h * BD + 4 * tj,
                   make_float4(acc2[a][0], acc2[a][1], acc2[a][2], acc2[a][3]));
    }
}

}  // namespace cwlt

extern "C" {

/* mask: (B, L) f32, nonzero = attend (HF attention_mask), may be NULL.  window = ONE-SIDED width w
 * (HF attention_window / 2).  p = dropout on the attention probabilities (0 in eval). */
int cwlt_band_attn_fwd(const void* q, const void* k, const void* v, const float* mask, void* out, float* lse, int B,
                       int H, int L, int head_dim, int window, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                       float scale, float p, uint64_t seed, const uint64_t* seed_base, int dtype, void* stream) {
    using namespace cwlt;
    if (!q || !k || !v || !out || B < 0 || H <= 0 || L < 0 || head_dim != BD || window < 0) return CWLT_ERR_ARG;
    if ((ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3) || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (B == 0 || L == 0) return CWLT_OK;
    const dim3 grid((L + BT - 1) / BT, B * H), block(256);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t th = drop_thresh(p);
    const float ks = drop_scale(p);
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((band_attn_fwd_kernel<float>), grid, block, 0, st, (const float*)q, (const float*)k,
                           (const float*)v, mask, (float*)out, lse, H, L, window, (long)ldq, (long)ldk, (long)ldv,
                           (long)ldo, scale, th, ks, seed, seed_base);
    else if (dtype == CWLT_BF16 && !((ldq | ldk | ldv | ldo) & 7) &&
             !(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15))
        hipLaunchKernelGGL(band_attn_fwd_bf16_kernel, grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, mask, (bf16_t*)out, lse, H, L, window, (long)ldq, (long)ldk, (long)ldv,
                           (long)ldo, scale, th, ks, seed, seed_base);
    else if (dtype == CWLT_BF16)   // unaligned views: the generic kernel
        hipLaunchKernelGGL((band_attn_fwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, mask, (bf16_t*)out, lse, H, L, window, (long)ldq, (long)ldk, (long)ldv,
                           (long)ldo, scale, th, ks, seed, seed_base);
    else
        return CWLT_ERR_DTYPE;
    return (int)hipGetLastError();
}

/* Backward of cwlt_band_attn_fwd.  out / lse: the forward's outputs (lse (B, H, L) f32); dout: gradient w.r.t.
 * out (row stride lddo); dq, dk, dv: gradients w.r.t. the RAW q, k, v, laid out like them with row strides
 * lddq / lddk / lddv (e.g. the three column blocks of one fused (B*L, 3*H*64) buffer). */
int cwlt_band_attn_bwd(const void* q, const void* k, const void* v, const float* mask, const void* out,
                       const float* lse, const void* dout, void* dq, void* dk, void* dv, int B, int H, int L,
                       int head_dim, int window, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                       int64_t lddq, int64_t lddk, int64_t lddv, float scale, float p, uint64_t seed, const uint64_t* seed_base, int dtype,
                       void* stream) {
    using namespace cwlt;
    if (B < 0 || H <= 0 || L < 0 || head_dim != BD || window < 0 || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (B == 0 || L == 0) return CWLT_OK;
    if (!q || !k || !v || !out || !lse || !dout || !dq || !dk || !dv) return CWLT_ERR_ARG;
    if ((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) & 3) return CWLT_ERR_ARG;
    const dim3 grid((L + BT - 1) / BT, B * H), block(256);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t th = drop_thresh(p);
    const float ks = drop_scale(p);
#define CWLT_BAND_BWD(T)                                                                                          \
    hipLaunchKernelGGL((band_attn_bwd_kernel<T, true>), grid, block, 0, st, (const T*)q, (const T*)k, (const T*)v, \
                       mask, (const T*)out, (const T*)dout, lse, (T*)dk, (T*)dv, H, L, window, (long)ldq, (long)ldk, \
                       (long)ldv, (long)ldo, (long)lddo, (long)lddk, (long)lddv, scale, th, ks, seed, seed_base);            \
    hipLaunchKernelGGL((band_attn_bwd_kernel<T, false>), grid, block, 0, st, (const T*)q, (const T*)k, (const T*)v, \
                       mask, (const T*)out, (const T*)dout, lse, (T*)dq, (T*)nullptr, H, L, window, (long)ldq,     \
                       (long)ldk, (long)ldv, (long)ldo, (long)lddo, (long)lddq, (long)0, scale, th, ks, seed, seed_base)
    if (dtype == CWLT_F32) { CWLT_BAND_BWD(float); }
    else if (dtype == CWLT_BF16) { CWLT_BAND_BWD(bf16_t); }
    else return CWLT_ERR_DTYPE;
#undef CWLT_BAND_BWD
    return (int)hipGetLastError();
}

}  // extern "C"
