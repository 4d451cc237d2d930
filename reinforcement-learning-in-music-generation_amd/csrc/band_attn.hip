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

namespace cwlt {

constexpr int BD = 64;   // head dim
constexpr int BT = 64;   // query / key tile
constexpr int BLD = 65;  // LDS row stride

template <typename T>
__global__ __launch_bounds__(256) void band_attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ v, const float* __restrict__ mask,
                                                            T* __restrict__ out, int H, int L, int w, long ldq,
                                                            long ldk, long ldv, long ldo, float scale,
                                                            uint32_t thresh, float keep_scale, uint64_t seed) {
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
    }
}

}  // namespace cwlt

extern "C" {

/* mask: (B, L) f32, nonzero = attend (HF attention_mask), may be NULL.  window = ONE-SIDED width w
 * (HF attention_window / 2).  p = dropout on the attention probabilities (0 in eval). */
int cwlt_band_attn_fwd(const void* q, const void* k, const void* v, const float* mask, void* out, int B, int H,
                       int L, int head_dim, int window, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                       float scale, float p, uint64_t seed, int dtype, void* stream) {
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
                           (const float*)v, mask, (float*)out, H, L, window, (long)ldq, (long)ldk, (long)ldv,
                           (long)ldo, scale, th, ks, seed);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((band_attn_fwd_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, mask, (bf16_t*)out, H, L, window, (long)ldq, (long)ldk, (long)ldv,
                           (long)ldo, scale, th, ks, seed);
    else
        return CWLT_ERR_DTYPE;
    return (int)hipGetLastError();
}

}  // extern "C"
