// Generation-time decode step: one CW token through the recurrent form of the whole model, f32.
//
// Replaces the per-token body of the reference's generation loop
// (/root/reference/dqn_policy/testing-no-type-cp.py:150,166 -> dqn_policy/model.py:200-238 with
// is_training=False -> fast_transformers RecurrentEncoderBuilder product, then the six head projections of
// forward_output_sampling, dqn_policy/model.py:273-278; same for ppo_policy/inference.py / ppo_policy/model.py).
//
// At one token per step every matrix product is a GEMV: the step is bound by launch latency and by streaming the
// 156 MB of f32 weights (they fit the 256 MB Infinity Cache, so after the first token they do not come from HBM).
// The design removes launches, not flops:
//   * one GEMV kernel with the surrounding elementwise work folded in: LayerNorm(s) of the input as a PROLOGUE
//     (every wave normalises the 512-2048 element input vector in its own registers -- no LDS, no barrier, no
//     separate LayerNorm launch), bias + exact-erf GELU + residual add as the EPILOGUE;
//   * a wave owns R output rows and streams them with 16-byte loads issued before anything else (all of a wave's
//     weight traffic is in flight at once), then reduces with DPP;
//   * per layer: QKV GEMV -> recurrent attention step (recurrent.hip) -> out-projection GEMV (+residual) ->
//     FFN1 GEMV (LN1 prologue, GELU) -> FFN2 GEMV (+residual); LN2 is the next layer's prologue.
//   => 5 launches per layer, 63 per token for the 12-layer model, enqueued by ONE C call (cwlt_decode_step) that
//   the host captures into a hipGraph.
#include "cwlt.h"
#include "cwlt_common.h"

extern "C" int cwlt_recurrent_cla_step(const void* q, const void* k, const void* v, float* S, float* Z, void* out,
                                       int N, int H, int head_dim, int64_t ldq, int64_t ldk, int64_t ldv,
                                       int64_t ldo, float eps, int dtype, void* stream);
extern "C" int cwlt_cw_embed_fwd(const int64_t* tokens, const void* const* tables, const int* widths,
                                 const int* nrows, int n_attr, void* out, int64_t rows, int64_t ldo, int dtype,
                                 void* stream);

namespace cwlt {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

// LayerNorm of the K-vector held as NCH float4 per lane (invalid slots are zero and stay zero)
template <int NCH>
__device__ __forceinline__ void ln_in_wave(float4 (&x)[NCH], const float* __restrict__ w, const float* __restrict__ b,
                                           int K4, int lane, float inv_k, float eps) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += sum4(x[c]);
    const float mean = wave_sum(s) * inv_k;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c * 64 + lane < K4) {
            const float dx = x[c].x - mean, dy = x[c].y - mean, dz = x[c].z - mean, dw = x[c].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_k + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = c * 64 + lane;
        if (i < K4) {
            const float4 g = ((const float4*)w)[i], o = ((const float4*)b)[i];
            x[c].x = (x[c].x - mean) * rstd * g.x + o.x;
            x[c].y = (x[c].y - mean) * rstd * g.y + o.y;
            x[c].z = (x[c].z - mean) * rstd * g.z + o.z;
            x[c].w = (x[c].w - mean) * rstd * g.w + o.w;
        }
    }
}

// out[n, r] = epilogue( W[r, :] . prologue(xin[n, :]) + bias[r] )
//   prologue: optional LayerNorm (ln_w/ln_b), optionally followed by a second one (ln2_w/ln2_b); the normalised
//             vector is also written to x_out[n, :] (by the first wave) when x_out != NULL
//   epilogue: act == 1 -> exact-erf GELU; res != NULL -> + res[n*ld_res + r]
// blockIdx.y = n (song); a wave owns R consecutive rows; K % 4 == 0, K <= 256*NCH.
template <int NCH, int R>
__global__ __launch_bounds__(256) void decode_gemv_kernel(
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ xin, long ld_x,
    const float* __restrict__ ln_w, const float* __restrict__ ln_b, const float* __restrict__ ln2_w,
    const float* __restrict__ ln2_b, float eps, const float* __restrict__ res, long ld_res, float* __restrict__ out,
    long ld_out, float* __restrict__ x_out, long ld_xo, int Nout, int K, int act) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int row0 = wave * R;
    if (row0 >= Nout) return;                      // wave-uniform; the kernel has no barriers
    const int n = blockIdx.y;
    const int K4 = K >> 2;
    float4 w[R][NCH];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = min(row0 + r, Nout - 1);
        const float4* wr = (const float4*)(W + (long)row * K);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = c * 64 + lane;
            w[r][c] = i < K4 ? wr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 x[NCH];
    const float4* xr = (const float4*)(xin + (long)n * ld_x);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int i = c * 64 + lane;
        x[c] = i < K4 ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (ln_w) {
        const float inv_k = 1.0f / (float)K;
        ln_in_wave<NCH>(x, ln_w, ln_b, K4, lane, inv_k, eps);
        if (ln2_w) ln_in_wave<NCH>(x, ln2_w, ln2_b, K4, lane, inv_k, eps);
        if (x_out && wave == 0) {
            float4* xo = (float4*)(x_out + (long)n * ld_xo);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int i = c * 64 + lane;
                if (i < K4) xo[i] = x[c];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            acc = fmaf(w[r][c].x, x[c].x, acc);
            acc = fmaf(w[r][c].y, x[c].y, acc);
            acc = fmaf(w[r][c].z, x[c].z, acc);
            acc = fmaf(w[r][c].w, x[c].w, acc);
        }
        acc = wave_sum(acc);
        const int row = row0 + r;
        if (lane == 0 && row < Nout) {
            float y = acc + (bias ? bias[row] : 0.f);
            if (act == 1) y = gelu_erf(y);
            if (res) y += res[(long)n * ld_res + row];
            out[(long)n * ld_out + row] = y;
        }
    }
}

struct GemvArgs {
    const float *W, *bias, *xin;
    long ld_x;
    const float *ln_w, *ln_b, *ln2_w, *ln2_b;
    float eps;
    const float* res;
    long ld_res;
    float* out;
    long ld_out;
    float* x_out;
    long ld_xo;
    int Nout, K, act;
};

template <int NCH, int R>
static void launch_gemv(const GemvArgs& a, int n_songs, hipStream_t st) {
    const int waves = (a.Nout + R - 1) / R;
    hipLaunchKernelGGL((decode_gemv_kernel<NCH, R>), dim3((waves + 3) / 4, n_songs), dim3(256), 0, st, a.W, a.bias,
                       a.xin, a.ld_x, a.ln_w, a.ln_b, a.ln2_w, a.ln2_b, a.eps, a.res, a.ld_res, a.out, a.ld_out,
                       a.x_out, a.ld_xo, a.Nout, a.K, a.act);
}

static int gemv(const GemvArgs& a, int n_songs, hipStream_t st) {
    if (a.K <= 0 || (a.K & 3) || a.K > 2048 || a.Nout <= 0) return CWLT_ERR_ARG;
    const int nch = (a.K + 255) / 256;
    const bool two = a.Nout >= 1024;               // enough rows to fill 256 CUs with two rows per wave
    if (nch <= 1) two ? launch_gemv<1, 2>(a, n_songs, st) : launch_gemv<1, 1>(a, n_songs, st);
    else if (nch <= 2) two ? launch_gemv<2, 2>(a, n_songs, st) : launch_gemv<2, 1>(a, n_songs, st);
    else if (nch <= 5) two ? launch_gemv<5, 2>(a, n_songs, st) : launch_gemv<5, 1>(a, n_songs, st);
    else two ? launch_gemv<8, 2>(a, n_songs, st) : launch_gemv<8, 1>(a, n_songs, st);
    return (int)hipGetLastError();
}

static bool model_ok(const cwlt_decode_model* m) {
    return m && m->layers && m->tables && m->widths && m->nrows && m->w_in && m->w_heads && m->n_layer > 0 &&
           m->n_head > 0 && m->d_model == m->n_head * 64 && m->d_ff > 0 && (m->d_ff & 3) == 0 && m->d_ff <= 2048 &&
           m->d_model <= 2048 && m->emb_width > 0 && (m->emb_width & 3) == 0 && m->emb_width <= 2048 &&
           m->n_logits > 0 && m->n_attr > 0;
}

}  // namespace cwlt

extern "C" {

int64_t cwlt_decode_workspace_floats(const cwlt_decode_model* m) {
    if (!cwlt::model_ok(m)) return -1;
    // emb | x0 | xn | qkv | a | s1 | x1 | h | s2
    return (int64_t)m->emb_width + 9L * m->d_model + m->d_ff;
}

int cwlt_decode_gemv(const float* W, const float* bias, const float* xin, const float* ln_w, const float* ln_b,
                     const float* ln2_w, const float* ln2_b, float eps, const float* res, float* out, float* x_out,
                     int n_out, int K, int act, int n_songs, int64_t ld_x, int64_t ld_res, int64_t ld_out,
                     int64_t ld_xo, void* stream) {
    using namespace cwlt;
    if (!W || !xin || !out || n_songs <= 0 || (act != 0 && act != 1) || (ln_w && !ln_b) || (ln2_w && (!ln_w || !ln2_b)))
        return CWLT_ERR_ARG;
    GemvArgs a{W, bias, xin, (long)ld_x, ln_w, ln_b, ln2_w, ln2_b, eps, res, (long)ld_res, out, (long)ld_out, x_out,
               (long)ld_xo, n_out, K, act};
    return gemv(a, n_songs, (hipStream_t)stream);
}

int cwlt_decode_step(const cwlt_decode_model* m, const int64_t* tokens, float* work, float* hidden, float* logits,
                     int n_songs, void* stream) {
    using namespace cwlt;
    if (!model_ok(m) || !tokens || !work || !logits || n_songs <= 0) return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int D = m->d_model, F = m->d_ff, E = m->emb_width;
    const long ws = (long)cwlt_decode_workspace_floats(m);
    float* emb = work;
    float* x0 = emb + E;
    float* xn = x0 + D;
    float* qkv = xn + D;
    float* att = qkv + 3 * D;
    float* s1 = att + D;
    float* x1 = s1 + D;
    float* hh = x1 + D;
    float* s2 = hh + F;
    int rc = cwlt_cw_embed_fwd(tokens, m->tables, m->widths, m->nrows, m->n_attr, emb, n_songs, ws, CWLT_F32, stream);
    if (rc) return rc;
    // x0 = in_linear(emb) + pe[0]   (one token per call: the positional encoding never advances, model.py:90-92)
    {
        GemvArgs a{m->w_in, m->b_in, emb, ws, nullptr, nullptr, nullptr, nullptr, 0.f, m->pe0, 0, x0, ws, nullptr, 0,
                   D, E, 0};
        if ((rc = gemv(a, n_songs, st))) return rc;
    }
    for (int l = 0; l < m->n_layer; ++l) {
        const cwlt_decode_layer& L = m->layers[l];
        if (!L.wqkv || !L.wo || !L.w1 || !L.w2 || !L.ln1_w || !L.ln1_b || !L.ln2_w || !L.ln2_b || !L.S || !L.Z)
            return CWLT_ERR_ARG;
        const float* x = l == 0 ? x0 : xn;         // layer input (normalised by the previous layer's norm2)
        {
            const cwlt_decode_layer* P = l ? &m->layers[l - 1] : nullptr;
            GemvArgs a{L.wqkv, L.bqkv, l ? s2 : x0, ws, P ? P->ln2_w : nullptr, P ? P->ln2_b : nullptr, nullptr,
                       nullptr, m->eps_ln, nullptr, 0, qkv, ws, l ? xn : nullptr, ws, 3 * D, D, 0};
            if ((rc = gemv(a, n_songs, st))) return rc;
        }
        rc = cwlt_recurrent_cla_step(qkv, qkv + D, qkv + 2 * D, L.S, L.Z, att, n_songs, m->n_head, 64, ws, ws, ws, ws,
                                     m->eps_attn, CWLT_F32, stream);
        if (rc) return rc;
        {
            GemvArgs a{L.wo, L.bo, att, ws, nullptr, nullptr, nullptr, nullptr, 0.f, x, ws, s1, ws, nullptr, 0, D, D, 0};
            if ((rc = gemv(a, n_songs, st))) return rc;
        }
        {
            GemvArgs a{L.w1, L.b1, s1, ws, L.ln1_w, L.ln1_b, nullptr, nullptr, m->eps_ln, nullptr, 0, hh, ws, x1, ws,
                       F, D, 1};
            if ((rc = gemv(a, n_songs, st))) return rc;
        }
        {
            GemvArgs a{L.w2, L.b2, hh, ws, nullptr, nullptr, nullptr, nullptr, 0.f, x1, ws, s2, ws, nullptr, 0, D, F, 0};
            if ((rc = gemv(a, n_songs, st))) return rc;
        }
    }
    // heads on norm(norm2_last(s2)); the normalised hidden row is what forward_hidden returns
    {
        const cwlt_decode_layer& P = m->layers[m->n_layer - 1];
        GemvArgs a{m->w_heads, m->b_heads, s2, ws, P.ln2_w, P.ln2_b, m->lnf_w, m->lnf_w ? m->lnf_b : nullptr, m->eps_ln,
                   nullptr, 0, logits, (long)m->n_logits, hidden, (long)D, m->n_logits, D, 0};
        if ((rc = gemv(a, n_songs, st))) return rc;
    }
    return CWLT_OK;
}

}  // extern "C"
