// One encoder layer per host call (bf16): cwlt_encoder_layer_fwd / _bwd enqueue the whole post-LN layer of
// fast_transformers' TransformerEncoderLayer + AttentionLayer + CausalLinearAttention (built at
// /root/reference/dqn_policy/model.py:128-137, called :232) on one stream -- 8 launches forward, 13-21 backward -- through
// the same entry points the per-op path uses, so both produce the same kernels' results.
//
// Why: at the reference's own RL setting (30 windows x 50 tokens, IRL_dqn_train.py:267-345, ppo_train.py:365-417) an
// update is ~970 launches of ~10 us of GPU work each and the HOST is the bound: 20 of its 22-25 ms are Python -- one
// ctypes call, two to six tensor allocations and an autograd node per kernel (HISTORY 9.4).  Here the host cost of a layer
// is one ctypes call and its launches; the caller allocates two buffers per layer (saved activations, the output) and
// reuses one scratch buffer.  Sizes and offsets: cwlt_encoder_layer_plan.  The plain projections go through
// cwlt_gemm_bf16_small (64 x 64 tiles): this path is for steps of a few thousand token rows; the per-op path with
// gemm_bf16.hip's 256 x 256 tiles keeps the training sizes.
#include <stdlib.h>

#include "cwlt_common.h"
#include "cwlt.h"

namespace {

constexpr int64_t ALIGN = 256;
inline int64_t up(int64_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

struct Dims {
    int64_t N, L, R;
    int D, F, H, P;
    bool drop, fin;
};

// byte offsets inside `saved`
struct Saved {
    int64_t qkv, a, zinv, fin, s1, x1, mean1, rstd1, gd, g, s2, mean2, rstd2, total;
};
// byte offsets inside the forward / backward scratch
struct FwdScratch {
    int64_t o, ws, total;
};
struct BwdScratch {
    int64_t ds2, dyl, dh, dox, dattn, dqkv, lnpart, wpart, mpart, cs, ws, dden, total;
    int64_t wp[4];             // the four weight gradients' partial-tile buffers (linear2, linear1, out, qkv), each inside wpart
};

bool dims_ok(const Dims& d) {
    return d.N > 0 && d.L > 0 && d.D == 512 && d.H == 8 && d.F > 0 && (d.F % 256) == 0 && d.R < (1ll << 31) / 4096;
}

Dims make_dims(int64_t n_seq, int64_t len, int d_model, int d_ff, int n_heads, float p, int want_backward) {
    Dims d;
    d.N = n_seq;
    d.L = len;
    d.R = n_seq * len;
    d.D = d_model;
    d.F = d_ff;
    d.H = n_heads;
    d.P = 1;
    d.drop = p > 0.f;
    d.fin = false;
    if (n_seq > 0 && len > 0 && n_seq < (1 << 30) && len < (1 << 30) && n_heads > 0) {
        d.P = cwlt_scan_segments((int)n_seq, n_heads, (int)len, CWLT_BF16);
        if (d.P < 1) d.P = 1;
        d.fin = want_backward && d.P == 1;
    }
    return d;
}

Saved plan_saved(const Dims& d) {
    Saved s;
    int64_t o = 0;
    const int64_t RD = d.R * d.D * 2, RF = d.R * d.F * 2, R4 = d.R * 4;
    s.qkv = o; o += up(3 * RD);
    s.a = o; o += up(RD);
    s.zinv = o; o += up(d.R * d.H * 4);
    s.fin = o; o += up(d.fin ? cwlt_scan_final_state_floats((int)d.N, d.H) * 4 : 0);
    s.s1 = o; o += up(RD);
    s.x1 = o; o += up(RD);
    s.mean1 = o; o += up(R4);
    s.rstd1 = o; o += up(R4);
    s.gd = o; o += up(RF);
    s.g = o; o += up(RF);
    s.s2 = o; o += up(RD);
    s.mean2 = o; o += up(R4);
    s.rstd2 = o; o += up(R4);
    s.total = o;
    return s;
}

FwdScratch plan_fwd(const Dims& d) {
    FwdScratch s;
    int64_t o = 0;
    s.o = o; o += up(d.R * d.D * 2);
    s.ws = o; o += up(d.P > 1 ? cwlt_scan_seg_floats((int)d.N, d.H, d.P, 0) * 4 : 0);
    s.total = o;
    return s;
}

int64_t max4(int64_t a, int64_t b, int64_t c, int64_t e) {
    int64_t m = a > b ? a : b;
    m = m > c ? m : c;
    return m > e ? m : e;
}

BwdScratch plan_bwd(const Dims& d) {
    BwdScratch s;
    int64_t o = 0;
    const int64_t RD = d.R * d.D * 2, RF = d.R * d.F * 2;
    const int D = d.D, F = d.F;
    s.ds2 = o; o += up(RD);
    s.dyl = o; o += up(d.drop ? RD : 0);
    s.dh = o; o += up(RF);
    s.dox = o; o += up(d.drop ? RD : 0);
    s.dattn = o; o += up(RD);
    s.dqkv = o; o += up(3 * RD);
    s.lnpart = o; o += up((int64_t)cwlt_ln_blocks(d.R) * 3 * D * 4);
    // the four weight gradients run as ONE grouped launch when their operands all survive to the end of the layer (dropout
    // on: the masked gradients are buffers of their own): four partial-tile buffers side by side; else one, reused
    const int64_t w4[4] = {(int64_t)cwlt_wgrad_splits(d.R, D, F) * D * F, (int64_t)cwlt_wgrad_splits(d.R, F, D) * F * D,
                           (int64_t)cwlt_wgrad_splits(d.R, D, D) * D * D, (int64_t)cwlt_wgrad_splits(d.R, 3 * D, D) * 3 * D * D};
    s.wpart = o;
    if (d.drop) {
        for (int i = 0; i < 4; ++i) {
            s.wp[i] = o;
            o += up(w4[i] * 4);
        }
    } else {
        for (int i = 0; i < 4; ++i) s.wp[i] = o;
        o += up(max4(w4[0], w4[1], w4[2], w4[3]) * 4);
    }
    s.mpart = o; o += up(cwlt_gemm_nt_tiles(d.R) * F * 4);
    s.cs = o; o += up(3 * d.N * d.P * D * 4);
    s.ws = o; o += up(!d.fin && d.P > 1 ? cwlt_scan_seg_floats((int)d.N, d.H, d.P, 1) * 4 : 0);
    s.dden = o; o += up(!d.fin ? d.R * d.H * 4 : 0);
    s.total = o;
    return s;
}

// float offsets of the parameter gradients inside `grads`: ln_bwd writes (dgamma | dbeta | dbias) as one (3, D) block,
// so [dgamma2, dbeta2, db2] and [dgamma1, dbeta1, dbo] are kept together
struct Grads {
    int64_t wqkv, bqkv, wo, w1, b1, w2, ln2, ln1, total;
};
Grads plan_grads(int D, int F) {
    Grads g;
    int64_t o = 0;
    const int64_t al = ALIGN / 4;
    auto take = [&](int64_t n) { const int64_t at = o; o += (n + al - 1) / al * al; return at; };
    g.wqkv = take(3ll * D * D);
    g.bqkv = take(3ll * D);
    g.wo = take((int64_t)D * D);
    g.w1 = take((int64_t)F * D);
    g.b1 = take(F);
    g.w2 = take((int64_t)D * F);
    g.ln2 = take(3ll * D);
    g.ln1 = take(3ll * D);
    g.total = o;
    return g;
}

// out[q HD + c] = sum_b cs[q][b][c], b ascending: the Q | K | V bias gradients from the scan's per-sequence column sums
__global__ void qkv_bias_sum_kernel(const float* __restrict__ cs, float* __restrict__ out, int nb, int HD) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * HD) return;
    const float* p = cs + (long)(i / HD) * nb * HD + (i % HD);
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += p[(long)b * HD];
    out[i] = s;
}

// CWLT_LAYER_GROUP_WGRAD=0: four separate weight-gradient launches per layer backward (A/B)
inline bool group_wgrads() {
    static const bool on = [] { const char* e = getenv("CWLT_LAYER_GROUP_WGRAD"); return !(e && e[0] == '0'); }();
    return on;
}

inline char* at(void* base, int64_t off) { return (char*)base + off; }
inline const char* at(const void* base, int64_t off) { return (const char*)base + off; }

#define CWLT_TRY(call)                 \
    {                                  \
        const int rc_ = (call);        \
        if (rc_ != CWLT_OK) return rc_; \
    }

}  // namespace

extern "C" {

int cwlt_encoder_layer_plan(int64_t n_seq, int64_t len, int d_model, int d_ff, int n_heads, float p_drop, int want_backward,
                            cwlt_encoder_layer_plan_t* plan) {
    if (!plan) return CWLT_ERR_ARG;
    const Dims d = make_dims(n_seq, len, d_model, d_ff, n_heads, p_drop, want_backward);
    if (!dims_ok(d)) return CWLT_ERR_ARG;
    const Saved s = plan_saved(d);
    const Grads g = plan_grads(d.D, d.F);
    plan->saved_bytes = s.total;
    plan->fwd_scratch_bytes = plan_fwd(d).total;
    plan->bwd_scratch_bytes = plan_bwd(d).total;
    plan->grad_floats = g.total;
    const int D = d.D;
    const int64_t off[CWLT_LAYER_NGRADS] = {g.wqkv, g.bqkv, g.wo, g.ln1 + 2 * D, g.w1, g.b1, g.w2, g.ln2 + 2 * D,
                                            g.ln1, g.ln1 + D, g.ln2, g.ln2 + D};
    for (int i = 0; i < CWLT_LAYER_NGRADS; ++i) plan->grad_off[i] = off[i];
    const int64_t sv[CWLT_LAYER_NSAVED] = {s.qkv, s.a, s.zinv, s.fin, s.s1, s.x1, s.mean1, s.rstd1, s.gd, s.g, s.s2,
                                           s.mean2, s.rstd2};
    for (int i = 0; i < CWLT_LAYER_NSAVED; ++i) plan->saved_off[i] = sv[i];
    return CWLT_OK;
}

int cwlt_encoder_layer_fwd(const cwlt_encoder_layer* a, void* stream) {
    if (!a) return CWLT_ERR_ARG;
    const Dims d = make_dims(a->n_seq, a->len, a->d_model, a->d_ff, a->n_heads, a->p_drop, a->want_backward);
    if (!dims_ok(d) || a->p_drop < 0.f || a->p_drop >= 1.f) return CWLT_ERR_ARG;
    if (!a->x || !a->y || !a->saved || !a->scratch || !a->wqkv || !a->wo || !a->w1 || !a->w2 || !a->bqkv || !a->bo ||
        !a->b1 || !a->b2 || !a->gamma1 || !a->beta1 || !a->gamma2 || !a->beta2)
        return CWLT_ERR_ARG;
    const Saved s = plan_saved(d);
    const FwdScratch f = plan_fwd(d);
    const int D = d.D, F = d.F, H = d.H;
    const int64_t R = d.R;
    char* qkv = at(a->saved, s.qkv);
    // fused Q | K | V projection
    CWLT_TRY(cwlt_gemm_bf16_small(a->x, a->wqkv, a->bqkv, qkv, R, 3 * D, D, D, D, 3 * D, 0, stream));
    // causal linear attention on the three column blocks, in place (row stride 3 D)
    CWLT_TRY(cwlt_causal_linear_fwd(qkv, qkv + 2 * D, qkv + 4 * D, at(a->saved, s.a), (float*)at(a->saved, s.zinv), (int)d.N, H,
                                    (int)d.L, D / H, 3 * D, 3 * D, 3 * D, D, a->attn_eps, d.P,
                                    d.P > 1 ? (float*)at(a->scratch, f.ws) : nullptr,
                                    d.fin ? (float*)at(a->saved, s.fin) : nullptr, CWLT_BF16, stream));
    // x1 = norm1(x + dropout(out_projection(attention)))
    CWLT_TRY(cwlt_gemm_bf16_small(at(a->saved, s.a), a->wo, a->bo, at(a->scratch, f.o), R, D, D, D, D, D, 0, stream));
    CWLT_TRY(cwlt_add_dropout_layernorm_fwd(a->x, at(a->scratch, f.o), a->gamma1, a->beta1, at(a->saved, s.s1),
                                            at(a->saved, s.x1), (float*)at(a->saved, s.mean1), (float*)at(a->saved, s.rstd1), R,
                                            D, a->ln_eps, a->p_drop, a->seed[0], a->seed_base, CWLT_BF16, stream));
    // g = dropout(gelu(linear1(x1))) and the backward's factor gd, one kernel
    // (below 256 rows on the split-K small tiles: a 128 x 256 tile leaves 8 workgroups with a 16-step chain each)
    if (R < 256) {
        CWLT_TRY(cwlt_gemm_bf16_small_gelu(at(a->saved, s.x1), a->w1, a->b1, at(a->saved, s.g),
                                           a->want_backward ? at(a->saved, s.gd) : nullptr, R, F, D, D, D, a->p_drop,
                                           a->seed[1], a->seed_base, stream));
    } else {
        CWLT_TRY(cwlt_gemm_nt_bias_gelu_dropout(at(a->saved, s.x1), a->w1, a->b1, at(a->saved, s.g), at(a->saved, s.gd), R, F,
                                                D, D, D, a->p_drop, a->seed[1], a->seed_base, stream));
    }
    // y = norm2(x1 + dropout(linear2(g)))
    CWLT_TRY(cwlt_gemm_bf16_small(at(a->saved, s.g), a->w2, a->b2, at(a->scratch, f.o), R, D, F, F, F, D, 0, stream));
    CWLT_TRY(cwlt_add_dropout_layernorm_fwd(at(a->saved, s.x1), at(a->scratch, f.o), a->gamma2, a->beta2, at(a->saved, s.s2),
                                            a->y, (float*)at(a->saved, s.mean2), (float*)at(a->saved, s.rstd2), R, D, a->ln_eps,
                                            a->p_drop, a->seed[2], a->seed_base, CWLT_BF16, stream));
    return CWLT_OK;
}

int cwlt_encoder_layer_bwd(const cwlt_encoder_layer* a, void* stream) {
    if (!a) return CWLT_ERR_ARG;
    const Dims d = make_dims(a->n_seq, a->len, a->d_model, a->d_ff, a->n_heads, a->p_drop, 1);
    if (!dims_ok(d) || !a->want_backward || a->p_drop < 0.f || a->p_drop >= 1.f) return CWLT_ERR_ARG;
    if (!a->x || !a->dy || !a->dx || !a->grads || !a->saved || !a->scratch || !a->wqkv_t || !a->wo_t || !a->w1_t ||
        !a->w2_t || !a->gamma1 || !a->gamma2)
        return CWLT_ERR_ARG;
    const Saved s = plan_saved(d);
    const BwdScratch b = plan_bwd(d);
    const Grads g = plan_grads(d.D, d.F);
    const int D = d.D, F = d.F, H = d.H;
    const int64_t R = d.R;
    hipStream_t st = (hipStream_t)stream;
    const bool grouped = d.drop && group_wgrads();
    float* lnpart = (float*)at(a->scratch, b.lnpart);
    char* ds2 = at(a->scratch, b.ds2);
    char* dyl = d.drop ? at(a->scratch, b.dyl) : ds2;          // gradient of linear2's output (= ds2 without dropout)
    char* dh = at(a->scratch, b.dh);
    char* dox = d.drop ? at(a->scratch, b.dox) : (char*)a->dx;  // gradient of the out-projection's output
    char* dattn = at(a->scratch, b.dattn);
    char* dqkv = at(a->scratch, b.dqkv);
    const char* qkv = at((const void*)a->saved, s.qkv);
    // norm2 backward: ds2 (residual gradient), dyl, (dgamma2 | dbeta2 | db2)
    CWLT_TRY(cwlt_add_dropout_layernorm_bwd(a->dy, nullptr, at(a->saved, s.s2), a->gamma2, (const float*)at(a->saved, s.mean2),
                                            (const float*)at(a->saved, s.rstd2), ds2, d.drop ? dyl : nullptr, lnpart,
                                            a->grads + g.ln2, R, D, a->p_drop, a->seed[2], a->seed_base, CWLT_BF16, stream));
    // linear2: weight gradient; input gradient x activation gradient (+ linear1's bias gradient) in one kernel
    if (!grouped)
        CWLT_TRY(cwlt_wgrad_bf16(dyl, at(a->saved, s.g), (float*)at(a->scratch, b.wp[0]), a->grads + g.w2, R, D, F, D, F, 0,
                                 stream));
    CWLT_TRY(cwlt_gemm_nt_mul(dyl, a->w2_t, at(a->saved, s.gd), dh, (float*)at(a->scratch, b.mpart), a->grads + g.b1, R, F, D, D,
                              D, F, F, stream));
    // linear1: input gradient lands ON the residual gradient; weight gradient
    CWLT_TRY(cwlt_gemm_bf16_small(dh, a->w1_t, nullptr, ds2, R, D, F, F, F, D, 1, stream));
    if (!grouped)
        CWLT_TRY(cwlt_wgrad_bf16(dh, at(a->saved, s.x1), (float*)at(a->scratch, b.wp[1]), a->grads + g.w1, R, F, D, F, D, 0,
                                 stream));
    // norm1 backward: dx <- residual gradient, dox, (dgamma1 | dbeta1 | dbo)
    CWLT_TRY(cwlt_add_dropout_layernorm_bwd(ds2, nullptr, at(a->saved, s.s1), a->gamma1, (const float*)at(a->saved, s.mean1),
                                            (const float*)at(a->saved, s.rstd1), a->dx, d.drop ? dox : nullptr, lnpart,
                                            a->grads + g.ln1, R, D, a->p_drop, a->seed[0], a->seed_base, CWLT_BF16, stream));
    // out-projection
    CWLT_TRY(cwlt_gemm_bf16_small(dox, a->wo_t, nullptr, dattn, R, D, D, D, D, D, 0, stream));
    if (!grouped)
        CWLT_TRY(cwlt_wgrad_bf16(dox, at(a->saved, s.a), (float*)at(a->scratch, b.wp[2]), a->grads + g.wo, R, D, D, D, D, 0,
                                 stream));
    // causal linear attention: dq | dk | dv side by side + their column sums per sequence (the Q/K/V bias gradients)
    float* cs = (float*)at(a->scratch, b.cs);
    const int64_t csn = d.N * d.P * D;
    if (d.fin) {
        CWLT_TRY(cwlt_causal_linear_bwd_sweep(qkv, qkv + 2 * D, qkv + 4 * D, at(a->saved, s.a), (const float*)at(a->saved, s.zinv),
                                              dattn, (const float*)at(a->saved, s.fin), dqkv, dqkv + 2 * D, dqkv + 4 * D, cs,
                                              cs + csn, cs + 2 * csn, (int)d.N, H, (int)d.L, D / H, 3 * D, 3 * D, 3 * D, D, D,
                                              3 * D, 3 * D, 3 * D, CWLT_BF16, stream));
    } else {
        float* ws = d.P > 1 ? (float*)at(a->scratch, b.ws) : nullptr;
        float* dden = (float*)at(a->scratch, b.dden);
        CWLT_TRY(cwlt_causal_linear_bwd_dkdv(qkv, qkv + 2 * D, qkv + 4 * D, at(a->saved, s.a), (const float*)at(a->saved, s.zinv),
                                             dattn, dqkv + 2 * D, dqkv + 4 * D, cs + csn, cs + 2 * csn, dden, (int)d.N, H,
                                             (int)d.L, D / H, 3 * D, 3 * D, 3 * D, D, D, 3 * D, 3 * D, d.P, ws, CWLT_BF16,
                                             stream));
        CWLT_TRY(cwlt_causal_linear_bwd_dq(qkv, qkv + 2 * D, qkv + 4 * D, at(a->saved, s.a), (const float*)at(a->saved, s.zinv),
                                           dattn, dqkv, cs, dden, (int)d.N, H, (int)d.L, D / H, 3 * D, 3 * D, 3 * D, D, D, 3 * D,
                                           d.P, ws, CWLT_BF16, stream));
    }
    hipLaunchKernelGGL(qkv_bias_sum_kernel, dim3((3 * D + 255) / 256), dim3(256), 0, st, cs, a->grads + g.bqkv,
                       (int)(d.N * d.P), D);
    CWLT_TRY((int)hipGetLastError());
    // Q | K | V projection: input gradient onto the residual gradient, weight gradient
    CWLT_TRY(cwlt_gemm_bf16_small(dqkv, a->wqkv_t, nullptr, a->dx, R, D, 3 * D, 3 * D, 3 * D, D, 1, stream));
    if (!grouped) {
        CWLT_TRY(cwlt_wgrad_bf16(dqkv, a->x, (float*)at(a->scratch, b.wp[3]), a->grads + g.wqkv, R, 3 * D, D, 3 * D, D, 0,
                                 stream));
        return CWLT_OK;
    }
    // the four weight gradients of the layer in one launch + one reduce (20-80 workgroups each at 1 500 rows, 240 together;
    // 8 launches of 12.6 + 5 us before).  Their operands are all still in place: with dropout on, the masked gradients dyl
    // and dox are buffers of their own, not the residual gradients the input-gradient products accumulate onto.
    const void* ga[4] = {dyl, dh, dox, dqkv};
    const void* gb[4] = {at(a->saved, s.g), at(a->saved, s.x1), at(a->saved, s.a), a->x};
    float* gp[4] = {(float*)at(a->scratch, b.wp[0]), (float*)at(a->scratch, b.wp[1]), (float*)at(a->scratch, b.wp[2]),
                    (float*)at(a->scratch, b.wp[3])};
    float* go[4] = {a->grads + g.w2, a->grads + g.w1, a->grads + g.wo, a->grads + g.wqkv};
    const int n1[4] = {D, F, D, 3 * D}, n2[4] = {F, D, D, D};
    const int64_t la[4] = {D, F, D, 3 * D}, lb[4] = {F, D, D, D};
    CWLT_TRY(cwlt_wgrad_bf16_group(ga, gb, gp, go, n1, n2, la, lb, 4, R, 0, stream));
    return CWLT_OK;
}

/* n consecutive layers (HOST array; layer i's y is layer i + 1's x, as the caller wired them): forward in order, backward
 * in reverse order (layer i's dx is layer i - 1's dy). */
int cwlt_encoder_fwd(const cwlt_encoder_layer* layers, int n, void* stream) {
    if (!layers || n < 0) return CWLT_ERR_ARG;
    for (int i = 0; i < n; ++i) CWLT_TRY(cwlt_encoder_layer_fwd(layers + i, stream));
    return CWLT_OK;
}
int cwlt_encoder_bwd(const cwlt_encoder_layer* layers, int n, void* stream) {
    if (!layers || n < 0) return CWLT_ERR_ARG;
    for (int i = n - 1; i >= 0; --i) CWLT_TRY(cwlt_encoder_layer_bwd(layers + i, stream));
    return CWLT_OK;
}

}  // extern "C"
