"""Builder surface of the encoder the reference gets from fast_transformers, MI355X-native.

`TransformerEncoderBuilder.from_kwargs(...).get()` / `RecurrentEncoderBuilder` / `TriangularCausalMask`
keep the call shape of /root/reference/dqn_policy/model.py:9-11,128-150,231-238 so the model files
read like the reference's, and the modules keep the package's parameter names
(`layers.{i}.attention.{query,key,value,out}_projection`, `layers.{i}.{linear1,linear2,norm1,norm2}`,
`norm`), so reference checkpoints load unchanged.

Each encoder layer is ONE autograd node with an explicit forward/backward schedule:
  the dense projections (the only MFMA work) are libcwlt GEMMs in bf16 at training sizes -- cwlt_gemm_bf16 for the
  plain forward / input-gradient products (ops.gemm_bf16), the FFN and residual-block forms with their epilogues
  (ops.ffn1_gelu_dropout, ops.gemm_nt_mul, ops.linear_ln), cwlt_wgrad_bf16 for the weight gradients; f32 parity runs
  and steps of a few thousand token rows go to hipBLASLt through torch.mm/addmm;
  everything between them is a libcwlt HIP kernel -- fused QKV projection feeding the causal
  linear attention scan in place, residual+dropout+LayerNorm, bias+GELU+dropout -- and the backward
  folds bias / gamma / beta gradients into the same passes.
"""
import os

import torch
import torch.nn as nn

from . import _lib, ops


# CWLT_FUSED_FFN_BWD=0: keep the two-kernel FFN backward (hipBLASLt input-gradient GEMM + cwlt_bias_gelu_dropout_bwd)
FUSED_FFN_BWD = os.environ.get("CWLT_FUSED_FFN_BWD", "1") != "0"
# CWLT_FUSED_FFN_FWD=0: linear1 through hipBLASLt followed by cwlt_bias_gelu_dropout_fwd instead of the one-kernel form
FUSED_FFN_FWD = os.environ.get("CWLT_FUSED_FFN_FWD", "1") != "0"


class TriangularCausalMask:
    """fast_transformers.masking.TriangularCausalMask(N, device=...): a lower-triangular marker."""

    def __init__(self, N, device="cpu"):
        self.N = N
        self.device = device
        self.lower_triangular = True


DX1_IN_PLACE = os.environ.get("CWLT_DX1_IN_PLACE", "1") != "0"


class _EncoderLayerFn(torch.autograd.Function):
    """One post-LN encoder layer (fast_transformers TransformerEncoderLayer + AttentionLayer +
    CausalLinearAttention), x: (N, L, D) -> (N, L, D)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2, H, p, seeds, layer,
                shadow=None, grad_mode=True):
        # `grad_mode`: torch.is_grad_enabled() at the call site.  Inside forward() autograd has switched it off, and
        # ctx.needs_input_grad is true for the parameters even under torch.no_grad() -- without the flag every inference
        # and rollout pass would pay for what only a backward needs (the gd output, the scan's final state).
        will_backward = bool(grad_mode) and any(ctx.needs_input_grad)
        N, L, D = x.shape
        R = N * L
        adt = x.dtype
        x2 = x.reshape(R, D)
        bqkv32 = None
        if shadow is not None:
            # compute-dtype copies kept (and refreshed with one multi-tensor copy) by the encoder: ops.ShadowSet;
            # an 8th entry is the stacked Q/K/V bias in f32 (what cwlt_gemm_bf16 adds before its one rounding)
            wqkv, bqkv, wo_a, bo_a, w1_a, w2_a, b2_a = shadow[:7]
            bqkv32 = shadow[7] if len(shadow) > 7 else None
        else:
            wqkv = torch.cat([wq, wk, wv], 0).to(adt)
            bqkv = torch.cat([bq, bk, bv], 0).to(adt)
            wo_a, w1_a, w2_a = wo.to(adt), w1.to(adt), w2.to(adt)
            bo_a, b2_a = bo.to(adt), b2.to(adt)
        g1f, be1f, g2f, be2f, b1f = (ops._f32(t) for t in (g1, be1, g2, be2, b1))

        ctx.wt = shadow[10] if shadow is not None and len(shadow) > 10 else None
        if shadow is not None and len(shadow) > 8 and shadow[8] is not None:
            # steps of a few thousand token rows: the whole layer is ONE host call (csrc/layer.hip) -- the same kernels,
            # enqueued from C; what the backward needs stays in one `saved` buffer
            cache, li = shadow[8], shadow[9]
            F = w1_a.shape[0]
            plan = ops.layer_plan(N, L, D, F, H, p, will_backward)
            x2 = x2.contiguous()
            saved = torch.empty(plan.saved_bytes, dtype=torch.uint8, device=x.device)
            out = torch.empty((R, D), dtype=adt, device=x.device)
            st = _lib.EncoderLayer(n_seq=N, len=L, d_model=D, d_ff=F, n_heads=H, want_backward=1 if will_backward else 0,
                                   p_drop=p, ln_eps=ops.LN_EPS, attn_eps=ops.CLA_EPS)
            st.seed[0], st.seed[1], st.seed[2] = seeds
            sb = ops._seed_base()
            st.seed_base = sb.value if sb is not None else None
            st.wqkv, st.wo, st.w1, st.w2 = (t.data_ptr() for t in (wqkv, wo_a, w1_a, w2_a))
            st.bqkv, st.bo, st.b1, st.b2 = (t.data_ptr() for t in (bqkv32, ops._f32(bo), b1f, ops._f32(b2)))
            st.gamma1, st.beta1, st.gamma2, st.beta2 = (t.data_ptr() for t in (g1f, be1f, g2f, be2f))
            st.x, st.y, st.saved = x2.data_ptr(), out.data_ptr(), saved.data_ptr()
            st.scratch = cache.get_scratch(plan.fwd_scratch_bytes).data_ptr()
            ops.encoder_layer_fwd(st)
            if will_backward:
                ctx.c = (plan, saved, st, cache, li, x2, F)
                ctx.cfg = (N, L, D, H, p, seeds)
                ctx.layer = layer
            return out.view(N, L, D)
        ctx.c = None

        small = ops.gemm_small_per_op(x2)          # test switch: the per-op path on cwlt_gemm_bf16_small (what the
                                                    # one-call layer uses), so that the two can be compared bit for bit
        if small or ops.gemm_bf16_supported(x2, wqkv):
            if bqkv32 is None:
                bqkv32 = torch.cat([ops._f32(bq), ops._f32(bk), ops._f32(bv)])
            qkv = (ops.gemm_bf16_small if small else ops.gemm_bf16)(x2, wqkv, bqkv32)      # (R, 3D)  MFMA
        else:
            qkv = torch.addmm(bqkv, x2, wqkv.t())
        qkv5 = qkv.view(N, L, 3, H, D // H)
        # with a backward to follow, the bf16 scan also hands over its final state: the backward is then one sweep
        _, _, _, a, zinv, fin = ops.cla_fwd(qkv5[:, :, 0], qkv5[:, :, 1], qkv5[:, :, 2],
                                            final_state=will_backward)
        ctx.fin = fin
        a2 = a.view(R, D)
        if ops.linear_ln_supported(a2, wo_a, x2):
            # out-projection + bias + dropout + residual + LayerNorm in ONE kernel: the projection's output never
            # reaches HBM (ops.linear_ln; the bias enters in f32)
            s1, x1, mean1, rstd1 = ops.linear_ln(a2, wo_a, ops._f32(bo), x2, g1f, be1f, ops.LN_EPS, p, seeds[0])
        else:
            o = (ops.gemm_bf16_small(a2, wo_a, ops._f32(bo)) if small
                 else ops.gemm_bf16(a2, wo_a, ops._f32(bo)) if ops.gemm_bf16_supported(a2, wo_a)
                 else torch.addmm(bo_a, a2, wo_a.t()))                     # MFMA
            s1, x1, mean1, rstd1 = ops.ln_fwd(x2, o, g1f, be1f, ops.LN_EPS, p, seeds[0])
            del o
        # bf16, when a backward will follow: the activation pass also leaves gd = mask / (1 - p) * gelu'(h + b1) (in place
        # of h); the backward then needs no activation pass at all -- dh = (dy . W2) * gd leaves the input-gradient GEMM's
        # epilogue (ops.gemm_nt_mul).  The forward does the same from its side: linear1, bias, GELU and dropout are ONE
        # kernel (ops.ffn1_gelu_dropout), the pre-activation never reaches HBM.
        fused_ffn = (FUSED_FFN_BWD and adt == torch.bfloat16 and will_backward
                     and D % 64 == 0 and w1_a.shape[0] % 256 == 0)
        if fused_ffn and small and R < 256:
            g, h = ops.gemm_bf16_small_gelu(x1, w1_a, b1f, p, seeds[1])   # what the one-call layer runs below 256 rows
        elif fused_ffn and FUSED_FFN_FWD and ops.ffn1_fused_supported(x1, w1_a, b1f):
            g, h = ops.ffn1_gelu_dropout(x1, w1_a, b1f, p, seeds[1])      # h holds gd
        else:
            h = (ops.gemm_bf16_small(x1, w1_a) if small
                 else ops.gemm_bf16(x1, w1_a) if ops.gemm_bf16_supported(x1, w1_a)
                 else torch.mm(x1, w1_a.t()))                              # (R, F)  MFMA, bias in next kernel
            fused_ffn = fused_ffn and h.is_contiguous()
            g = ops.gelu_fwd(h, b1f, p, seeds[1], gd_inplace=fused_ffn)
        if ops.linear_ln_supported(g, w2_a, x1):
            s2, out, mean2, rstd2 = ops.linear_ln(g, w2_a, ops._f32(b2), x1, g2f, be2f, ops.LN_EPS, p, seeds[2])
        else:
            y = (ops.gemm_bf16_small(g, w2_a, ops._f32(b2)) if small
                 else ops.gemm_bf16(g, w2_a, ops._f32(b2)) if ops.gemm_bf16_supported(g, w2_a)
                 else torch.addmm(b2_a, g, w2_a.t()))                      # MFMA
            s2, out, mean2, rstd2 = ops.ln_fwd(x1, y, g2f, be2f, ops.LN_EPS, p, seeds[2])
            del y

        ctx.save_for_backward(x2, qkv, a, zinv, s1, mean1, rstd1, x1, h, g, s2, mean2, rstd2, g1f, g2f, b1f)
        # The weight copies may be the encoder's persistent shadow buffers, which a LATER forward refreshes in place
        # (same values unless an optimizer step came in between -- PPO runs two actor forwards before one backward),
        # so they are kept as plain attributes rather than version-checked saved tensors.
        ctx.weights = (wqkv, wo_a, w1_a, w2_a)
        ctx.cfg = (N, L, D, H, p, seeds)
        ctx.fused_ffn = fused_ffn
        ctx.layer = layer
        return out.view(N, L, D)

    @staticmethod
    def backward(ctx, dout):
        if ctx.c is not None:
            plan, saved, st, cache, li, x2, F = ctx.c
            N, L, D, H, p, seeds = ctx.cfg
            R = N * L
            dout2 = dout.reshape(R, D).contiguous()
            dx = torch.empty((R, D), dtype=dout2.dtype, device=dout2.device)
            grads = torch.empty(plan.grad_floats, dtype=torch.float32, device=dout2.device)
            st.wqkv_t, st.wo_t, st.w1_t, st.w2_t = (t.data_ptr() for t in cache.views[4 * li:4 * li + 4])
            st.dy, st.dx, st.grads = dout2.data_ptr(), dx.data_ptr(), grads.data_ptr()
            st.scratch = cache.get_scratch(plan.bwd_scratch_bytes).data_ptr()
            ops.encoder_layer_bwd(st)
            ctx.c = None
            go = plan.grad_off

            def gv(i, *shape):
                n = 1
                for d_ in shape:
                    n *= d_
                return grads[go[i]:go[i] + n].view(shape)

            return _EncoderLayerFn._hand_over(ctx.layer, dx.view(N, L, D), D, gv(0, 3 * D, D), gv(1, 3 * D), gv(2, D, D),
                                              gv(3, D), gv(4, F, D), gv(5, F), gv(6, D, F), gv(7, D), gv(8, D), gv(9, D),
                                              gv(10, D), gv(11, D))
        x2, qkv, a, zinv, s1, mean1, rstd1, x1, h, g, s2, mean2, rstd2, g1f, g2f, b1f = ctx.saved_tensors
        wqkv, wo_a, w1_a, w2_a = ctx.weights
        N, L, D, H, p, seeds = ctx.cfg
        R = N * L
        dout2 = dout.reshape(R, D)

        # weight gradients: hand-written split-K MFMA GEMM (f32 result) in bf16 mode, hipBLASLt otherwise
        def wgrad(a_, b_):
            if ops.wgrad_supported(a_, b_):
                return ops.wgrad(a_, b_)
            return torch.mm(a_.t(), b_)

        ds2, dy, dg2, dbe2, db2 = ops.ln_bwd(dout2, None, s2, g2f, mean2, rstd2, p, seeds[2])
        # Input-gradient GEMMs run as NT products on explicitly transposed weights (2 MB each): hipBLASLt's TN kernels
        # for these shapes are 20-30 % faster than its NN kernels (tuning table: 0.94 vs 1.20 ms for FFN2 at B = 512)
        # ... at large R; below ~16k rows the three 2 MB transposes cost more than the GEMMs gain
        nt = os.environ.get("CWLT_DGRAD_NT", "1") != "0" and dy.dtype == torch.bfloat16 and R >= 16384

        # bf16 at training sizes: cwlt_gemm_bf16 on the transposed weight (both operands K-contiguous; a 0.5-2 MB copy),
        # `out` = the residual gradient the product is added onto
        small = ops.gemm_small_per_op(dout2)

        # the transposed copies: refreshed for all layers by ONE launch per forward when the encoder passed them
        # (ops.LayerCache.refresh_transposed), else made here (a 0.5-2 MB copy kernel per use)
        wt = ctx.wt

        def tr(w_, j):
            return wt[j] if wt is not None else w_.t().contiguous()

        def dgrad(g_, w_, j, out=None):
            if small:
                return ops.gemm_bf16_small(g_, tr(w_, j), out=out, accumulate=out is not None)
            if ops.gemm_bf16_supported(g_, w_.t(), out, transposed_w=True):
                return ops.gemm_bf16(g_, tr(w_, j), out=out, accumulate=out is not None)
            if out is not None:
                return out.addmm_(g_, tr(w_, j).t() if nt else w_)
            return torch.mm(g_, tr(w_, j).t()) if nt else torch.mm(g_, w_)

        dw2 = wgrad(dy, g)                                                 # (D, F)
        if ctx.fused_ffn:
            # h holds gd (see forward): input gradient of linear2, activation + dropout backward and linear1's bias
            # gradient in ONE kernel; dgact (R x F) never reaches memory
            dh, db1 = ops.gemm_nt_mul(dy, tr(w2_a, 3), h)
        else:
            dgact = dgrad(dy, w2_a, 3)                                        # (R, F)
            dh, db1 = ops.gelu_bwd(dgact, h, b1f, p, seeds[1])
            del dgact
        if DX1_IN_PLACE:
            # linear1's input gradient lands ON the residual gradient (C = D = ds2, beta = 1): norm1's backward then
            # reads one gradient stream instead of two, and the extra read sits in a GEMM that has HBM time to spare.
            # dy may alias ds2 (p = 0); its last readers (dw2, dh) are already enqueued.
            dgrad(dh, w1_a, 2, out=ds2)                                       # (R, D)
            dx1 = None
        else:
            dx1 = dgrad(dh, w1_a, 2)                                          # (R, D)
        dw1 = wgrad(dh, x1)                                                # (F, D)
        del dh
        ds1, do, dg1, dbe1, dbo = ops.ln_bwd(ds2, dx1, s1, g1f, mean1, rstd1, p, seeds[0])
        da = dgrad(do, wo_a, 1)                                               # (R, D)
        dwo = wgrad(do, a.view(R, D))
        qkv5 = qkv.view(N, L, 3, H, D // H)
        dqkv, dbqkv = ops.cla_bwd(qkv5[:, :, 0], qkv5[:, :, 1], qkv5[:, :, 2], a, zinv, da.view(N, L, H, D // H),
                                  want_colsum=True, final_state=ctx.fin)
        ctx.fin = None
        dqkv2 = dqkv.view(R, 3 * D)
        if small:
            dx = ops.gemm_bf16_small(dqkv2, tr(wqkv, 0), out=ds1, accumulate=True)
        elif ops.gemm_bf16_supported(dqkv2, wqkv.t(), ds1, transposed_w=True):
            dx = ops.gemm_bf16(dqkv2, tr(wqkv, 0), out=ds1, accumulate=True)
        else:
            dx = ds1.addmm_(dqkv2, wqkv)                                   # residual + projection gradient, in place
                                                                           # (NN is hipBLASLt's faster form for this shape)
                                                                           # (out-of-place addmm first copies ds1: 268 MB)
        dwqkv = wgrad(dqkv2, x2)                                           # (3D, D)
        return _EncoderLayerFn._hand_over(ctx.layer, dx.view(N, L, D), D, dwqkv, dbqkv, dwo, dbo, dw1, db1, dw2, db2, dg1,
                                          dbe1, dg2, dbe2)

    @staticmethod
    def _hand_over(layer, dx, D, dwqkv, dbqkv, dwo, dbo, dw1, db1, dw2, db2, dg1, dbe1, dg2, dbe2):
        if layer is not None and ops.direct_grads(layer.linear1.weight):
            # write the 16 parameter gradients straight into .grad (flat f32 buckets) with one multi-tensor
            # convert+store, instead of .float() + autograd's accumulate; under data parallelism the buckets
            # (dist.GradSync) are notified per parameter and launch their all-reduce as they complete
            at = layer.attention
            ops.deliver_grads(((at.query_projection.weight, dwqkv[:D]), (at.query_projection.bias, dbqkv[:D]),
                               (at.key_projection.weight, dwqkv[D:2 * D]), (at.key_projection.bias, dbqkv[D:2 * D]),
                               (at.value_projection.weight, dwqkv[2 * D:]), (at.value_projection.bias, dbqkv[2 * D:]),
                               (at.out_projection.weight, dwo), (at.out_projection.bias, dbo),
                               (layer.linear1.weight, dw1), (layer.linear1.bias, db1),
                               (layer.linear2.weight, dw2), (layer.linear2.bias, db2),
                               (layer.norm1.weight, dg1), (layer.norm1.bias, dbe1),
                               (layer.norm2.weight, dg2), (layer.norm2.bias, dbe2)))
            return (dx,) + (None,) * 22
        dwqkv = dwqkv.float()
        return (dx,
                dwqkv[:D], dbqkv[:D], dwqkv[D:2 * D], dbqkv[D:2 * D], dwqkv[2 * D:], dbqkv[2 * D:],
                dwo.float(), dbo, dw1.float(), db1, dw2.float(), db2, dg1, dbe1, dg2, dbe2,
                None, None, None, None, None, None)


def _layer_params(layer):
    """The 16 parameters of one layer in the order _EncoderLayerFn takes them."""
    at = layer.attention
    return (at.query_projection.weight, at.query_projection.bias, at.key_projection.weight, at.key_projection.bias,
            at.value_projection.weight, at.value_projection.bias, at.out_projection.weight, at.out_projection.bias,
            layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias,
            layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias)


class _EncoderStackFn(torch.autograd.Function):
    """All layers of a TransformerEncoder as ONE autograd node and one host call each way (csrc/layer.hip:
    cwlt_encoder_fwd / _bwd): the same kernels as _EncoderLayerFn's one-call branch, without a Python-level node,
    allocations and seed draws per layer.  Used for steps of a few thousand token rows (TransformerEncoder._layer_c_ok).
    What does not change between calls (weight / transposed-weight / stacked-bias pointers, layer dims, the flat
    parameter list, the gradient views) lives in the encoder's ops.LayerCache."""

    @staticmethod
    def forward(ctx, x, enc, cache, grad_mode, *params):
        will_backward = bool(grad_mode) and any(ctx.needs_input_grad)
        arr = cache.arr
        n = len(arr)
        N, L, D = x.shape
        R = N * L
        _, F, H = cache.dims
        p = enc.layers[0].dropout.p if enc.training else 0.0
        plan = ops.layer_plan(N, L, D, F, H, p, will_backward)
        dev = x.device
        x2 = x.reshape(R, D).contiguous()
        acts = torch.empty((n, R, D), dtype=x.dtype, device=dev)
        saved = torch.empty((n, plan.saved_bytes), dtype=torch.uint8, device=dev)
        seeds = ops.next_seeds(3 * n) if p > 0 else [0] * (3 * n)
        sb = ops._seed_base()
        sb = sb.value if sb is not None else None
        scratch = cache.get_scratch(plan.fwd_scratch_bytes).data_ptr()
        xp, ap, sp, sstride, astride = x2.data_ptr(), acts.data_ptr(), saved.data_ptr(), plan.saved_bytes, R * D * 2
        wb = 1 if will_backward else 0
        vec = [t.data_ptr() for t in params]                      # f32 master parameters: bias / gamma / beta pointers
        for i in range(n):
            st = arr[i]
            o = 16 * i
            st.n_seq, st.len, st.want_backward, st.p_drop, st.seed_base = N, L, wb, p, sb
            st.seed[0], st.seed[1], st.seed[2] = seeds[3 * i:3 * i + 3]
            st.bo, st.b1, st.b2 = vec[o + 7], vec[o + 9], vec[o + 11]
            st.gamma1, st.beta1, st.gamma2, st.beta2 = vec[o + 12:o + 16]
            st.x = xp if i == 0 else ap + (i - 1) * astride
            st.y = ap + i * astride
            st.saved = sp + i * sstride
            st.scratch = scratch
        ops.encoder_fwd(arr, n)
        if will_backward:
            ctx.c = (plan, saved, acts, x2, cache, seeds, sb, (N, L, D, p), vec)
        return acts[n - 1].view(N, L, D)

    @staticmethod
    def backward(ctx, dout):
        plan, saved, acts, x2, cache, seeds, sb, (N, L, D, p), vec = ctx.c
        ctx.c = None
        arr = cache.arr
        n = len(arr)
        R = N * L
        dout2 = dout.reshape(R, D).contiguous()
        dxs = torch.empty((n, R, D), dtype=dout2.dtype, device=dout2.device)
        grads, views = cache.grad_buffer(plan)
        gf = plan.grad_floats
        scratch = cache.get_scratch(plan.bwd_scratch_bytes).data_ptr()
        xp, ap, sp, sstride, astride = x2.data_ptr(), acts.data_ptr(), saved.data_ptr(), plan.saved_bytes, R * D * 2
        dp, gp, dyp = dxs.data_ptr(), grads.data_ptr(), dout2.data_ptr()
        for i in range(n):
            # every per-call field again: another forward of this encoder may have run since (PPO: two actor passes before
            # one backward)
            st = arr[i]
            o = 16 * i
            st.n_seq, st.len, st.want_backward, st.p_drop, st.seed_base = N, L, 1, p, sb
            st.seed[0], st.seed[1], st.seed[2] = seeds[3 * i:3 * i + 3]
            st.gamma1, st.gamma2 = vec[o + 12], vec[o + 14]
            st.x = xp if i == 0 else ap + (i - 1) * astride
            st.saved = sp + i * sstride
            st.dy = dyp if i == n - 1 else dp + (i + 1) * astride
            st.dx = dp + i * astride
            st.grads = gp + i * gf * 4
            st.scratch = scratch
        ops.encoder_bwd(arr, n)
        dx = dxs[0].view(N, L, D)
        if ops.direct_grads(cache.params[8]):
            ops.deliver_grads_flat(cache.params, views, cache.all_need_grad)
            return (dx, None, None, None) + (None,) * (16 * n)
        return (dx, None, None, None) + tuple(v.clone() for v in views)


class AttentionLayer(nn.Module):
    """Holds the four projections under fast_transformers' names; the arithmetic is in _EncoderLayerFn."""

    def __init__(self, d_model, n_heads, d_keys, d_values):
        super().__init__()
        self.query_projection = nn.Linear(d_model, d_keys * n_heads)
        self.key_projection = nn.Linear(d_model, d_keys * n_heads)
        self.value_projection = nn.Linear(d_model, d_values * n_heads)
        self.out_projection = nn.Linear(d_values * n_heads, d_model)
        self.n_heads = n_heads


class TransformerEncoderLayer(nn.Module):
    def __init__(self, attention, d_model, d_ff, dropout):
        super().__init__()
        self.attention = attention
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)  # only carries p / train-eval state

    def shadow_groups(self):
        """Parameter groups of one layer for ops.ShadowSet, in the order _EncoderLayerFn unpacks them."""
        at = self.attention
        return [(at.query_projection.weight, at.key_projection.weight, at.value_projection.weight),
                (at.query_projection.bias, at.key_projection.bias, at.value_projection.bias),
                (at.out_projection.weight,), (at.out_projection.bias,), (self.linear1.weight,), (self.linear2.weight,),
                (self.linear2.bias,)]

    def forward(self, x, attn_mask=None, shadow=None):
        if attn_mask is not None and not getattr(attn_mask, "lower_triangular", False):
            raise RuntimeError("CausalLinearAttention only supports full lower triangular masks")
        p = self.dropout.p if self.training else 0.0
        seeds = tuple(ops.next_seed() for _ in range(3)) if p > 0 else (0, 0, 0)
        at = self.attention
        return _EncoderLayerFn.apply(
            x, at.query_projection.weight, at.query_projection.bias, at.key_projection.weight, at.key_projection.bias,
            at.value_projection.weight, at.value_projection.bias, at.out_projection.weight, at.out_projection.bias,
            self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
            self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias, at.n_heads, p, seeds, self, shadow,
            torch.is_grad_enabled())


class TransformerEncoder(nn.Module):
    def __init__(self, layers, norm_layer=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.norm = norm_layer
        self._shadow = None                                  # ops.ShadowSet of the layers' compute-dtype weights
        self._shadow32 = None                                # ... and of the stacked Q/K/V biases in f32 (bf16 mode)
        self._cache = None                                   # ops.LayerCache of the one-call layers (few token rows)
        self._tcache = None                                  # ops.LayerCache holding the transposed weights (training sizes)

    def _layer_c_ok(self, x):
        """Whether this forward runs its layers as one host call each (csrc/layer.hip): bf16, at most
        ops.LAYER_C_MAX_ROWS token rows, the repo's layer shape, f32 master parameters."""
        if not (ops.LAYER_C and x.dtype == torch.bfloat16 and 0 < x.shape[0] * x.shape[1] <= ops.LAYER_C_MAX_ROWS
                and x.shape[2] == 512 and len(self.layers) > 0):
            return False
        ok = getattr(self, "_layer_c_params", None)
        if ok is None:
            l0 = self.layers[0]
            ok = all(layer.attention.n_heads == 8 and layer.linear1.weight.shape[0] % 256 == 0
                     and layer.linear1.weight.shape == l0.linear1.weight.shape and layer.dropout.p == l0.dropout.p
                     and layer.linear1.weight.shape[1] == 512
                     and all(q.dtype == torch.float32 and q.is_contiguous() for q in layer.parameters())
                     for layer in self.layers)
            self._layer_c_params = ok
        return ok

    def forward(self, x, attn_mask=None, length_mask=None):
        if length_mask is not None:
            raise NotImplementedError("the reference never passes a length mask (dqn_policy/model.py:232)")
        if not x.is_cuda:
            raise RuntimeError("rlmg_amd encoder runs on the GPU only (no CPU fallback)")
        per = 7                                             # buffers per layer (TransformerEncoderLayer.shadow_groups)
        sh = self._shadow
        if sh is None or not sh.matches(x.dtype, x.device):
            sh = self._shadow = ops.ShadowSet([g for layer in self.layers for g in layer.shadow_groups()], x.dtype)
        bufs = sh.refresh()
        bufs32 = None
        cache = None
        if self._layer_c_ok(x):
            # few token rows: one host call per layer (csrc/layer.hip); with a backward to come, the transposed weight
            # copies of all layers are refreshed by one launch
            cache = self._cache
            if cache is None or cache.owner is not sh:
                cache = self._cache = ops.LayerCache([bufs[per * i + j] for i in range(len(self.layers))
                                                      for j in (0, 2, 4, 5)], sh)
            if torch.is_grad_enabled():
                cache.refresh_transposed()
        if x.dtype == torch.bfloat16 and (cache is not None or (x.shape[0] * x.shape[1] >= ops.GEMM_BF16_MIN_ROWS
                                                                and ops.GEMM_BF16)):
            s32 = self._shadow32
            if s32 is None or not s32.matches(torch.float32, x.device):
                s32 = self._shadow32 = ops.ShadowSet([layer.shadow_groups()[1] for layer in self.layers], torch.float32)
            bufs32 = s32.refresh()
        if cache is not None and ops.LAYER_C_STACK:
            if attn_mask is not None and not getattr(attn_mask, "lower_triangular", False):
                raise RuntimeError("CausalLinearAttention only supports full lower triangular masks")
            if getattr(cache, "qkv_bias_owner", None) is not bufs32:
                cache.setup_stack([q for layer in self.layers for q in _layer_params(layer)],
                                  [bufs[per * i + j] for i in range(len(self.layers)) for j in (0, 2, 4, 5)], bufs32,
                                  (x.shape[2], self.layers[0].linear1.weight.shape[0], self.layers[0].attention.n_heads))
            x = _EncoderStackFn.apply(x, self, cache, torch.is_grad_enabled(), *cache.params)
        else:
            tcache = None
            if (cache is None and ops.DGRAD_WT_CACHE and x.dtype == torch.bfloat16 and torch.is_grad_enabled()
                    and x.shape[0] * x.shape[1] >= 16384):
                # training sizes: the input-gradient products run on transposed weight copies (encoder._EncoderLayerFn
                # backward); all 48 of them by one launch here instead of a copy kernel per use there
                tcache = self._tcache
                if tcache is None or tcache.owner is not sh:
                    tcache = self._tcache = ops.LayerCache([bufs[per * i + j] for i in range(len(self.layers))
                                                            for j in (0, 2, 4, 5)], sh)
                tcache.refresh_transposed()
            for i, layer in enumerate(self.layers):
                sh_i = tuple(bufs[per * i:per * (i + 1)])
                if cache is not None:
                    sh_i = sh_i + (bufs32[i], cache, i)
                elif tcache is not None:
                    sh_i = sh_i + (bufs32[i] if bufs32 is not None else None, None, i, tuple(tcache.views[4 * i:4 * i + 4]))
                elif bufs32 is not None:
                    sh_i = sh_i + (bufs32[i],)
                x = layer(x, attn_mask, sh_i)
        if self.norm is not None:
            x = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x


class _Builder:
    recurrent = False

    def __init__(self, **kw):
        self.kw = kw

    @classmethod
    def from_kwargs(cls, **kw):
        return cls(**kw)

    def get(self):
        kw = self.kw
        if kw.get("attention_type", "causal-linear") != "causal-linear":
            raise ValueError("only attention_type='causal-linear' is built (the one the reference uses)")
        if kw.get("activation", "gelu") != "gelu":
            raise ValueError("only activation='gelu' is built (dqn_policy/model.py:134)")
        H = kw["n_heads"]
        dq, dv = kw["query_dimensions"], kw["value_dimensions"]
        if dq != 64 or dv != 64:
            raise ValueError("the gfx950 scan kernel is specialised for 64-wide heads (got %d/%d)" % (dq, dv))
        d_model = dv * H
        d_ff = kw.get("feed_forward_dimensions", 1024)
        p = kw.get("dropout", 0.1)
        if self.recurrent:
            from .recurrent import RecurrentTransformerEncoder, RecurrentTransformerEncoderLayer
            layers = [RecurrentTransformerEncoderLayer(AttentionLayer(d_model, H, dq, dv), d_model, d_ff, p)
                      for _ in range(kw["n_layers"])]
            return RecurrentTransformerEncoder(layers, nn.LayerNorm(d_model))
        layers = [TransformerEncoderLayer(AttentionLayer(d_model, H, dq, dv), d_model, d_ff, p)
                  for _ in range(kw["n_layers"])]
        return TransformerEncoder(layers, nn.LayerNorm(d_model))


class TransformerEncoderBuilder(_Builder):
    recurrent = False


class RecurrentEncoderBuilder(_Builder):
    recurrent = True
