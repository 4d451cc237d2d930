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

from . import ops


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

        if ops.gemm_bf16_supported(x2, wqkv):
            if bqkv32 is None:
                bqkv32 = torch.cat([ops._f32(bq), ops._f32(bk), ops._f32(bv)])
            qkv = ops.gemm_bf16(x2, wqkv, bqkv32)                          # (R, 3D)  MFMA
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
            o = (ops.gemm_bf16(a2, wo_a, ops._f32(bo)) if ops.gemm_bf16_supported(a2, wo_a)
                 else torch.addmm(bo_a, a2, wo_a.t()))                     # MFMA
            s1, x1, mean1, rstd1 = ops.ln_fwd(x2, o, g1f, be1f, ops.LN_EPS, p, seeds[0])
            del o
        # bf16, when a backward will follow: the activation pass also leaves gd = mask / (1 - p) * gelu'(h + b1) (in place
        # of h); the backward then needs no activation pass at all -- dh = (dy . W2) * gd leaves the input-gradient GEMM's
        # epilogue (ops.gemm_nt_mul).  The forward does the same from its side: linear1, bias, GELU and dropout are ONE
        # kernel (ops.ffn1_gelu_dropout), the pre-activation never reaches HBM.
        fused_ffn = (FUSED_FFN_BWD and adt == torch.bfloat16 and will_backward
                     and D % 64 == 0 and w1_a.shape[0] % 256 == 0)
        if fused_ffn and FUSED_FFN_FWD and ops.ffn1_fused_supported(x1, w1_a, b1f):
            g, h = ops.ffn1_gelu_dropout(x1, w1_a, b1f, p, seeds[1])      # h holds gd
        else:
            h = (ops.gemm_bf16(x1, w1_a) if ops.gemm_bf16_supported(x1, w1_a)
                 else torch.mm(x1, w1_a.t()))                              # (R, F)  MFMA, bias in next kernel
            fused_ffn = fused_ffn and h.is_contiguous()
            g = ops.gelu_fwd(h, b1f, p, seeds[1], gd_inplace=fused_ffn)
        if ops.linear_ln_supported(g, w2_a, x1):
            s2, out, mean2, rstd2 = ops.linear_ln(g, w2_a, ops._f32(b2), x1, g2f, be2f, ops.LN_EPS, p, seeds[2])
        else:
            y = (ops.gemm_bf16(g, w2_a, ops._f32(b2)) if ops.gemm_bf16_supported(g, w2_a)
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
        def dgrad(g_, w_, out=None):
            if ops.gemm_bf16_supported(g_, w_.t(), out, transposed_w=True):
                return ops.gemm_bf16(g_, w_.t().contiguous(), out=out, accumulate=out is not None)
            if out is not None:
                return out.addmm_(g_, w_.t().contiguous().t() if nt else w_)
            return torch.mm(g_, w_.t().contiguous().t()) if nt else torch.mm(g_, w_)

        dw2 = wgrad(dy, g)                                                 # (D, F)
        if ctx.fused_ffn:
            # h holds gd (see forward): input gradient of linear2, activation + dropout backward and linear1's bias
            # gradient in ONE kernel; dgact (R x F) never reaches memory
            dh, db1 = ops.gemm_nt_mul(dy, w2_a.t().contiguous(), h)
        else:
            dgact = dgrad(dy, w2_a)                                        # (R, F)
            dh, db1 = ops.gelu_bwd(dgact, h, b1f, p, seeds[1])
            del dgact
        if DX1_IN_PLACE:
            # linear1's input gradient lands ON the residual gradient (C = D = ds2, beta = 1): norm1's backward then
            # reads one gradient stream instead of two, and the extra read sits in a GEMM that has HBM time to spare.
            # dy may alias ds2 (p = 0); its last readers (dw2, dh) are already enqueued.
            dgrad(dh, w1_a, out=ds2)                                       # (R, D)
            dx1 = None
        else:
            dx1 = dgrad(dh, w1_a)                                          # (R, D)
        dw1 = wgrad(dh, x1)                                                # (F, D)
        del dh
        ds1, do, dg1, dbe1, dbo = ops.ln_bwd(ds2, dx1, s1, g1f, mean1, rstd1, p, seeds[0])
        da = dgrad(do, wo_a)                                               # (R, D)
        dwo = wgrad(do, a.view(R, D))
        qkv5 = qkv.view(N, L, 3, H, D // H)
        dqkv, dbqkv = ops.cla_bwd(qkv5[:, :, 0], qkv5[:, :, 1], qkv5[:, :, 2], a, zinv, da.view(N, L, H, D // H),
                                  want_colsum=True, final_state=ctx.fin)
        ctx.fin = None
        dqkv2 = dqkv.view(R, 3 * D)
        if ops.gemm_bf16_supported(dqkv2, wqkv.t(), ds1, transposed_w=True):
            dx = ops.gemm_bf16(dqkv2, wqkv.t().contiguous(), out=ds1, accumulate=True)
        else:
            dx = ds1.addmm_(dqkv2, wqkv)                                   # residual + projection gradient, in place
                                                                           # (NN is hipBLASLt's faster form for this shape)
                                                                           # (out-of-place addmm first copies ds1: 268 MB)
        dwqkv = wgrad(dqkv2, x2)                                           # (3D, D)
        layer = ctx.layer
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
            return (dx.view(N, L, D),) + (None,) * 22
        dwqkv = dwqkv.float()
        return (dx.view(N, L, D),
                dwqkv[:D], dbqkv[:D], dwqkv[D:2 * D], dbqkv[D:2 * D], dwqkv[2 * D:], dbqkv[2 * D:],
                dwo.float(), dbo, dw1.float(), db1, dw2.float(), db2, dg1, dbe1, dg2, dbe2,
                None, None, None, None, None, None)


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
        if x.dtype == torch.bfloat16 and x.shape[0] * x.shape[1] >= ops.GEMM_BF16_MIN_ROWS and ops.GEMM_BF16:
            s32 = self._shadow32
            if s32 is None or not s32.matches(torch.float32, x.device):
                s32 = self._shadow32 = ops.ShadowSet([layer.shadow_groups()[1] for layer in self.layers], torch.float32)
            bufs32 = s32.refresh()
        for i, layer in enumerate(self.layers):
            sh_i = tuple(bufs[per * i:per * (i + 1)])
            x = layer(x, attn_mask, sh_i + (bufs32[i],) if bufs32 is not None else sh_i)
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
