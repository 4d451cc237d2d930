"""Recurrent (one token per call) form of the causal-linear encoder, for generation.

Surface of fast_transformers' RecurrentEncoderBuilder product as the reference uses it
(dqn_policy/model.py:141-150,236-238; dqn_policy/testing-no-type-cp.py:126-179):
    h, memory = encoder(x (N, d_model), memory=memory)
with per-layer state [S (N, H, 64, 64), Zs (N, H, 64)].  Parameter names equal the training
encoder's, so checkpoints interchange.

The per-token attention step (state update + normalised read-out, elu+1 inside) is one libcwlt kernel
(csrc/recurrent.hip); LayerNorm / FFN activation reuse the training kernels; the projections are
GEMV-sized hipBLASLt calls.  Generation is outside the training hot path: a persistent
single-launch decode step across all 12 layers is the natural next step.
"""
import torch
import torch.nn as nn

from . import ops


class RecurrentTransformerEncoderLayer(nn.Module):
    def __init__(self, attention, d_model, d_ff, dropout):
        super().__init__()
        self.attention = attention
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, state=None):
        at = self.attention
        N, H = x.shape[0], at.n_heads
        D = x.shape[1]
        with torch.no_grad():
            wqkv = torch.cat([at.query_projection.weight, at.key_projection.weight, at.value_projection.weight], 0)
            bqkv = torch.cat([at.query_projection.bias, at.key_projection.bias, at.value_projection.bias], 0)
            qkv = torch.addmm(bqkv.to(x.dtype), x, wqkv.to(x.dtype).t())            # (N, 3D)
            if state is None:
                S = torch.zeros((N, H, D // H, D // H), dtype=torch.float32, device=x.device)
                Zs = torch.zeros((N, H, D // H), dtype=torch.float32, device=x.device)
            else:
                S, Zs = state
                if len(S) != N:
                    raise ValueError("The batch size changed during iteration")
            a = ops.recurrent_cla_step(qkv, S, Zs, H)                                # state updated in place
            p = self.dropout.p if self.training else 0.0
            o = at.out_projection(a)
            _, x1, _, _ = ops.ln_fwd(x.contiguous(), o, ops._f32(self.norm1.weight), ops._f32(self.norm1.bias),
                                     self.norm1.eps, p, ops.next_seed() if p > 0 else 0, save_s=False)
            h = torch.mm(x1, self.linear1.weight.to(x.dtype).t())
            g = ops.gelu_fwd(h, ops._f32(self.linear1.bias), p, ops.next_seed() if p > 0 else 0)
            y = self.linear2(g)
            _, x2, _, _ = ops.ln_fwd(x1, y, ops._f32(self.norm2.weight), ops._f32(self.norm2.bias), self.norm2.eps, p,
                                     ops.next_seed() if p > 0 else 0, save_s=False)
        return x2, [S, Zs]


class RecurrentTransformerEncoder(nn.Module):
    def __init__(self, layers, norm_layer=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.norm = norm_layer

    def forward(self, x, state=None, memory=None):
        if not x.is_cuda:
            raise RuntimeError("rlmg_amd encoder runs on the GPU only (no CPU fallback)")
        if state is None:
            state = memory
        if state is None:
            state = [None] * len(self.layers)
        state = list(state)
        for i, layer in enumerate(self.layers):
            x, state[i] = layer(x, state[i])
        if self.norm is not None:
            with torch.no_grad():
                x = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x, state
