"""Recurrent (one token per call) form of the causal-linear encoder, for generation.

Surface of fast_transformers' RecurrentEncoderBuilder product as the reference uses it
(dqn_policy/model.py:141-150,236-238; dqn_policy/testing-no-type-cp.py:126-179):
    h, memory = encoder(x (N, d_model), memory=memory)
with per-layer state [S (N, H, 64, 64), Zs (N, H, 64)].  Parameter names equal the training
encoder's, so checkpoints interchange.

Scope note: generation is OUTSIDE the training hot path this round.  The
step below is latency-bound GEMV-sized work expressed with torch GPU ops (no CPU path); a persistent
single-kernel decode step is listed as the next widening step.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

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
        q = F.elu(at.query_projection(x).view(N, H, -1)) + 1
        k = F.elu(at.key_projection(x).view(N, H, -1)) + 1
        v = at.value_projection(x).view(N, H, -1)
        if state is None:
            S = x.new_zeros((N, H, q.shape[-1], v.shape[-1]))
            Zs = x.new_zeros((N, H, q.shape[-1]))
        else:
            S, Zs = state
            if len(S) != N:
                raise ValueError("The batch size changed during iteration")
        Zs = Zs + k
        S = S + torch.einsum("nhd,nhm->nhdm", k, v)
        Z = 1.0 / (torch.einsum("nhd,nhd->nh", q, Zs) + ops.CLA_EPS)
        a = torch.einsum("nhd,nhdm,nh->nhm", q, S, Z).reshape(N, -1)
        x = self.norm1(x + self.dropout(at.out_projection(a)))
        y = self.dropout(F.gelu(self.linear1(x)))
        y = self.dropout(self.linear2(y))
        return self.norm2(x + y), [S, Zs]


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
            x = self.norm(x)
        return x, state
