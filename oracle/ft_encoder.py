"""fast_transformers 0.4.0 encoder stack -- CPU oracle (TEST INFRASTRUCTURE; see oracle/__init__.py).

Restates, from the published package (idiap/fast-transformers v0.4.0; requirements.txt:54 -- a
third-party dependency that is NOT under /root/reference and not installable here, so this part of
the oracle is "parity unpinned"), exactly the objects the reference builds with

    TransformerEncoderBuilder.from_kwargs(n_layers, n_heads, query_dimensions, value_dimensions,
        feed_forward_dimensions, activation='gelu', dropout=0.1,
        attention_type="causal-linear").get()                       dqn_policy/model.py:128-137
    RecurrentEncoderBuilder.from_kwargs(... same ...).get()          dqn_policy/model.py:141-150

Module / parameter names equal the package's, so state_dict keys match the reference checkpoints:
layers.{i}.attention.{query,key,value,out}_projection, layers.{i}.{linear1,linear2,norm1,norm2},
norm.  Post-LN layers, exact-erf GELU, LayerNorm eps 1e-5, final LayerNorm.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cla


class TriangularCausalMask:
    """masking.TriangularCausalMask(N): only its lower-triangular marker is ever consulted."""

    def __init__(self, N, device="cpu"):
        self.N = N
        self.device = device
        self.lower_triangular = True

    @property
    def bool_matrix(self):
        return torch.tril(torch.ones(self.N, self.N, dtype=torch.bool, device=self.device))


class CausalLinearAttention(nn.Module):
    def __init__(self, query_dimensions, eps=cla.EPS):
        super().__init__()
        self.eps = eps

    def forward(self, queries, keys, values, attn_mask=None):
        if attn_mask is not None and not getattr(attn_mask, "lower_triangular", False):
            raise RuntimeError("CausalLinearAttention only supports full lower triangular masks")
        return cla.cla_reference(queries, keys, values, self.eps)


class RecurrentLinearAttention(nn.Module):
    def __init__(self, query_dimensions, eps=cla.EPS):
        super().__init__()
        self.eps = eps

    def forward(self, query, key, value, state=None):
        return cla.cla_recurrent_step(query, key, value, state, self.eps)


class AttentionLayer(nn.Module):
    """attention_layer.AttentionLayer: four Linear projections around the inner attention."""

    def __init__(self, attention, d_model, n_heads, d_keys, d_values):
        super().__init__()
        self.inner_attention = attention
        self.query_projection = nn.Linear(d_model, d_keys * n_heads)
        self.key_projection = nn.Linear(d_model, d_keys * n_heads)
        self.value_projection = nn.Linear(d_model, d_values * n_heads)
        self.out_projection = nn.Linear(d_values * n_heads, d_model)
        self.n_heads = n_heads

    def forward(self, queries, keys, values, attn_mask):
        N, L, _ = queries.shape
        S = keys.shape[1]
        H = self.n_heads
        q = self.query_projection(queries).view(N, L, H, -1)
        k = self.key_projection(keys).view(N, S, H, -1)
        v = self.value_projection(values).view(N, S, H, -1)
        new_values = self.inner_attention(q, k, v, attn_mask).reshape(N, L, -1)
        return self.out_projection(new_values)


class RecurrentAttentionLayer(AttentionLayer):
    def forward(self, query, key, value, state=None):
        N = query.shape[0]
        H = self.n_heads
        q = self.query_projection(query).view(N, H, -1)
        k = self.key_projection(key).view(N, H, -1)
        v = self.value_projection(value).view(N, H, -1)
        new_value, state = self.inner_attention(q, k, v, state)
        return self.out_projection(new_value.reshape(N, -1)), state


class TransformerEncoderLayer(nn.Module):
    """transformers.TransformerEncoderLayer (post-LN)."""

    def __init__(self, attention, d_model, d_ff, dropout, activation):
        super().__init__()
        self.attention = attention
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.activation = F.relu if activation == "relu" else F.gelu

    def forward(self, x, attn_mask=None):
        x = x + self.dropout(self.attention(x, x, x, attn_mask))
        y = x = self.norm1(x)
        y = self.dropout(self.activation(self.linear1(y)))
        y = self.dropout(self.linear2(y))
        return self.norm2(x + y)


class RecurrentTransformerEncoderLayer(TransformerEncoderLayer):
    def forward(self, x, state=None):
        x2, state = self.attention(x, x, x, state)
        x = x + self.dropout(x2)
        y = x = self.norm1(x)
        y = self.dropout(self.activation(self.linear1(y)))
        y = self.dropout(self.linear2(y))
        return self.norm2(x + y), state


class TransformerEncoder(nn.Module):
    def __init__(self, layers, norm_layer=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.norm = norm_layer

    def forward(self, x, attn_mask=None, length_mask=None):
        for layer in self.layers:
            x = layer(x, attn_mask)
        if self.norm is not None:
            x = self.norm(x)
        return x


class RecurrentTransformerEncoder(nn.Module):
    def __init__(self, layers, norm_layer=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.norm = norm_layer

    def forward(self, x, state=None, memory=None):
        if memory is not None and state is None:   # `memory=` is the package's deprecated alias
            state = memory
        if state is None:
            state = [None] * len(self.layers)
        state = list(state)
        for i, layer in enumerate(self.layers):
            x, s = layer(x, state[i])
            state[i] = s
        if self.norm is not None:
            x = self.norm(x)
        return x, state


class _Builder:
    _recurrent = False

    def __init__(self, **kw):
        self.kw = kw

    @classmethod
    def from_kwargs(cls, **kw):
        return cls(**kw)

    def get(self):
        kw = self.kw
        if kw.get("attention_type", "causal-linear") != "causal-linear":
            raise ValueError("oracle restates attention_type='causal-linear' only")
        H = kw["n_heads"]
        dq, dv = kw["query_dimensions"], kw["value_dimensions"]
        d_model = dv * H
        d_ff = kw.get("feed_forward_dimensions", 1024)
        act = kw.get("activation", "relu")
        p = kw.get("dropout", 0.1)
        layers = []
        for _ in range(kw["n_layers"]):
            if self._recurrent:
                att = RecurrentAttentionLayer(RecurrentLinearAttention(dq), d_model, H, dq, dv)
                layers.append(RecurrentTransformerEncoderLayer(att, d_model, d_ff, p, act))
            else:
                att = AttentionLayer(CausalLinearAttention(dq), d_model, H, dq, dv)
                layers.append(TransformerEncoderLayer(att, d_model, d_ff, p, act))
        enc = RecurrentTransformerEncoder if self._recurrent else TransformerEncoder
        return enc(layers, nn.LayerNorm(d_model))


class TransformerEncoderBuilder(_Builder):
    _recurrent = False


class RecurrentEncoderBuilder(_Builder):
    _recurrent = True
