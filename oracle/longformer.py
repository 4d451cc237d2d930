"""Longformer encoder as the reference uses it -- CPU oracle (TEST INFRASTRUCTURE; see oracle/__init__.py).

Restates HF transformers `LongformerModel(inputs_embeds=..., attention_mask=...)` (modeling_longformer.py:
LongformerEmbeddings, LongformerSelfAttention sliding-window path with no global attention,
LongformerSelfOutput / Intermediate / Output) for the configs at dqn_policy/AIRL_model.py:78-90 and
ppo_policy/model.py:440-451, in dense form: full (L x L) scores with the band and key masks applied.
Pinned against the HF class itself (transformers 5.15.0 in this image; the reference pins 4.40.2 -- the
banded-attention arithmetic is the long-standing one) by tests/test_oracle_longformer.py.
Works directly on an HF-style state dict, so no module tree is needed.
"""
import math

import torch
import torch.nn.functional as F


def band_attention(q, k, v, mask, window):
    """q, k, v (B, L, H, D); mask (B, L) nonzero = attend or None; window one-sided.  -> (B, L, H*D)."""
    B, L, H, D = q.shape
    s = torch.einsum("blhd,bshd->bhls", q / math.sqrt(D), k)
    idx = torch.arange(L)
    band = (idx[:, None] - idx[None, :]).abs() <= window
    allow = band[None, None].expand(B, H, L, L).clone()
    if mask is not None:
        allow &= (mask != 0)[:, None, None, :]
    s = s.masked_fill(~allow, float("-inf"))
    p = torch.softmax(s.float(), -1).to(q.dtype)
    p = torch.nan_to_num(p, nan=0.0)
    if mask is not None:
        p = p * (mask != 0)[:, None, :, None].to(p.dtype)      # masked query rows -> 0
    return torch.einsum("bhls,bshd->blhd", p, v).reshape(B, L, H * D)


def longformer_forward(sd, x, mask, n_layer, n_head, window, prefix="", eps=1e-12, pad_id=1):
    """sd: HF-keyed state dict (tensors); x (B, L, hidden) inputs_embeds; mask (B, L) or None."""
    B, L, Dm = x.shape
    g = lambda k: sd[prefix + k]
    h = x + g("embeddings.position_embeddings.weight")[pad_id + 1: pad_id + 1 + L] \
        + g("embeddings.token_type_embeddings.weight")[0]
    h = F.layer_norm(h, (Dm,), g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
    for i in range(n_layer):
        p = "encoder.layer.%d." % i
        q = F.linear(h, g(p + "attention.self.query.weight"), g(p + "attention.self.query.bias"))
        k = F.linear(h, g(p + "attention.self.key.weight"), g(p + "attention.self.key.bias"))
        v = F.linear(h, g(p + "attention.self.value.weight"), g(p + "attention.self.value.bias"))
        a = band_attention(q.view(B, L, n_head, -1), k.view(B, L, n_head, -1), v.view(B, L, n_head, -1), mask, window)
        o = F.linear(a, g(p + "attention.output.dense.weight"), g(p + "attention.output.dense.bias"))
        h1 = F.layer_norm(o + h, (Dm,), g(p + "attention.output.LayerNorm.weight"),
                          g(p + "attention.output.LayerNorm.bias"), eps)
        it = F.gelu(F.linear(h1, g(p + "intermediate.dense.weight"), g(p + "intermediate.dense.bias")))
        y = F.linear(it, g(p + "output.dense.weight"), g(p + "output.dense.bias"))
        h = F.layer_norm(y + h1, (Dm,), g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), eps)
    return h
