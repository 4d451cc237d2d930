"""Shared trunk of the compound-word (CW) Linear Transformer: embeddings -> in_linear -> positional
encoding -> causal-linear encoder -> per-attribute heads, on the libcwlt kernels.

The reference declares this network four times (dqn_policy/model.py:97-255 `LinearTransformer`,
dqn_policy/agent_pretrain.py:213 `TransformerModel`, ppo_policy/model.py:98-250 `Actor_Transformer`,
:285-394 `Critic_Transformer`); here it is one base class whose submodule names reproduce the
reference's state_dict keys exactly:
    word_emb_{tempo,chord,barbeat,pitch,duration,velocity}.lut.weight, pos_emb.pe,
    in_linear.{weight,bias}, transformer_encoder.*, proj_{tempo,...}.{weight,bias}

`compute_dtype` selects the storage type of activations: torch.float32 reproduces the reference's
arithmetic (parity mode, logits within 1e-4); torch.bfloat16 is the throughput mode (bf16 GEMMs on
MFMA, f32 scan state / statistics / losses, f32 master weights).  A freshly built net starts in the mode named by
the environment variable CWLT_COMPUTE_DTYPE ("f32", the default, or "bf16"), so the drop-in entry points
(IRL_dqn_train / ppo_train main loops) can be switched to the throughput mode without touching them.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .encoder import RecurrentEncoderBuilder, TransformerEncoderBuilder, TriangularCausalMask

ATTRS = ("tempo", "chord", "barbeat", "pitch", "duration", "velocity")
EMB_SIZES = (128, 256, 64, 512, 128, 128)   # dqn_policy/model.py:110


class Embeddings(nn.Module):
    """lut(x) * sqrt(d_emb) -- dqn_policy/model.py:67-74.  (The trunk fuses the six lookups; this
    module's own forward is the single-table form of the same kernel.)"""

    def __init__(self, n_token, d_model):
        super().__init__()
        self.lut = nn.Embedding(n_token, d_model)
        self.d_model = d_model

    def forward(self, x):
        return ops.cw_embed(x.unsqueeze(-1), [self.lut.weight], torch.float32)


class PositionalEncoding(nn.Module):
    """x + pe[:, :T] then dropout -- dqn_policy/model.py:77-92 (pe is a registered buffer)."""

    def __init__(self, d_model, dropout=0.1, max_len=20000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        p = self.dropout.p if self.training else 0.0
        return ops.PosEncDropoutFn.apply(x, self.pe, p, ops.next_seed() if p > 0 else 0)


def default_compute_dtype():
    import os
    name = os.environ.get("CWLT_COMPUTE_DTYPE", "f32").lower()
    if name in ("f32", "fp32", "float32"):
        return torch.float32
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    raise ValueError("CWLT_COMPUTE_DTYPE must be f32 or bf16, got %r" % name)


class CWTrunk(nn.Module):
    def __init__(self, n_token, d_model, n_layer, n_head, d_inner=2048, dropout=0.1, is_training=True,
                 emb_sizes=EMB_SIZES):
        super().__init__()
        self.d_model, self.n_layer, self.n_head = d_model, n_layer, n_head
        self.d_head = d_model // n_head
        self.dropout, self.d_inner = dropout, d_inner
        self.n_token = list(n_token)
        self.emb_sizes = list(emb_sizes)
        self.compute_dtype = default_compute_dtype()
        for name, n, d in zip(ATTRS, self.n_token, self.emb_sizes):
            setattr(self, "word_emb_" + name, Embeddings(n, d))
        self.pos_emb = PositionalEncoding(d_model, dropout)
        self.in_linear = nn.Linear(int(np.sum(self.emb_sizes)), d_model)
        builder = TransformerEncoderBuilder if is_training else RecurrentEncoderBuilder
        self.transformer_encoder = builder.from_kwargs(
            n_layers=n_layer, n_heads=n_head, query_dimensions=d_model // n_head,
            value_dimensions=d_model // n_head, feed_forward_dimensions=d_inner,
            activation="gelu", dropout=dropout, attention_type="causal-linear").get()
        self._recurrent = not is_training

    # -- pieces ------------------------------------------------------------------------------------
    def _declare_heads(self):
        for name, n in zip(ATTRS, self.n_token):
            setattr(self, "proj_" + name, nn.Linear(self.d_model, n))

    def _tables(self):
        return [getattr(self, "word_emb_" + a).lut.weight for a in ATTRS]

    def _heads(self):
        return [getattr(self, "proj_" + a) for a in ATTRS]

    def embed(self, x):
        """(…, 6) int64 -> (…, d_model): CW embedding + in_linear + positional encoding (+dropout)."""
        if not x.is_cuda:
            raise RuntimeError("rlmg_amd models run on the GPU only (no CPU fallback): move inputs to cuda")
        adt = self.compute_dtype
        embs = ops.cw_embed(x, self._tables(), adt)
        w, b = self.in_linear.weight, self.in_linear.bias
        lead = embs.shape[:-1]
        return ops.linear(embs.reshape(-1, embs.shape[-1]), w, b).view(*lead, w.shape[0])

    def fused_logits(self, h):
        """One 512 x sum(n_token) GEMM for all heads -> (rows, W) with W = sum n_token padded to 64."""
        adt = h.dtype
        heads = self._heads()
        w = torch.cat([m.weight for m in heads], 0)
        b = torch.cat([m.bias for m in heads], 0)
        pad = (-w.shape[0]) % 64
        if pad:
            w = torch.cat([w, w.new_zeros(pad, w.shape[1])], 0)
            b = torch.cat([b, b.new_zeros(pad)], 0)
        return ops.linear(h.reshape(-1, h.shape[-1]), w, b)

    def split_logits(self, logits, lead_shape):
        outs, o = [], 0
        for n in self.n_token:
            outs.append(logits[:, o:o + n].reshape(*lead_shape, n))
            o += n
        return tuple(outs)

    # -- reference surface -------------------------------------------------------------------------
    def compute_loss(self, predict, target, loss_mask):
        """predict (B, C, T) logits, target (B, T), loss_mask (B, T) -- dqn_policy/model.py:163-167."""
        logits = predict.permute(0, 2, 1).reshape(-1, predict.shape[1])
        pad = (-logits.shape[1]) % 4
        if pad:
            logits = torch.nn.functional.pad(logits, (0, pad), value=0.0)
        loss = ops.heads_ce(logits, target.reshape(-1, 1), loss_mask, (predict.shape[1],))
        return loss[0]

    def embed_pos(self, x):
        """(N, T, 6) int64 -> (N, T, d_model): embedding + in_linear + positional encoding + dropout in one pass over
        the token rows (ops.EmbedProjFn: projected tables; the (N, T, 1216) embeddings are never materialised)."""
        if not x.is_cuda:
            raise RuntimeError("rlmg_amd models run on the GPU only (no CPU fallback): move inputs to cuda")
        pe = self.pos_emb
        p = pe.dropout.p if pe.training else 0.0
        return ops.embed_proj(x, self._tables(), self.in_linear.weight, self.in_linear.bias, pe.pe, p,
                              ops.next_seed() if p > 0 else 0, self.compute_dtype)

    def forward_hidden(self, x, memory=None, is_training=True):
        if (is_training and ops.EMBED_PROJ and x.dim() == 3 and not self._recurrent
                and x.shape[0] * x.shape[1] >= ops.EMBED_PROJ_MIN_ROWS):
            pos_emb = self.embed_pos(x)
            attn_mask = TriangularCausalMask(pos_emb.size(1), device=x.device)
            return self.transformer_encoder(pos_emb, attn_mask)
        emb_linear = self.embed(x)
        if is_training:
            if self._recurrent:
                raise RuntimeError("model was built with is_training=False (recurrent encoder)")
            pos_emb = self.pos_emb(emb_linear)
            attn_mask = TriangularCausalMask(pos_emb.size(1), device=x.device)
            return self.transformer_encoder(pos_emb, attn_mask)
        pos_emb = self.pos_emb(emb_linear).squeeze(0)
        return self.transformer_encoder(pos_emb, memory=memory)

    def _losses(self, h, target, loss_mask):
        logits = self.fused_logits(h)
        return ops.heads_ce(logits, target, loss_mask, self.n_token)

    def train_step(self, x, target, loss_mask):
        """-> 6 masked-mean CE losses (tempo, chord, barbeat, pitch, duration, velocity)."""
        h = self.forward_hidden(x)
        losses = self._losses(h, target, loss_mask)
        return tuple(losses[i] for i in range(len(self.n_token)))
