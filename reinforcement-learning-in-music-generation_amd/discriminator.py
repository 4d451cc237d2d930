"""Shared body of the two Longformer discriminators / reward models of the reference:
dqn_policy/AIRL_model.py:46-170 (`LongFormer`, 10 layers, window 50, score classifier) and
ppo_policy/model.py:400-495 (`LongFormer`, 12 layers, window 512, per-attribute `eval_*` heads).
CW embedding (6 tables, widths [128,256,64,512,256,256]) -> `proj` -> Longformer -> heads."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .cw_transformer import ATTRS, Embeddings, default_compute_dtype
from .longformer import LongformerModel

DISC_EMB_SIZES = (128, 256, 64, 512, 256, 256)      # dqn_policy/AIRL_model.py:57


class CWLongformerBase(nn.Module):
    def __init__(self, n_token, d_model, n_layer, n_head, max_seq, attention_window):
        super().__init__()
        self.n_token = list(n_token)
        self.MAX_SEQ, self.D_MODEL, self.N_layer, self.N_head = max_seq, d_model, n_layer, n_head
        self.CE_loss = nn.CrossEntropyLoss()
        self.emb_sizes = list(DISC_EMB_SIZES)
        for name, n, d in zip(ATTRS, self.n_token, self.emb_sizes):
            setattr(self, "word_emb_" + name, Embeddings(n, d))
        self.proj = nn.Linear(sum(self.emb_sizes), d_model)
        for name, n in zip(ATTRS, self.n_token):
            setattr(self, "proj_" + name, nn.Linear(d_model, n))
        self._lf_args = dict(max_position_embeddings=max_seq, hidden_size=d_model, num_hidden_layers=n_layer,
                             num_attention_heads=n_head, intermediate_size=1024, attention_window=attention_window,
                             hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        self.compute_dtype = default_compute_dtype()

    def _build_longformer(self):
        self.longformer = LongformerModel(**self._lf_args)

    def _encode(self, data, masks):
        """(B, T, 6) int64 + (B, T) mask -> (B, T, d_model) last_hidden_state."""
        if not data.is_cuda:
            raise RuntimeError("rlmg_amd models run on the GPU only (no CPU fallback)")
        adt = self.compute_dtype
        self.longformer.compute_dtype = adt
        tabs = [getattr(self, "word_emb_" + a).lut.weight for a in ATTRS]
        emb = ops.cw_embed(data, tabs, adt)
        x = F.linear(emb, self.proj.weight.to(adt), self.proj.bias.to(adt))
        return self.longformer(inputs_embeds=x, attention_mask=masks).last_hidden_state

    def _fused_logits(self, h):
        heads = [getattr(self, "proj_" + a) for a in ATTRS]
        w = torch.cat([m.weight for m in heads], 0)
        b = torch.cat([m.bias for m in heads], 0)
        pad = (-w.shape[0]) % 64
        if pad:
            w = torch.cat([w, w.new_zeros(pad, w.shape[1])], 0)
            b = torch.cat([b, b.new_zeros(pad)], 0)
        return F.linear(h.reshape(-1, h.shape[-1]), w.to(h.dtype), b.to(h.dtype))

    def compute_CEloss(self, predict, target, loss_mask):
        """AIRL_model.py:125-129: CrossEntropyLoss() is already a mean, so (mean * mask).sum() / mask.sum()
        is the plain mean CE."""
        loss = self.CE_loss(predict, target)
        loss = loss * loss_mask
        return torch.sum(loss) / torch.sum(loss_mask)
