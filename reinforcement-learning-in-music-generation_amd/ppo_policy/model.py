"""Drop-in for /root/reference/ppo_policy/model.py: `Actor_Transformer`, `Critic_Transformer`,
`LongFormer` (reward model), samplers, `network_paras` -- on the MI355X-native trunk.

Run with this directory as the working directory: `from model import Actor_Transformer, ...`.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import rlmg_amd  # noqa: E402,F401
from rlmg_amd import ops  # noqa: E402
from rlmg_amd.cw_transformer import ATTRS, CWTrunk, Embeddings, PositionalEncoding  # noqa: E402,F401
from rlmg_amd.discriminator import CWLongformerBase  # noqa: E402
from rlmg_amd.sampling import nucleus, sample_cw, sampling, softmax_with_temperature, weighted_sampling  # noqa: E402,F401

try:
    from config import ActorConfig, DiscriConfig
except ImportError:
    from .config import ActorConfig, DiscriConfig


def network_paras(model):
    return sum(int(np.prod(p.size())) for p in model.parameters() if p.requires_grad)


class Actor_Transformer(CWTrunk):
    """ppo_policy/model.py:98-280: the CW trunk + `value_funtion` MLP (512 -> 128 -> 1)."""

    def __init__(self, n_token, is_training=True):
        super().__init__(n_token, ActorConfig["D_MODEL"], ActorConfig["N_LAYER"], ActorConfig["N_HEAD"],
                         d_inner=2048, dropout=0.1, is_training=is_training)
        self.loss_func = nn.CrossEntropyLoss(reduction="none")
        print("Token_class >>>>>:", self.n_token)
        if not is_training:
            print(" [o] using RNN backend.")
        self.value_funtion = nn.Sequential(nn.Linear(self.d_model, 128), nn.ReLU(), nn.Linear(128, 1))
        self._declare_heads()

    def forward_output(self, h):
        return self.split_logits(self.fused_logits(h), h.shape[:-1])

    def forward_output_sampling(self, h):
        y = [t.float() for t in self.forward_output(h)]
        return sample_cw(y)             # ppo_policy/model.py:264-269 draw order


class Critic_Transformer(CWTrunk):
    """ppo_policy/model.py:285-394: trunk + 6 heads + 6 `*_value` Linear(n_f -> 1); value =
    mean over the sequence, mean over the 6 attributes."""

    def __init__(self, n_token):
        super().__init__(n_token, ActorConfig["D_MODEL"], ActorConfig["N_LAYER"], ActorConfig["N_HEAD"],
                         d_inner=2048, dropout=0.1, is_training=True)
        self.loss_func = nn.CrossEntropyLoss(reduction="none")
        self._declare_heads()
        for name, n in zip(ATTRS, self.n_token):
            setattr(self, name + "_value", nn.Linear(n, 1))

    def value_produce(self, x):
        """(B, T, 6) -> (B, 1).  Everything behind the trunk is linear -- Linear(512 -> n_f), Linear(n_f -> 1), the mean
        over T, the mean over the attributes (ppo_policy/model.py:345-394) -- so the mean over T is taken FIRST, on the
        hidden states (f32 accumulation), and the heads see B rows instead of B * T: no (B * T, sum n_f) logits, no GEMM
        over the token rows in either direction (the backward of a mean is a broadcast)."""
        h = self.forward_hidden(x)
        if os.environ.get("CWLT_CRITIC_MEAN_FIRST", "1") != "0":
            hm = h.mean(dim=1, dtype=torch.float32)                                    # (B, D)
            heads = self._heads()
            logits = torch.nn.functional.linear(hm, torch.cat([m.weight for m in heads], 0),
                                                torch.cat([m.bias for m in heads], 0))   # (B, sum n_f) = T-mean logits
        else:
            B, T = h.shape[0], h.shape[1]
            logits = self.fused_logits(h).float().view(B, T, -1).mean(dim=1)          # (B, W)
        total = 0
        o = 0
        for name, n in zip(ATTRS, self.n_token):
            head = getattr(self, name + "_value")
            total = total + torch.nn.functional.linear(logits[:, o:o + n], head.weight, head.bias)
            o += n
        return total / len(self.n_token)


LinearTransformer = Actor_Transformer     # the name ppo_policy/my_pretrain.py:18 imports (its model.py never defines it)


class LongFormer(CWLongformerBase):
    """ppo_policy/model.py:400-495: the reward model (frozen inside ppo_train.py, trained by
    `my_pretrain.py --reward_pretrain`).  12-layer Longformer with attention_window = D_MODEL (512); the reference
    pads the 50-token window to 512 -- here the band simply covers the window."""

    def __init__(self, n_token):
        super().__init__(n_token, DiscriConfig["D_MODEL"], DiscriConfig["N_LAYER"], DiscriConfig["N_HEAD"],
                         DiscriConfig["MAX_SEQ"], attention_window=DiscriConfig["D_MODEL"])
        for name, n in zip(ATTRS, self.n_token):
            setattr(self, "eval_" + name, nn.Linear(n, 1))
        self.sigmoid = nn.Sigmoid()
        self._build_longformer()
        self.longformer_config = dict(self._lf_args, position_embedding_type="relative_key", hidden_act="gelu")

    def token_forward(self, data, target, loss_mask):
        """(B, T, 6), _, (B, T) -> (B, 1): mean over the 6 attributes of sigmoid(mean_T eval_f(proj_f(h))).
        `target` is unused, as in the reference (:459).  Differentiable, as in the reference, when autograd is on
        (the RL loop calls it under torch.no_grad(): ppo_train.py::_rollout_step_device)."""
        h = self._encode(data, loss_mask)
        B, T = h.shape[0], h.shape[1]
        logits = self._fused_logits(h).float().view(B, T, -1).mean(dim=1)
        total, o = 0, 0
        for name, n in zip(ATTRS, self.n_token):
            head = getattr(self, "eval_" + name)
            total = total + torch.sigmoid(torch.nn.functional.linear(logits[:, o:o + n], head.weight, head.bias))
            o += n
        return total / len(self.n_token)

    def train_step(self, x, target, loss_mask):
        """What `my_pretrain.py --reward_pretrain` calls on this class (my_pretrain.py:78,184).  The reference's
        LongFormer never defines it (that branch raises AttributeError as committed); the losses it has the pieces
        for are the six per-attribute `compute_CEloss(proj_f(h), target_f, mask)` -- the same token CE its DQN-side
        twin computes (dqn_policy/AIRL_model.py:131-170) -- so that is what this returns: 6 losses, each
        (mean CE * mask).sum() / mask.sum() = the plain mean CE over all positions (compute_CEloss's arithmetic)."""
        h = self._encode(x, loss_mask)
        logits = self._fused_logits(h)
        rows = logits.shape[0]
        ones = torch.ones(rows, device=logits.device)
        losses = ops.heads_ce(logits, target.reshape(rows, len(self.n_token)), ones, self.n_token)
        return tuple(losses[i] for i in range(len(self.n_token)))
