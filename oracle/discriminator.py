"""AIRL discriminator / PPO reward model -- CPU oracle (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows /root/reference/dqn_policy/AIRL_model.py:101-122 (`LongFormer.forward`: 6 embeddings * sqrt(d) ->
cat -> proj -> Longformer -> mean over seq -> score_classifier) and ppo_policy/model.py:459-495
(`LongFormer.token_forward`: ... -> 6 heads -> eval_f -> mean over seq -> sigmoid -> mean of 6), operating on
the modules' state dicts.  Pinned by tests/golden/{airl_small,ppo_reward_small}.npz, recorded from the
reference's own classes (tests/golden/make_golden.py).  `airl_token_ce` / `airl_disc_loss` restate
AIRL_model.py:131-170 and the loss of AIRL.py:150-170; every function is plain differentiable torch, and the
gradients are pinned by tests/golden/airl_grads_small.npz (the reference's own backward).
"""
import math

import torch
import torch.nn.functional as F

from . import longformer as olf

ATTRS = ("tempo", "chord", "barbeat", "pitch", "duration", "velocity")


def _embed_proj(sd, data):
    embs = [F.embedding(data[..., i], sd["word_emb_%s.lut.weight" % a]) * math.sqrt(sd["word_emb_%s.lut.weight" % a].shape[1])
            for i, a in enumerate(ATTRS)]
    return F.linear(torch.cat(embs, -1), sd["proj.weight"], sd["proj.bias"])


def airl_forward(sd, data, masks, n_layer, n_head, attention_window, bn_eps=1e-5, batch_stats=False):
    """forward of dqn_policy/AIRL_model.py::LongFormer -> (B, 1), dropout off.  batch_stats=False: eval mode
    (BatchNorm1d on its running statistics).  batch_stats=True: BatchNorm1d on the statistics of THIS batch -- what
    `RewardDiscri.all_forward` computes, which forces train() even while scoring (dqn_policy/AIRL.py:61-65)."""
    x = _embed_proj(sd, data)
    h = olf.longformer_forward(sd, x, masks, n_layer, n_head, attention_window // 2, prefix="longformer.")
    m = h.mean(dim=1)
    y = F.linear(m, sd["score_classifier.0.weight"], sd["score_classifier.0.bias"])
    if batch_stats:
        y = F.batch_norm(y, None, None, sd["score_classifier.1.weight"], sd["score_classifier.1.bias"], True, 0.1, bn_eps)
    else:
        y = F.batch_norm(y, sd["score_classifier.1.running_mean"], sd["score_classifier.1.running_var"],
                         sd["score_classifier.1.weight"], sd["score_classifier.1.bias"], False, 0.1, bn_eps)
    y = torch.tanh(y)
    y = torch.tanh(F.linear(y, sd["score_classifier.3.weight"], sd["score_classifier.3.bias"]))
    return torch.sigmoid(F.linear(y, sd["score_classifier.5.weight"], sd["score_classifier.5.bias"]))


def calculate_reward(sd, states, mask_states, batch_size, n_layer, n_head, attention_window):
    """dqn_policy/AIRL.py:69-91 `RewardDiscri.calculate_reward` with dropout off: the buffer is scored in consecutive
    batches of `batch_size` through `all_forward` (train mode: BatchNorm on each batch's own statistics); a tail
    shorter than a batch is never scored and keeps the initial 1.0."""
    n = states.shape[0]
    pred = torch.ones((n, 1))
    for idx in range(n // batch_size):
        s, e = idx * batch_size, (idx + 1) * batch_size
        pred[s:e] = airl_forward(sd, states[s:e].long(), mask_states[s:e].long(), n_layer, n_head, attention_window,
                                 batch_stats=True)
    return pred


def ppo_reward_forward(sd, data, masks, n_layer, n_head, attention_window):
    """ppo_policy/model.py::LongFormer.token_forward -> (B, 1)."""
    x = _embed_proj(sd, data)
    h = olf.longformer_forward(sd, x, masks, n_layer, n_head, attention_window // 2, prefix="longformer.")
    total = 0
    for a in ATTRS:
        y = F.linear(h, sd["proj_%s.weight" % a], sd["proj_%s.bias" % a])
        hid = F.linear(y, sd["eval_%s.weight" % a], sd["eval_%s.bias" % a]).mean(dim=1)
        total = total + torch.sigmoid(hid)
    return total / len(ATTRS)


def airl_token_ce(sd, data, target, masks, n_layer, n_head, attention_window):
    """dqn_policy/AIRL_model.py:131-170 `token_forward`: mean over the 6 heads of compute_CEloss, which is
    (mean CE * mask).sum() / mask.sum() == the plain mean CE over all positions (:125-129)."""
    x = _embed_proj(sd, data)
    h = olf.longformer_forward(sd, x, masks, n_layer, n_head, attention_window // 2, prefix="longformer.")
    total = 0
    for i, a in enumerate(ATTRS):
        y = F.linear(h, sd["proj_%s.weight" % a], sd["proj_%s.bias" % a])
        total = total + F.cross_entropy(y.reshape(-1, y.shape[-1]), target[..., i].reshape(-1))
    return total / len(ATTRS)


def airl_disc_loss(sd, x_exp, x_agent, masks, n_layer, n_head, attention_window):
    """One batch of dqn_policy/AIRL.py:150-170 in eval mode: (BCE(expert, 1), BCE(agent, 0), token CE)."""
    s_exp = airl_forward(sd, x_exp, masks, n_layer, n_head, attention_window)
    s_ag = airl_forward(sd, x_agent, masks, n_layer, n_head, attention_window)
    ce = airl_token_ce(sd, x_agent, x_exp, masks, n_layer, n_head, attention_window)
    return (F.binary_cross_entropy(s_exp, torch.ones_like(s_exp)),
            F.binary_cross_entropy(s_ag, torch.zeros_like(s_ag)), ce)
