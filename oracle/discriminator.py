"""AIRL discriminator / PPO reward model -- CPU oracle (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows /root/reference/dqn_policy/AIRL_model.py:101-122 (`LongFormer.forward`: 6 embeddings * sqrt(d) ->
cat -> proj -> Longformer -> mean over seq -> score_classifier) and ppo_policy/model.py:459-495
(`LongFormer.token_forward`: ... -> 6 heads -> eval_f -> mean over seq -> sigmoid -> mean of 6), operating on
the modules' state dicts.  Pinned by tests/golden/{airl_small,ppo_reward_small}.npz, recorded from the
reference's own classes (tests/golden/make_golden.py).
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


def airl_forward(sd, data, masks, n_layer, n_head, attention_window, bn_eps=1e-5):
    """eval-mode forward of dqn_policy/AIRL_model.py::LongFormer -> (B, 1)."""
    x = _embed_proj(sd, data)
    h = olf.longformer_forward(sd, x, masks, n_layer, n_head, attention_window // 2, prefix="longformer.")
    m = h.mean(dim=1)
    y = F.linear(m, sd["score_classifier.0.weight"], sd["score_classifier.0.bias"])
    y = F.batch_norm(y, sd["score_classifier.1.running_mean"], sd["score_classifier.1.running_var"],
                     sd["score_classifier.1.weight"], sd["score_classifier.1.bias"], False, 0.1, bn_eps)
    y = torch.tanh(y)
    y = torch.tanh(F.linear(y, sd["score_classifier.3.weight"], sd["score_classifier.3.bias"]))
    return torch.sigmoid(F.linear(y, sd["score_classifier.5.weight"], sd["score_classifier.5.bias"]))


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
