"""Drop-in for /root/reference/dqn_policy/AIRL_model.py: `LongFormer`, the AIRL discriminator
(10-layer Longformer, window 50, mean-pool + score classifier) on the libcwlt kernels.  Differentiable when
autograd is enabled (discriminator training, AIRL.py:135-170); scoring callers wrap it in torch.no_grad()."""
import os
import sys

import torch
import torch.nn as nn

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import rlmg_amd  # noqa: E402,F401
from rlmg_amd import ops  # noqa: E402
from rlmg_amd.cw_transformer import ATTRS, Embeddings  # noqa: E402,F401
from rlmg_amd.discriminator import CWLongformerBase  # noqa: E402

# --- modules config (AIRL_model.py:22-27) --- #
MAX_SEQ_LEN = 1024
D_MODEL = 512
N_LAYER = 10
N_HEAD = 8
path_exp = "exp"
N_STATES = 50


class LongFormer(CWLongformerBase):
    def __init__(self, n_token):
        super().__init__(n_token, D_MODEL, N_LAYER, N_HEAD, MAX_SEQ_LEN * 2, attention_window=N_STATES)
        print("Disc token >>>>> ", self.n_token)
        self._build_longformer()
        self.score_classifier = nn.Sequential(
            nn.Linear(D_MODEL, 128), nn.BatchNorm1d(128), nn.Tanh(),
            nn.Linear(128, 64), nn.Tanh(), nn.Linear(64, 1), nn.Sigmoid())

    def forward(self, data, masks):
        """data (B, window, 6), masks (B, window) -> (B, 1) in (0, 1)   (AIRL_model.py:101-122)."""
        seq = self._encode(data, masks)
        return self.score_classifier(seq.float().mean(dim=1))

    def score_in_groups(self, data, masks, group, windows_per_pass=2000):
        """forward() of consecutive batches of `group` windows, (n, window, 6) -> (n, 1), with the Longformer body
        run over up to `windows_per_pass` windows at a time.  Equal to calling forward() batch by batch: every
        op before the score classifier acts on one window (or one token) alone; the classifier's BatchNorm1d is
        what sees a batch, so it is applied -- running statistics included -- per group (`_classify_groups`)."""
        n = data.shape[0]
        if n % group:
            raise ValueError("score_in_groups needs a whole number of groups")
        means = []
        step = windows_per_pass - windows_per_pass % group
        with ops.frozen_weights():                 # one scoring call: the weight copies are refreshed once, not per pass
            for s in range(0, n, step):
                e = min(n, s + step)
                means.append(self._encode(data[s:e], masks[s:e]).float().mean(dim=1))
        return self._classify_groups(torch.cat(means, 0), group)

    def _classify_groups(self, mean, group):
        """score_classifier on consecutive groups of `group` rows of `mean` (n, d_model), all groups in one pass.
        BatchNorm1d in train mode normalises each group with its own statistics and moves the running statistics
        once per group, in order: running <- (1 - m) running + m stat_g, unrolled here to a weighted sum."""
        lin0, bn, _, lin3, _, lin5, _ = self.score_classifier
        F = torch.nn.functional
        if not (bn.training and bn.track_running_stats and bn.momentum is not None):
            return torch.cat([self.score_classifier(mean[g0:g0 + group]) for g0 in range(0, mean.shape[0], group)], 0)
        G = mean.shape[0] // group
        z = F.linear(mean, lin0.weight, lin0.bias).view(G, group, -1)
        mu = z.mean(dim=1)
        var_b = z.var(dim=1, unbiased=False)
        zn = (z - mu[:, None]) * torch.rsqrt(var_b[:, None] + bn.eps) * bn.weight + bn.bias
        with torch.no_grad():
            m = bn.momentum
            w = m * (1.0 - m) ** torch.arange(G - 1, -1, -1, device=mean.device, dtype=mean.dtype)
            var_u = var_b * (group / max(group - 1, 1))
            bn.running_mean.mul_((1.0 - m) ** G).add_((w[:, None] * mu).sum(0))
            bn.running_var.mul_((1.0 - m) ** G).add_((w[:, None] * var_u).sum(0))
            bn.num_batches_tracked += G
        y = torch.tanh(zn.view(G * group, -1))
        y = torch.tanh(F.linear(y, lin3.weight, lin3.bias))
        return torch.sigmoid(F.linear(y, lin5.weight, lin5.bias))

    def token_forward(self, data, target, loss_mask):
        """Mean of the 6 token CE losses of the discriminator's heads (AIRL_model.py:131-170).  compute_CEloss
        there multiplies an already-meaned CE by the mask and divides by its sum, i.e. the plain mean CE."""
        h = self._encode(data, loss_mask)
        logits = self._fused_logits(h)
        rows = logits.shape[0]
        ones = torch.ones(rows, device=logits.device)
        tgt = target.reshape(rows, len(self.n_token))
        if torch.is_grad_enabled():
            return ops.heads_ce(logits, tgt, ones, self.n_token).sum() / len(self.n_token)
        res = ops.heads_forward(logits, self.n_token, tgt, ones)
        return (res["loss_sum"] / rows).sum() / len(self.n_token)
