"""Drop-in for /root/reference/dqn_policy/model.py: `LinearTransformer`, `Embeddings`,
`PositionalEncoding`, the numpy samplers and `network_paras`, on the MI355X-native trunk.

Run with this directory as the working directory (as the reference is): `from model import
LinearTransformer`, `from config import AgentConfig`.
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
from rlmg_amd.cw_transformer import CWTrunk, Embeddings, PositionalEncoding  # noqa: E402,F401
from rlmg_amd.sampling import nucleus, sample_cw, sampling, softmax_with_temperature, weighted_sampling  # noqa: E402,F401

try:
    from config import AgentConfig
except ImportError:  # imported as a package module rather than from the script directory
    from .config import AgentConfig


def network_paras(model):
    """Trainable parameter count (dqn_policy/model.py:61-65)."""
    return sum(int(np.prod(p.size())) for p in model.parameters() if p.requires_grad)


class LinearTransformer(CWTrunk):
    """dqn_policy/model.py:97-298.  Same constructor, methods and state_dict keys."""

    def __init__(self, n_token, is_training=True):
        super().__init__(n_token, AgentConfig["D_MODEL"], AgentConfig["N_LAYER"], AgentConfig["N_HEAD"],
                         d_inner=2048, dropout=0.1, is_training=is_training)
        self.loss_func = nn.CrossEntropyLoss(reduction="none")
        print("Token_class >>>>>:", self.n_token)
        if not is_training:
            print(" [o] using RNN backend.")
        # "blend with type": declared by the reference (model.py:153), never used in forward
        self.project_concat_type = nn.Linear(self.d_model, self.d_model)
        self._declare_heads()

    def forward_output(self, h, y=None):
        """6 logits tensors (B, T, n_f); the second argument is ignored, as in the reference (:241-249)."""
        return self.split_logits(self.fused_logits(h), h.shape[:-1])

    def forward(self, x, target=None):
        return self.forward_output(self.forward_hidden(x), target)

    def forward_output_sampling(self, h):
        """Generation-time sampling of the next CW token (model.py:259-298) -> np.ndarray[6]."""
        y = [t.float() for t in self.forward_output(h)]
        return sample_cw(y)             # draws in the reference's order: tempo, barbeat, chord, pitch, ...
