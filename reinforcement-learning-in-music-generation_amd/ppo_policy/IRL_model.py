"""Drop-in surface for /root/reference/ppo_policy/IRL_model.py: the older copy of the PPO reward
`LongFormer` (window 50).  The reference file is dead code (its `forward` uses an undefined
`self.score_classifier`, IRL_model.py:118); `token_forward` is the usable part and is what is kept."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import rlmg_amd  # noqa: E402,F401
from rlmg_amd.cw_transformer import Embeddings  # noqa: E402,F401

try:
    from model import LongFormer as _RewardLongFormer
    from config import DiscriConfig
except ImportError:
    from .model import LongFormer as _RewardLongFormer
    from .config import DiscriConfig

N_STATES = 50


class LongFormer(_RewardLongFormer):
    def __init__(self, n_token):
        super().__init__(n_token)
        self._lf_args["attention_window"] = N_STATES       # IRL_model.py: attention_window = N_STATES
        self._build_longformer()
