"""MI355X-native hot path of the compound-word Linear-Transformer + AIRL / PPO / DQN training
stack (drop-in for daniel05155/Reinforcement-Learning-in-Music-Generation's model.py /
AIRL_model.py / ppo_train.py / IRL_dqn_train.py surfaces).

Compute goes through libcwlt.so (hand-written gfx950 HIP kernels behind the C-ABI in
include/cwlt.h).  There is no CPU fallback: if the library is missing, or a tensor is not on a GPU,
the ops raise.
"""
from . import _lib  # noqa: F401  (does not load the .so until first use)

__all__ = ["_lib"]
