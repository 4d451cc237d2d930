"""CPU: the oracle's PPO reward model (oracle/discriminator.py::ppo_reward_forward) and its autograd gradients vs
the fixture recorded from the reference's own ppo_policy/model.py::LongFormer.token_forward + HF Longformer backward
(tests/golden/make_golden.py::ppo_reward_grads_small)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

from oracle import discriminator as odisc  # noqa: E402


def test_oracle_reward_model_grads_match_reference_fixture():
    import rlmg_amd  # noqa: F401  (host class only, as a parameter container)
    from rlmg_amd.ppo_policy import config as pcfg, model as pmodel
    fx = np.load(os.path.join(HERE, "golden", "ppo_reward_grads_small.npz"), allow_pickle=False)
    old = dict(pcfg.DiscriConfig)
    pcfg.DiscriConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    try:
        net = fill_params(pmodel.LongFormer(fx["n_token"].tolist()), seed=33)
    finally:
        pcfg.DiscriConfig.update(old)
    sd = {k: v.detach().double() for k, v in net.state_dict().items()}
    names = fx["names"].tolist()
    for k in names:
        sd[k].requires_grad_(True)
    score = odisc.ppo_reward_forward(sd, torch.from_numpy(fx["x"]), torch.from_numpy(fx["mask"]), 2, 2, 128)
    assert (score.detach() - torch.from_numpy(fx["score"]).double()).abs().max().item() < 2e-6
    (score * torch.from_numpy(fx["w"]).double()).sum().backward()
    for k, want_norm in zip(names, fx["norms"]):
        g = sd[k].grad
        if g is None:           # HF embeds its window PADDING through word_embeddings: a defined, all-zero gradient
            assert want_norm == 0.0, k
            continue
        assert abs(g.norm().item() - want_norm) < 1e-6 + 1e-4 * want_norm, k
        want = torch.from_numpy(fx["grad." + k]).double()
        got = g[:8] if g.numel() > 4096 else g
        assert (got - want).abs().max().item() < 1e-6 + 1e-4 * want.abs().max().item(), k
