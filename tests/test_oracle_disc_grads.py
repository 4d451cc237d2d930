"""CPU: the oracle's discriminator training loss and its autograd gradients vs the fixture recorded from the
reference's own AIRL_model.LongFormer + HF Longformer backward (tests/golden/make_golden.py::airl_grads_small)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

from oracle import discriminator as odisc  # noqa: E402


def test_oracle_disc_loss_and_grads_match_reference_fixture():
    import rlmg_amd  # noqa: F401  (host classes only; nothing is computed with them)
    from rlmg_amd.dqn_policy import AIRL_model
    fx = np.load(os.path.join(HERE, "golden", "airl_grads_small.npz"), allow_pickle=False)
    old = (AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD)
    AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = 128, 2, 2
    try:
        net = fill_params(AIRL_model.LongFormer(fx["n_class"].tolist()), seed=43)
    finally:
        AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = old
    with torch.no_grad():
        net.score_classifier[1].running_mean.copy_(torch.linspace(-0.2, 0.2, 128))
        net.score_classifier[1].running_var.copy_(torch.linspace(0.5, 1.5, 128))
    sd = {k: v.detach().double() for k, v in net.state_dict().items()}
    names = fx["names"].tolist()
    for k in names:
        sd[k].requires_grad_(True)
    x_exp, x_ag, mask = (torch.from_numpy(fx[k]) for k in ("x_exp", "x_agent", "mask"))
    e, a, c = odisc.airl_disc_loss(sd, x_exp, x_ag, mask, 2, 2, 50)
    assert np.allclose([e.item(), a.item(), c.item()], fx["losses"], atol=2e-5)
    (e + (a + c)).backward()
    for k, want_norm in zip(names, fx["norms"]):
        g = sd[k].grad
        assert g is not None, k
        assert abs(g.norm().item() - want_norm) < 2e-5 + 1e-4 * want_norm, k
        want = torch.from_numpy(fx["grad." + k]).double()
        got = g[:8] if g.numel() > 4096 else g
        assert (got - want).abs().max().item() < 2e-5, k
