"""CPU: oracle/rl_math.py (and the oracle networks under it) against tests/golden/ppo_rl_small.npz, which was
recorded from the REFERENCE's own `PPO`, `AgentMemory` and `ExpertMemory` (ppo_policy/ppo_train.py:69-417, imported
unmodified by tests/golden/make_golden.py::ppo_rl_small and driven as its main loop :460-506 drives them).
After this test the RL restatement no longer rests on a reading of the reference: A11 (PPO side), A14, A17, A18 of
SURVEY section 8 are pinned by reference-generated vectors."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

from oracle import cw_model, discriminator as odisc, rl_math  # noqa: E402

FX = np.load(os.path.join(HERE, "golden", "ppo_rl_small.npz"), allow_pickle=False)
N_TOKEN = FX["n_token"].tolist()
E, W, NA = 30, 50, 25


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _nets():
    actor = fill_params(cw_model.CWLinearTransformer(N_TOKEN, 128, 2, 2, variant="actor"), seed=21).eval()
    critic = fill_params(cw_model.CWLinearTransformer(N_TOKEN, 128, 2, 2, variant="critic"), seed=22).eval()
    return actor, critic


def _reward_sd():
    import rlmg_amd  # noqa: F401  -- the product module only as a CPU parameter container (state-dict names)
    from rlmg_amd.ppo_policy import config as pcfg, model as pmodel
    old = dict(pcfg.DiscriConfig)
    pcfg.DiscriConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    try:
        net = fill_params(pmodel.LongFormer(N_TOKEN), seed=31)
    finally:
        pcfg.DiscriConfig.update(old)
    return {k: v.detach() for k, v in net.state_dict().items()}


def test_rollout_of_one_song_matches_reference():
    """30 env steps: rl_math.ppo_choose_action on the oracle actor's logits, next = cat(state[:25], action)
    (ppo_train.py:483), oracle critic value and reward model of the next state -- all against the recorded loop."""
    actor, critic = _nets()
    sd = _reward_sd()
    state = _t(FX["state0"])
    mask = _t(FX["train_mask"])
    with torch.no_grad():
        for num in range(E):
            ys = actor.forward_output(actor.forward_hidden(state.unsqueeze(0)))
            action, logp = rl_math.ppo_choose_action(ys, NA)
            assert torch.equal(action, _t(FX["actions"][num])), num
            assert (logp - _t(FX["logps"][num])).abs().max().item() < 1e-5, num
            state = torch.cat((state[:NA], action), dim=0)
            v = critic.value_produce(state.unsqueeze(0))
            assert abs(v.item() - float(FX["values"][num])) < 1e-5
            r = odisc.ppo_reward_forward(sd, state.unsqueeze(0), mask[num:num + W].unsqueeze(0).long(), 2, 2, 128)
            assert abs(r.item() - float(FX["rewards"][num])) < 1e-5
            # what the reference's buffers hold for this step (stored state == next_state: ppo_train.py:486,494)
            assert torch.equal(state, _t(FX["agent_get.states"][num]))
            assert torch.equal(state, _t(FX["agent_get.next_states"][num]))
            assert torch.equal(_t(FX["expert_x"][num:num + W]), _t(FX["expert_get.states"][num]))
            assert torch.equal(_t(FX["expert_x"][num + 50:num + 50 + W]), _t(FX["expert_get.next_states"][num]))
            assert torch.equal(mask[num:num + W].long(), _t(FX["expert_get.mask_state"][num]))
            assert torch.equal(mask[num + 1:num + 1 + W].long(), _t(FX["expert_get.mask_next_state"][num]))
    # the buffer returns stored log-probs through .long(): truncation toward zero (ppo_train.py:135)
    assert np.array_equal(FX["agent_get.log_actions"], np.trunc(FX["logps"]).astype(np.int64))
    assert np.allclose(FX["agent_get.values"][:, 0], FX["values"]) and np.allclose(FX["agent_get.rewards"][:, 0],
                                                                                    FX["rewards"])


def test_returns_advantages_match_reference():
    rewards = _t(FX["agent_get.rewards"])
    raw = rl_math.ppo_returns([r for r in rewards], 0.99, normalize=False)
    assert (raw - _t(FX["returns_raw"])).abs().max().item() < 1e-6
    ret = rl_math.ppo_returns([r for r in rewards], 0.99)
    assert (ret - _t(FX["returns"])).abs().max().item() < 1e-5
    adv = rl_math.ppo_advantages(ret, _t(FX["agent_get.values"]))
    assert (adv - _t(FX["advantages"])).abs().max().item() < 1e-5


def test_select_update_and_update_policy_step_match_reference():
    actor, critic = _nets()
    states = _t(FX["agent_get.states"])
    with torch.no_grad():
        ys = actor.forward_output(actor.forward_hidden(states))
        sa, sl = rl_math.ppo_select_update(ys, NA)
        sv = critic.value_produce(states)
    assert torch.equal(sa, _t(FX["select_action"]))
    assert (sl - _t(FX["select_logp"])).abs().max().item() < 1e-5
    assert (sv - _t(FX["select_value"])).abs().max().item() < 1e-5
    # one inner step of update_policy (ppo_train.py:380-411) restated with rl_math on the oracle nets
    ys = actor.forward_output(actor.forward_hidden(states))
    _, new_logp = rl_math.ppo_select_update(ys, NA)
    value_pred = critic.value_produce(states)
    adv, ret = _t(FX["advantages"]), _t(FX["returns"])
    policy_loss = rl_math.ppo_policy_loss(new_logp, _t(FX["agent_get.log_actions"]), adv, 0.2)
    ce = actor.train_step(states, _t(FX["expert_get.states"]), _t(FX["expert_get.mask_state"]))
    actor_loss = policy_loss + sum(ce) / 6
    value_loss = torch.nn.functional.mse_loss(ret, value_pred).sum()
    assert abs(actor_loss.item() - float(FX["update_actor_loss"])) < 1e-5
    assert abs(value_loss.item() - float(FX["update_value_loss"])) < 1e-5
    actor_loss.backward()
    value_loss.backward()
    for who, net in (("actor", actor), ("critic", critic)):
        ps = dict(net.named_parameters())
        for key in FX.files:
            if key.startswith("grad.%s." % who):
                g = ps[key[len("grad.%s." % who):]].grad
                g = g[:8] if g.numel() > 4096 else g
                want = _t(FX[key])
                assert (g - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item()), key
        names = FX["gradnames." + who].tolist()
        norms = FX["gradnorm." + who]
        for n_, want in zip(names, norms):
            if n_ not in ps:                      # reference-only parameters: none expected
                raise AssertionError(n_)
            got = -1.0 if ps[n_].grad is None else ps[n_].grad.double().norm().item()
            assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (who, n_, got, want)
