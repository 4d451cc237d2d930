"""CPU: the restated DQN + AIRL rollout loop (oracle/dqn_loop.py: `rollout`, `RefDQN`, the two ring buffers;
oracle/discriminator.py::calculate_reward) against tests/golden/dqn_loop_small.npz, which was recorded by running the
REFERENCE's own `__main__` block (/root/reference/dqn_policy/IRL_dqn_train.py:386-497) on the CPU for two songs with
five module constants replaced in memory (tests/golden/make_golden.py::dqn_loop_small).  Pins A19 (DQN loop) of SURVEY
section 8 -- next-state composition, expert windows and masks, the overwrite of every stored reward from the 61st step on,
both `sampling` calls under one np.random stream, the CE target being the AGENT's own sampled next state -- and, through
40 consecutive updates, the whole of `DQN.update` including its optimizer and schedule."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

from oracle import cw_model, discriminator as odisc, dqn_loop  # noqa: E402

FX = np.load(os.path.join(HERE, "golden", "dqn_loop_small.npz"), allow_pickle=False)
N_CLASS = FX["n_class"].tolist()


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _disc_sd():
    import rlmg_amd  # noqa: F401  -- the product module only as a CPU parameter container (state-dict names)
    from rlmg_amd.dqn_policy import AIRL_model
    old = (AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD)
    AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = 128, 2, 2
    try:
        net = fill_params(AIRL_model.LongFormer(N_CLASS), seed=41)
    finally:
        AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = old
    return {k: v.detach() for k, v in net.state_dict().items()}


def test_restated_loop_reproduces_the_reference_run():
    torch.manual_seed(0)
    eval_net = fill_params(cw_model.CWLinearTransformer(N_CLASS, 128, 2, 2, variant="dqn"), seed=61).eval()
    target_net = cw_model.CWLinearTransformer(N_CLASS, 128, 2, 2, variant="dqn").eval()
    agent = dqn_loop.RefDQN(eval_net, target_net)
    sd = _disc_sd()
    bs, BUF = int(FX["score_batch"]), int(FX["buffer_size"])
    rewards, given = [], []

    def update_disc(agent_traj, expert_traj, train=False):
        assert train is False
        with torch.no_grad():
            traj = odisc.calculate_reward(sd, agent_traj[0], expert_traj[5], bs, 2, 2, 50)
            answer = odisc.calculate_reward(sd, expert_traj[0], expert_traj[5], bs, 2, 2, 50)
        rewards.append((traj, answer))
        return traj, answer

    def update(agent_transition, expert_transition, mask_next_states, update_flag, epoch):
        given.append({k: v.clone() for k, v in agent_transition.items()} |
                     {"e_" + k: v.clone() for k, v in expert_transition.items()} |
                     {"mask": mask_next_states.clone(), "flag": update_flag, "epoch": epoch,
                      "lr": agent.optim.param_groups[0]["lr"]})
        return agent.update(agent_transition, expert_transition, mask_next_states, update_flag, epoch)

    actions = []

    def choose(x, target):
        a = agent.choose_action(x, target)
        actions.append(a.clone())
        return a

    x, y = _t(FX["x"].astype(np.int64)), _t(FX["y"].astype(np.int64))
    x = torch.cat((x[:, :, :3], x[:, :, 4:]), dim=-1)[:, :1000]            # :427-433: the `type` column is dropped
    y = torch.cat((y[:, :, :3], y[:, :, 4:]), dim=-1)[:, :2000]
    np.random.seed(int(FX["np_seed"]))
    ab, eb, gene = dqn_loop.rollout(x, y, _t(FX["mask"]), choose, update_disc, update, 2, BUF, batch_size=30)

    want_actions = FX["actions"].astype(np.int64)
    got_actions = torch.stack(actions).numpy()
    # no weight has moved before the first update (step 61): bit-exact; afterwards 40 Adam steps at lr 1e-2 separate two
    # f32 implementations by rounding only -- the greedy ids still agree (a flipped near-tie would show here)
    assert np.array_equal(got_actions[:BUF + 1], want_actions[:BUF + 1])
    assert (got_actions == want_actions).mean() > 0.995, (got_actions != want_actions).sum()
    n_upd = len(given)
    assert n_upd == 40 == len(FX["update.losses"])
    same = np.array_equal(got_actions, want_actions)
    for i in range(n_upd):
        tol = 2e-4 if i else 2e-6
        if same or i == 0:
            assert np.allclose(rewards[i][0][:, 0].numpy(), FX["traj_reward"][i], atol=1e-5), i
            assert np.allclose(rewards[i][1][:, 0].numpy(), FX["answer_reward"][i], atol=1e-5), i
            for k in ("state", "action", "nextstate", "e_nextstate", "mask"):
                assert abs(float(given[i][k].double().sum()) - FX["update.sum." + k][i]) < 1e-6, (i, k)
            assert np.allclose(np.array(agent.losses[i]), FX["update.losses"][i], rtol=tol, atol=tol), i
        assert abs(given[i]["lr"] - FX["update.lr_before"][i]) < 1e-12 and given[i]["epoch"] == FX["update.epoch"][i]
        assert given[i]["flag"] is True
    for i in range(3):
        for k in ("state", "action", "nextstate", "e_nextstate"):
            assert np.array_equal(given[i][k].numpy(), FX["update3." + k][i].astype(np.int64)), (i, k)
        assert np.array_equal(given[i]["mask"].numpy(), FX["update3.mask"][i])
        # the CE target handed to update() is the AGENT's sampled next state (:486-487), not the expert's
        assert torch.equal(given[i]["e_nextstate"], given[i]["nextstate"])
    assert np.array_equal(FX["update.reward"][0], given[0]["reward"].numpy())
    assert np.array_equal(FX["update.e_done"][0], given[0]["e_done"].numpy())
    assert [ab.memory_counter, eb.memory_counter] == FX["final.counters"].tolist()
    if same:
        assert np.array_equal(ab.states_agent.astype(np.int64), FX["final.agent_states"].astype(np.int64))
        assert np.array_equal(ab.actions_agent.astype(np.int64), FX["final.agent_actions"].astype(np.int64))
        assert np.array_equal(ab.next_states_agent.astype(np.int64), FX["final.agent_next"].astype(np.int64))
        assert np.allclose(ab.rewards_agent, FX["final.agent_rewards"], atol=1e-5)
        assert np.allclose(np.array(gene), FX["gene_reward"], atol=1e-6)
    assert np.array_equal(eb.states_exp.astype(np.int64), FX["final.expert_states"].astype(np.int64))
    assert np.array_equal(eb.next_states_exp.astype(np.int64), FX["final.expert_next"].astype(np.int64))
    assert np.allclose(eb.rewards_exp, FX["final.expert_rewards"])
    assert np.array_equal(eb.mask_state.numpy(), FX["final.mask_state"])
    assert np.array_equal(eb.mask_next_state.numpy(), FX["final.mask_next_state"])
    # every stored reward was overwritten by the last scoring call; the 12-window tail of the 60 keeps 1.0
    assert np.allclose(FX["final.agent_rewards"][:, 0], FX["traj_reward"][-1]) and (FX["traj_reward"][:, 48:] == 1.0).all()
