"""GPU: the product's DQN side -- `DQN.choose_action`, `DQN.update`, `AgentMemory` / `ExpertMemory`
(rlmg_amd/dqn_policy/IRL_dqn_train.py) and `RewardDiscri.update_disc(train=False)` / `calculate_reward`
(rlmg_amd/dqn_policy/AIRL.py) -- against tests/golden/dqn_rl_small.npz, recorded from the REFERENCE's own classes
(/root/reference/dqn_policy/IRL_dqn_train.py:78-345, /root/reference/dqn_policy/AIRL.py:33-91,121-236; see
tests/golden/make_golden.py::dqn_rl_small).  Greedy ids bit-exact, losses / rewards / gradients within 1e-4 (f32), as
BASELINE.json's north_star states it."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401

pytestmark = pytest.mark.gpu
FX = np.load(os.path.join(HERE, "golden", "dqn_rl_small.npz"), allow_pickle=False)
N_CLASS = FX["n_class"].tolist()
TOL = 1e-4


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture
def small_agent(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.dqn_policy import IRL_dqn_train as T, config
    old = dict(config.AgentConfig)
    config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    try:
        torch.manual_seed(0)
        agent = T.DQN(N_CLASS, Pretrain=False)
        fill_params(agent.eval_net, seed=61).eval()
        fill_params(agent.target_net, seed=99).eval()
        yield T, agent
    finally:
        config.AgentConfig.update(old)


def test_choose_action_is_bit_exact(cuda, small_agent):
    T, agent = small_agent
    x = _t(FX["choose.x"]).to(cuda)
    action = agent.choose_action(x, x)
    assert action.dtype == torch.int64 and tuple(action.shape) == (25, 6)
    assert torch.equal(action.cpu(), _t(FX["choose.action"]))


def test_one_update_and_the_schedule_match_the_reference(cuda, small_agent):
    T, agent = small_agent
    st, ns, ex = _t(FX["update.state"]), _t(FX["update.nextstate"]), _t(FX["update.expert_next"])
    ac, rw, dn, mask = _t(FX["update.action"]), _t(FX["update.reward"]), _t(FX["update.done"]), _t(FX["update.mask"])
    agent_tr = {"state": st.to(cuda), "action": ac.to(cuda), "reward": rw, "nextstate": ns.to(cuda), "done": dn.to(cuda)}
    expert_tr = {"state": st.to(cuda), "action": ac.to(cuda), "reward": rw, "nextstate": ex.to(cuda), "done": dn.to(cuda)}
    before = {k: v.detach().clone() for k, v in agent.eval_net.state_dict().items()}
    m, c, t = agent.update(agent_tr, expert_tr, mask.to(cuda), False, 0)
    for got, want in zip((m, c, t), FX["update.losses"]):
        assert abs(got - want) <= TOL * max(1.0, abs(want)), (got, want)
    # target_count 0: the target net was loaded from the eval net BEFORE the step
    for k, v in agent.target_net.state_dict().items():
        assert torch.equal(v, before[k]), k
    ps = dict(agent.eval_net.named_parameters())
    norms = dict(zip(FX["update.gradnames"].tolist(), FX["update.gradnorm"].tolist()))
    for k, want in norms.items():
        g = ps[k].grad
        assert g is not None, k
        assert abs(g.double().norm().item() - want) <= 2e-4 * max(want, 1e-6), (k, g.double().norm().item(), want)
    for key in FX.files:
        if key.startswith("update.grad."):
            k = key[len("update.grad."):]
            g = ps[k].grad
            g = (g[:8] if g.numel() > 4096 else g).cpu()
            w = _t(FX[key])
            assert (g - w).abs().max().item() <= TOL * max(1e-3, w.abs().max().item()), k
    for key in FX.files:
        if key.startswith("update.after."):
            k = key[len("update.after."):]
            v = ps[k].detach()
            v = (v[:8] if v.numel() > 4096 else v).cpu()
            solid = _t(FX["update.grad." + k]).abs() > 1e-5      # Adam's first step: lr g / (|g| + 1e-8)
            if solid.any():
                assert (v - _t(FX[key]))[solid].abs().max().item() <= TOL, k
            # everywhere: one step moves a weight by at most lr = 1e-2 (+ rounding)
            assert (v - _t(FX[key])).abs().max().item() <= 2.1e-2, k
    # updates 2..52: MultiStepLR per update, no target sync until target_count reaches 50 (update 51)
    lrs = [float(agent.optim.param_groups[0]["lr"])]
    for i in range(2, 53):
        if i == 51:
            pre = {k: v.detach().clone() for k, v in agent.eval_net.state_dict().items()}
        agent.update(agent_tr, expert_tr, mask.to(cuda), False, 0)
        lrs.append(float(agent.optim.param_groups[0]["lr"]))
        if i == 50:
            for k, v in agent.target_net.state_dict().items():
                assert torch.equal(v, before[k]), k
        if i == 51:
            for k, v in agent.target_net.state_dict().items():
                assert torch.equal(v, pre[k]), k
    assert np.allclose(np.array(lrs), FX["update.lr_after"], rtol=1e-9, atol=1e-15)
    assert [agent.target_count, agent.cnt_update] == FX["update.counters"].tolist()


def test_ring_buffers_match_the_reference(cuda, small_agent, monkeypatch):
    T, _ = small_agent
    monkeypatch.setattr(T, "BUFFER_SIZE", 8)
    ab, eb = T.AgentMemory(), T.ExpertMemory()
    for i in range(11):
        s_, a_, n_ = (_t(FX["ring.in." + k][i]).to(cuda) for k in ("state", "action", "next"))
        r_, d_ = _t(FX["ring.in.reward"][i]).to(cuda), _t(FX["ring.in.done"][i]).to(cuda)
        ab.store_transition(s_, a_, r_, n_, d_)
        eb.store_transition(s_, a_, r_, n_, d_, _t(FX["ring.in.mstate"][i]).to(cuda), _t(FX["ring.in.mnext"][i]).to(cuda))
    assert [ab.memory_counter, eb.memory_counter] == FX["ring.counter"].tolist()

    def same(t, want, what):
        t = t.cpu().numpy()
        assert str(t.dtype) == str(want.dtype) and t.shape == want.shape, (what, t.dtype, want.dtype, t.shape)
        assert np.array_equal(t, want) if t.dtype.kind == "i" else np.allclose(t, want, rtol=0, atol=1e-7), what

    for i, t in enumerate(ab.get()):
        same(t, FX["ring.agent_get.%d" % i], "agent get %d" % i)
    for i, t in enumerate(eb.get()):
        same(t, FX["ring.expert_get.%d" % i], "expert get %d" % i)
    np.random.seed(4242)
    for i, t in enumerate(ab.sampling(5)):
        same(t, FX["ring.agent_sample.%d" % i], "agent sample %d" % i)
    for i, t in enumerate(eb.sampling(5)):
        same(t, FX["ring.expert_sample.%d" % i], "expert sample %d" % i)


def test_update_disc_scoring_matches_the_reference(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.dqn_policy import AIRL, AIRL_model
    old = (AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD)
    AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = 128, 2, 2
    try:
        rd = AIRL.RewardDiscri(N_CLASS, Pretrain=False)
    finally:
        AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = old
    fill_params(rd.disc_model, seed=41)
    with torch.no_grad():
        rd.disc_model.score_classifier[1].running_mean.copy_(torch.linspace(-0.2, 0.2, 128))
        rd.disc_model.score_classifier[1].running_var.copy_(torch.linspace(0.5, 1.5, 128))
    # the record was taken with dropout off (all_forward forces train(): BatchNorm on batch statistics stays)
    zeroed = 0
    for mod in rd.disc_model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "p_hidden"):
            mod.p_hidden = mod.p_attn = 0.0
            zeroed += 1
    assert zeroed >= 1
    os.makedirs("ckpt")
    torch.save({"epoch": 0, "model_state_dict": rd.disc_model.state_dict()}, rd.IRL_ckpt_path)
    # scramble the live weights: calculate_reward must score with what the checkpoint file holds (AIRL.py:73)
    fill_params(rd.disc_model, seed=7)
    rd.batch_size = int(FX["reward.batch_size"])
    n = FX["reward.agent_states"].shape[0]
    dones = torch.zeros(n, 1).long()
    m_states, m_next = _t(FX["reward.mask_states"]), torch.ones(n, 50)
    a_states, e_states = _t(FX["reward.agent_states"]), _t(FX["reward.expert_states"])
    traj, answer = rd.update_disc((a_states, None, None, a_states, dones),
                                  (e_states, None, None, e_states, dones, m_states, m_next), train=False)
    assert tuple(traj.shape) == (n, 1) and not traj.is_cuda
    assert (traj - _t(FX["reward.traj"])).abs().max().item() <= TOL
    assert (answer - _t(FX["reward.answer"])).abs().max().item() <= TOL
    assert traj[8:].tolist() == [[1.0]] * 3                       # the tail shorter than a batch keeps 1.0
    assert os.path.exists(rd.reward_path)
