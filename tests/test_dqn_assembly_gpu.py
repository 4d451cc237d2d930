"""GPU: the ASSEMBLY of the DQN side -- not its kernels, which have their own tests -- against the literal restatement of
the reference (the DQN class of /root/reference/dqn_policy/IRL_dqn_train.py calls `.cuda()` in its constructor and
imports a module the reference does not contain, so it cannot be run in the build container; oracle/rl_math.py and
oracle/dqn_loop.py restate :240-345 and :436-497):

  * the whole `DQN.update` on a fixed batch, eval mode: MSEloss, CEloss, total loss and named `eval_net` gradients
    against `rl_math.dqn_td_loss` + the oracle's `train_step` + autograd on the oracle's network; the target-network
    sync, and the MultiStepLR stepped once per UPDATE (lr 1e-2 -> 1e-3 after 20 updates -> 1e-4 after 40, :344-345);
  * the loop body of `IRL_dqn_train.main` with a scripted agent: next-state composition, expert windows and masks, the
    overwrite of every stored reward, the two `sampling` calls under one seeded `np.random`, and the arguments `update`
    receives (the CE target is the AGENT's own sampled next state, :486-487), compared with `oracle.dqn_loop.rollout`.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401
from oracle import cw_model, dqn_loop, rl_math  # noqa: E402

pytestmark = pytest.mark.gpu
N_CLASS = [56, 135, 18, 87, 18, 25]


def _small(cfg_dict):
    old = dict(cfg_dict)
    cfg_dict.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    return old


def _batch(B, seed):
    g = torch.Generator().manual_seed(seed)
    st = torch.stack([torch.randint(0, n, (B, 50), generator=g) for n in N_CLASS], -1)
    ns = torch.stack([torch.randint(0, n, (B, 50), generator=g) for n in N_CLASS], -1)
    ac = torch.stack([torch.randint(0, n, (B, 25), generator=g) for n in N_CLASS], -1)
    rw = torch.rand(B, 1, generator=g)
    dn = torch.randint(0, 2, (B, 1), generator=g)
    ex = torch.stack([torch.randint(0, n, (B, 50), generator=g) for n in N_CLASS], -1)
    mask = (torch.rand(B, 50, generator=g) > 0.2).float()
    return st, ns, ac, rw, dn, ex, mask


def test_whole_dqn_update_matches_the_restatement(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.dqn_policy import IRL_dqn_train as T, config
    old = _small(config.AgentConfig)
    try:
        torch.manual_seed(0)
        agent = T.DQN(N_CLASS, Pretrain=False)
        fill_params(agent.eval_net, seed=61)
        fill_params(agent.target_net, seed=99)           # different on purpose: update() must sync it (target_count 0)
        agent.eval_net.eval()
        agent.target_net.eval()
        ref = fill_params(cw_model.CWLinearTransformer(N_CLASS, 128, 2, 2, variant="dqn"), seed=61).eval()
        st, ns, ac, rw, dn, ex, mask = _batch(30, 5)

        # the restatement: IRL_dqn_train.py:285-336 on the oracle's logits (target net == eval net after the sync)
        y = ref.forward_output(ref.forward_hidden(st))
        with torch.no_grad():
            yt = ref.forward_output(ref.forward_hidden(ns))
        mse_ref, _ = rl_math.dqn_td_loss(y, yt, ac, rw, dn, T.GAMMA, T.N_ACTIONS)
        ce = ref.train_step(st, ex, mask)
        ce_ref = sum(ce) / 6
        total_ref = 0.3 * mse_ref + 0.7 * ce_ref
        ref.zero_grad()
        total_ref.backward()

        agent_tr = {"state": st.to(cuda), "action": ac.to(cuda), "reward": rw, "nextstate": ns.to(cuda),
                    "done": dn.to(cuda)}
        expert_tr = {"state": st.to(cuda), "action": ac.to(cuda), "reward": rw, "nextstate": ex.to(cuda),
                     "done": dn.to(cuda)}
        m, c, t = agent.update(agent_tr, expert_tr, mask.to(cuda), False, 0)
        assert abs(m - mse_ref.item()) < 1e-4 * max(1.0, abs(mse_ref.item())), (m, mse_ref.item())
        assert abs(c - ce_ref.item()) < 1e-4 * max(1.0, abs(ce_ref.item())), (c, ce_ref.item())
        assert abs(t - total_ref.item()) < 1e-4 * max(1.0, abs(total_ref.item()))
        # the target network was synchronised with the eval net BEFORE the step (target_count 0 -> load_state_dict)
        assert agent.target_count == 1
        want_tgt = fill_params(cw_model.CWLinearTransformer(N_CLASS, 128, 2, 2, variant="dqn"), seed=61)
        assert torch.allclose(agent.target_net.in_linear.weight.cpu(), want_tgt.in_linear.weight)
        # named gradients of eval_net (they stay in .grad after the optimizer step) against the oracle's autograd
        names = ["in_linear.weight", "word_emb_pitch.lut.weight", "proj_chord.weight", "proj_tempo.bias",
                 "transformer_encoder.layers.0.attention.query_projection.weight",
                 "transformer_encoder.layers.1.linear2.weight", "transformer_encoder.layers.1.norm2.weight",
                 "transformer_encoder.norm.bias"]
        got = dict(agent.eval_net.named_parameters())
        want = dict(ref.named_parameters())
        for n in names:
            gg, ww = got[n].grad.cpu(), want[n].grad
            assert (gg - ww).abs().max().item() <= 1e-4 * max(1e-3, ww.abs().max().item()), n
        # MultiStepLR is stepped per update: 1e-2 for updates 1..20, 1e-3 for 21..40, 1e-4 afterwards
        lrs = [float(agent.optim.param_groups[0]["lr"])]
        for i in range(1, 41):
            agent.update(agent_tr, expert_tr, mask.to(cuda), False, 0)
            lrs.append(float(agent.optim.param_groups[0]["lr"]))
        # lrs[k] = learning rate after k + 1 updates
        assert abs(lrs[0] - 1e-2) < 1e-12 and abs(lrs[18] - 1e-2) < 1e-12
        assert abs(lrs[19] - 1e-3) < 1e-12 and abs(lrs[38] - 1e-3) < 1e-12
        assert abs(lrs[39] - 1e-4) < 1e-12 and abs(lrs[40] - 1e-4) < 1e-12
        assert agent.target_count == 41 and agent.cnt_update == 41
    finally:
        config.AgentConfig.update(old)


class _ScriptedAgent(object):
    """choose_action: a deterministic function of the state; update: records what it is given."""
    calls = None

    def __init__(self, n_class, Pretrain=False):
        self.n = torch.tensor(list(n_class))
        _ScriptedAgent.calls = []

    def choose_action(self, x, target=None):
        dev = x.device
        return ((x.cpu()[0, 25:50] * 3 + target.cpu()[0, :25] + 1) % self.n).to(dev)

    def update(self, agent_transition, expert_transition, mask_next_states, update_flag, epoch):
        rec = {k: v.detach().cpu().clone() for k, v in agent_transition.items()}
        rec.update({"e_" + k: v.detach().cpu().clone() for k, v in expert_transition.items()})
        rec["mask"] = mask_next_states.detach().cpu().clone()
        rec["flag"], rec["epoch"] = update_flag, epoch
        _ScriptedAgent.calls.append(rec)
        return 0.0, 0.0, 0.0


def _scripted_rewards(agent_traj, expert_traj, train=False):
    states, nxt = agent_traj[0].cpu().double(), agent_traj[3].cpu().double()
    r = ((states.sum((1, 2)) * 7 + nxt.sum((1, 2))) % 97) / 97.0
    return r.float().unsqueeze(1), torch.ones_like(r).float().unsqueeze(1)


def test_dqn_main_loop_composition_matches_the_restated_loop(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.dqn_policy import IRL_dqn_train as T
    songs, Tlen, BUF, NS = 3, 2000, 60, 2
    g = torch.Generator().manual_seed(3)
    disk = [56, 135, 18, 3, 87, 18, 25]
    x = torch.stack([torch.randint(0, n, (songs, Tlen), generator=g) for n in disk], -1).numpy()
    y = torch.stack([torch.randint(0, n, (songs, Tlen), generator=g) for n in disk], -1).numpy()
    mask = (torch.rand(songs, Tlen, generator=g) > 0.3).float().numpy()
    keys = ["tempo", "chord", "bar-beat", "type", "pitch", "duration", "velocity"]
    e2w = {k: {"%s_%d" % (k, i): i for i in range(n)} for k, n in zip(keys, disk)}
    monkeypatch.setattr(T.cwdata, "load_dqn", lambda *a, **k: ((e2w, None), {"x": x, "y": y, "mask": mask}))
    monkeypatch.setattr(T, "DQN", _ScriptedAgent)

    class _Rewarder(object):
        def __init__(self, n_class, Pretrain=False):
            pass

        def update_disc(self, agent_traj, expert_traj, train=False):
            r, a = _scripted_rewards(agent_traj, expert_traj, train)
            return r.to(cuda), a.to(cuda)

    monkeypatch.setattr(T, "RewardDiscri", _Rewarder)
    made = []
    real_agent, real_expert = T.AgentMemory, T.ExpertMemory

    class _Agent(real_agent):
        def __init__(self):
            super().__init__()
            made.append(self)

    class _Expert(real_expert):
        def __init__(self):
            super().__init__()
            made.append(self)

    monkeypatch.setattr(T, "AgentMemory", _Agent)
    monkeypatch.setattr(T, "ExpertMemory", _Expert)
    monkeypatch.setattr(T, "NUM_SONGS", NS)
    monkeypatch.setattr(T, "BUFFER_SIZE", BUF)
    np.random.seed(4242)
    T.main()
    ab, eb = made[0], made[1]
    got_calls = _ScriptedAgent.calls

    # the restated loop on the same data, the same scripted agent / rewarder, the same np.random stream
    tx, ty = torch.from_numpy(x), torch.from_numpy(y)
    tx = torch.cat((tx[:, :, :3], tx[:, :, 4:]), dim=-1)[:, :T.SEQ_LEN].long()          # :427-433
    ty = torch.cat((ty[:, :, :3], ty[:, :, 4:]), dim=-1)[:, :T.SEQ_LEN * 2].long()
    ref_agent = _ScriptedAgent([56, 135, 18, 87, 18, 25])
    np.random.seed(4242)
    rab, reb, _ = dqn_loop.rollout(tx, ty, torch.from_numpy(mask), ref_agent.choose_action, _scripted_rewards,
                                   ref_agent.update, NS, BUF, batch_size=T.batch_size)
    want_calls = _ScriptedAgent.calls
    assert ab.memory_counter == rab.memory_counter == NS * 50 and eb.memory_counter == reb.memory_counter
    for name in ("states_agent", "actions_agent", "next_states_agent", "dones_agent"):
        assert np.array_equal(getattr(ab, name).cpu().numpy(), getattr(rab, name).astype(np.int64)), name
    assert np.allclose(ab.rewards_agent.cpu().numpy(), rab.rewards_agent, atol=1e-6)     # every slot overwritten
    for name in ("states_exp", "actions_exp", "next_states_exp", "dones_exp"):
        assert np.array_equal(getattr(eb, name).cpu().numpy(), getattr(reb, name).astype(np.int64)), name
    assert np.allclose(eb.rewards_exp.cpu().numpy(), reb.rewards_exp)
    assert torch.equal(eb.mask_state.cpu(), reb.mask_state) and torch.equal(eb.mask_next_state.cpu(), reb.mask_next_state)
    # next = first 25 tokens of the state + the 25 action tokens, for every stored transition
    assert torch.equal(ab.next_states_agent[:, :25], ab.states_agent[:, :25])
    assert torch.equal(ab.next_states_agent[:, 25:], ab.actions_agent)
    # updates: one per env step once the counter exceeds the buffer size, same arguments
    assert len(got_calls) == len(want_calls) == NS * 50 - BUF
    for a, b in zip(got_calls, want_calls):
        for k in ("state", "action", "nextstate", "done", "e_state", "e_action", "e_nextstate", "e_done"):
            assert torch.equal(a[k].long(), b[k].long()), k
        assert torch.allclose(a["reward"].float(), b["reward"].float(), atol=1e-6)
        assert torch.allclose(a["e_reward"].float(), b["e_reward"].float(), atol=1e-6)
        assert torch.equal(a["mask"].float(), b["mask"].float())
        assert a["flag"] is True and a["epoch"] == b["epoch"]
        # the CE target handed to update() is the agent's own sampled next state (:486-487)
        assert torch.equal(a["e_nextstate"], a["nextstate"])
