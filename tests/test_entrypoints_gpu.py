"""GPU: the drop-in RL classes and entry-point loops run end to end on synthetic data (short runs), and the
class-level methods agree with the literal restatement of the reference."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401
from oracle import cw_model, rl_math  # noqa: E402

pytestmark = pytest.mark.gpu


def _small(cfg_dict):
    old = dict(cfg_dict)
    cfg_dict.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    return old


def test_dqn_class_choose_action_and_update(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.dqn_policy import IRL_dqn_train as T, config
    old = _small(config.AgentConfig)
    try:
        n_class = [56, 135, 18, 87, 18, 25]
        torch.manual_seed(0)
        agent = T.DQN(n_class, Pretrain=False)
        fill_params(agent.eval_net, seed=61)
        agent.eval_net.eval()                    # dropout off for the comparison
        ref = fill_params(cw_model.CWLinearTransformer(n_class, 128, 2, 2, variant="dqn"), seed=61).eval()
        g = torch.Generator().manual_seed(1)
        x = torch.stack([torch.randint(0, n, (1, 50), generator=g) for n in n_class], -1)
        with torch.no_grad():
            want = rl_math.dqn_choose_action(ref.forward_output(ref.forward_hidden(x)), 25)
        got = agent.choose_action(x.to(cuda), None)
        assert torch.equal(got.cpu(), want)
        # one update on a random batch: finite losses, parameters move, lr schedule steps
        agent.eval_net.train()
        B = 30
        st = torch.stack([torch.randint(0, n, (B, 50), generator=g) for n in n_class], -1).to(cuda)
        ns = torch.stack([torch.randint(0, n, (B, 50), generator=g) for n in n_class], -1).to(cuda)
        ac = torch.stack([torch.randint(0, n, (B, 25), generator=g) for n in n_class], -1).to(cuda)
        tr = {"state": st, "action": ac, "reward": torch.rand(B, 1), "nextstate": ns, "done": torch.zeros(B, 1)}
        before = agent.eval_net.in_linear.weight.detach().clone()
        m, c, t = agent.update(tr, dict(tr), torch.ones(B, 50, device=cuda), False, 0)
        assert all(map(lambda v: v == v and abs(v) < 1e6, (m, c, t)))
        assert not torch.equal(before, agent.eval_net.in_linear.weight.detach())
        assert abs(t - (0.3 * m + 0.7 * c)) < 1e-4 * max(1.0, abs(t))
    finally:
        config.AgentConfig.update(old)


def test_ppo_class_methods(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.ppo_policy import config, ppo_train as P
    old_a, old_d = _small(config.ActorConfig), _small(config.DiscriConfig)
    try:
        n_token = [49, 19, 19, 89, 67, 25]
        agent = P.PPO(n_token, Pretrain=False)
        fill_params(agent.actor_net, seed=71)
        agent.actor_net.eval()
        ref = fill_params(cw_model.CWLinearTransformer(n_token, 128, 2, 2, variant="actor"), seed=71).eval()
        g = torch.Generator().manual_seed(2)
        x = torch.stack([torch.randint(0, n, (1, 50), generator=g) for n in n_token], -1)
        with torch.no_grad():
            wa, wl = rl_math.ppo_choose_action(ref.forward_output(ref.forward_hidden(x)), 25)
        ga, gl = agent.choose_action(x.to(cuda))
        assert torch.equal(ga.cpu(), wa) and (gl.cpu() - wl).abs().max().item() < 1e-4
        xs = torch.stack([torch.randint(0, n, (5, 50), generator=g) for n in n_token], -1)
        with torch.no_grad():
            sa, sl = rl_math.ppo_select_update(ref.forward_output(ref.forward_hidden(xs)), 25)
        a2, l2, v2 = agent.select_udpate(xs.to(cuda))
        assert torch.equal(a2.cpu(), sa) and (l2.detach().cpu() - sl).abs().max().item() < 1e-4
        assert v2.shape == (5, 1)
        r = torch.rand(30, 1, generator=g)
        ret = agent.calculate_returns(r.to(cuda), 0.99)
        want = rl_math.ppo_returns([t for t in r], 0.99)
        assert (ret.cpu() - want).abs().max().item() < 1e-4
    finally:
        config.ActorConfig.update(old_a)
        config.DiscriConfig.update(old_d)


def test_ppo_main_loop_short_run(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.ppo_policy import config, ppo_train as P
    old_a, old_d = _small(config.ActorConfig), _small(config.DiscriConfig)
    monkeypatch.setattr(P, "NUM_SONGS", 1)
    monkeypatch.setattr(P, "PPO_STEPS", 2)
    monkeypatch.setenv("CWLT_NO_PRETRAIN", "1")
    try:
        P.main()
        assert os.path.exists("ckpt/ppo_best.pt")
    finally:
        config.ActorConfig.update(old_a)
        config.DiscriConfig.update(old_d)


def test_dqn_main_loop_short_run(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.dqn_policy import AIRL_model, IRL_dqn_train as T, config
    old = _small(config.AgentConfig)
    oldd = (AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD)
    AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = 128, 2, 2
    monkeypatch.setattr(T, "NUM_SONGS", 3)
    monkeypatch.setattr(T, "BUFFER_SIZE", 100)       # buffer fills after 2 songs -> updates start in song 3
    monkeypatch.setenv("CWLT_NO_PRETRAIN", "1")
    try:
        T.main()
        assert os.path.exists("exp/IRL_reward.pickle")
    finally:
        config.AgentConfig.update(old)
        AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = oldd


@pytest.mark.parametrize("loop", ["dqn", "ppo"])
def test_rl_main_loops_short_run_bf16(cuda, tmp_path, monkeypatch, loop):
    """The drop-in RL loops in the throughput mode (CWLT_COMPUTE_DTYPE=bf16): bf16 trunk kernels, MFMA band
    attention, grouped buffer scoring and the graphed rollout step run end to end with finite results."""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("CWLT_COMPUTE_DTYPE", "bf16")
    monkeypatch.setenv("CWLT_NO_PRETRAIN", "1")
    import pickle
    if loop == "dqn":
        from rlmg_amd.dqn_policy import AIRL_model, IRL_dqn_train as T, config
        old = _small(config.AgentConfig)
        oldd = (AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD)
        AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = 128, 2, 2
        monkeypatch.setattr(T, "NUM_SONGS", 3)
        monkeypatch.setattr(T, "BUFFER_SIZE", 100)
        try:
            T.main()
            with open("exp/IRL_reward.pickle", "rb") as f:
                rew = pickle.load(f)
            assert torch.isfinite(rew["Agent"]).all() and torch.isfinite(rew["Expert"]).all()
        finally:
            config.AgentConfig.update(old)
            AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = oldd
    else:
        from rlmg_amd.ppo_policy import config, ppo_train as P
        old_a, old_d = _small(config.ActorConfig), _small(config.DiscriConfig)
        monkeypatch.setattr(P, "NUM_SONGS", 1)
        monkeypatch.setattr(P, "PPO_STEPS", 2)
        try:
            P.main()
            with open("./ckpt/policy_loss.pickle", "rb") as f:
                pl = pickle.load(f)["policy_loss"]
            assert len(pl) == 1 and pl[0] == pl[0] and abs(pl[0]) < 1e6
        finally:
            config.ActorConfig.update(old_a)
            config.DiscriConfig.update(old_d)


def test_agent_pretrain_train_short_run(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.dqn_policy import agent_pretrain as A, config
    from rlmg_amd import data as cwdata
    old = _small(config.AgentConfig)
    real_load = cwdata.load_dqn
    monkeypatch.setattr(cwdata, "load_dqn", lambda a, b, **kw: real_load(a, b, n_seq=8, T=256))
    try:
        loss = A.train(n_epoch=2, log=lambda *a: None)
        assert loss == loss and 0 < loss < 10
        assert any(f.startswith("trainloss_") for f in os.listdir("ckpt"))
        # exp/log.txt in the reference's Saver format: 2 epochs x 2 batches of 4 + epoch lines
        lines = open(os.path.join("exp", "log.txt")).read().splitlines()
        assert sum(l.startswith("batch loss") for l in lines) == 4
        assert sum(l.startswith("epoch loss") for l in lines) == 2
        key, val, step, _ = [l for l in lines if l.startswith("epoch loss")][-1].split(" | ")
        assert abs(float(val) - loss) < 1e-9 and int(step) == 4
    finally:
        config.AgentConfig.update(old)


def test_ppo_update_rollouts_grouping_is_exact(cuda, tmp_path, monkeypatch):
    """Stacking `group` rollouts per network pass gives the gradients of one-rollout-at-a-time accumulation, and
    those equal the literal per-rollout recipe (select_udpate + policy loss + CE + critic MSE) summed by hand."""
    monkeypatch.chdir(tmp_path)
    from rlmg_amd import rl_ops
    from rlmg_amd.ppo_policy import config, ppo_train as P
    old_a, old_d = _small(config.ActorConfig), _small(config.DiscriConfig)
    try:
        n_token = [49, 19, 19, 89, 67, 25]
        E, R, W, NA = 4, 3, 50, 25
        g = torch.Generator().manual_seed(5)
        states = torch.stack([torch.randint(0, n, (E, R, W), generator=g) for n in n_token], -1).to(cuda)
        expert = torch.stack([torch.randint(0, n, (R, E + W + 3), generator=g) for n in n_token], -1).to(cuda)
        mask = torch.ones(R, E + W + 3)
        mask[1, 40:] = 0                                   # ragged: rollout 1 has a shorter valid span
        mask = mask.to(cuda)
        old_int = (-3 * torch.rand(E, R, NA, 6, generator=g)).long().to(cuda)
        advs = [torch.randn(E, generator=g).to(cuda) for _ in range(R)]
        rets = [torch.randn(E, generator=g).to(cuda) for _ in range(R)]

        def fresh():
            torch.manual_seed(0)
            agent = P.PPO(n_token, Pretrain=False)
            fill_params(agent.actor_net, seed=71)
            fill_params(agent.critic_net, seed=72)
            agent.actor_net.eval()
            agent.critic_net.eval()
            for opt in (agent.actor_optim, agent.critic_optim):      # keep the gradients, skip the step
                monkeypatch.setattr(opt, "step", lambda *a, **k: None)
            return agent

        def grads(agent):
            return [p.grad.detach().clone() for net in (agent.actor_net, agent.critic_net) for p in net.parameters()]

        a1 = fresh()
        a1.update_rollouts(states, old_int, advs, rets, expert, mask, group=1)
        g1 = grads(a1)
        a3 = fresh()
        a3.update_rollouts(states, old_int, advs, rets, expert, mask, group=2)     # groups of 2 + 1
        g3 = grads(a3)
        ref = fresh()
        ref.actor_sync.zero_grad()
        ref.critic_sync.zero_grad()
        for r in range(R):
            st = states[:, r]
            _, new_logp, value_pred = ref.select_udpate(st)
            pl = rl_ops.ppo_policy_loss(new_logp, old_int[:, r], advs[r], P.PPO_CLIP)
            ce = ref.actor_net.train_step(st, expert[r, :E + W].unfold(0, W, 1)[:E].permute(0, 2, 1),
                                          mask[r, :E + W].unfold(0, W, 1)[:E])
            ((pl + (ce[0] + ce[1] + ce[2] + ce[3] + ce[4] + ce[5]) / 6) / R).backward()
            (torch.nn.functional.mse_loss(rets[r], value_pred).sum() / R).backward()
        ref.actor_sync.finish()
        ref.critic_sync.finish()
        g0 = grads(ref)
        scale = max(t.abs().max().item() for t in g0)
        assert scale > 1e-4
        for x, y, z in zip(g0, g1, g3):
            assert (x - y).abs().max().item() < 2e-5 * max(1.0, scale)
            assert (x - z).abs().max().item() < 2e-5 * max(1.0, scale)
    finally:
        config.ActorConfig.update(old_a)
        config.DiscriConfig.update(old_d)
