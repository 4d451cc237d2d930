"""GPU: the PPO product path on BASELINE.json configs[2]'s shapes (window 1024, N_ACTIONS 512, EPISODES 30) and on
configs[4]'s window (4096, N_ACTIONS 2048) at REPO DIMS (512 / 12 / 8), parity (fp32) mode, against the literal
restatement of the reference's arithmetic (oracle/rl_math.py fed with the CPU oracle's logits).  The reference's
loop is ppo_policy/ppo_train.py:251-417; the many-rollout generalisation is bench_ppo.py's.

Greedy ids are compared bit for bit wherever the oracle's own top-2 logit margin exceeds MARGIN (a GPU / CPU f32
difference of ~1e-5 in a logit can only flip an argmax inside that margin); rows inside it are counted and must
be rare."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401
from rlmg_amd import ops, rl_ops  # noqa: E402
from oracle import cw_model, rl_math  # noqa: E402

pytestmark = pytest.mark.gpu
N_TOKEN = [49, 19, 19, 89, 67, 25]
MARGIN = 1e-3
TOL = 1e-4


def _tokens(gen, shape):
    return torch.stack([torch.randint(0, n, shape, generator=gen) for n in N_TOKEN], -1)


def _margins(ys):
    """(T, 6): top-1 minus top-2 logit per position and attribute (batch element 0 of each (1, T, n_f) tensor)."""
    out = []
    for y in ys:
        top = y[0].topk(2, dim=-1).values
        out.append(top[:, 0] - top[:, 1])
    return torch.stack(out, -1)


@pytest.fixture()
def ppo_at_window(cuda, tmp_path, monkeypatch):
    def make(W):
        monkeypatch.chdir(tmp_path)
        from rlmg_amd.ppo_policy import config as pcfg, ppo_train as P
        monkeypatch.setattr(P, "N_ACTIONS", W // 2)
        monkeypatch.setattr(P, "NUM_ACTION", W // 2)
        monkeypatch.setattr(P, "N_STATES", W)
        monkeypatch.setattr(P, "WINDOW_SIZE", W)
        monkeypatch.setitem(pcfg.DiscriConfig, "MAX_SEQ", max(2048, W + 2))
        torch.manual_seed(0)
        agent = P.PPO(N_TOKEN, Pretrain=False)
        fill_params(agent.actor_net, seed=71)
        fill_params(agent.critic_net, seed=72)
        for net in (agent.actor_net, agent.critic_net, agent.eval_net):
            net.eval()                                  # dropout off: deterministic on both sides
            net.compute_dtype = torch.float32
        return agent, P
    return make


@pytest.mark.parametrize("T,NA", [(1024, 512), (4096, 2048)])
def test_rollout_gather_all_modes_at_long_windows(cuda, T, NA):
    """cwlt_rollout_gather at NA = 512 / 2048 rows (modes 0 = DQN `-idx` rows, 1 = PPO rows + `+idx` log-prob quirk,
    2 = select_udpate rows) against the reference's loops restated literally."""
    g = torch.Generator().manual_seed(T)
    n_class = (56, 135, 18, 87, 18, 25)
    ys = [torch.randn(2, T, n, generator=g) * 2 for n in n_class]
    W = sum(n_class) + (-sum(n_class)) % 64
    fused = torch.zeros(2, T, W)
    o = 0
    for y in ys:
        fused[..., o:o + y.shape[-1]] = y
        o += y.shape[-1]
    res = ops.heads_forward(fused.view(2 * T, W).to(cuda), n_class, want_argmax=True, want_probs=True)
    ids, probs = res["argmax"].view(2, T, 6), res["probs"].view(2, T, -1)
    for r in range(2):
        yr = [y[r:r + 1] for y in ys]
        a0, _ = rl_ops.rollout_gather(ids[r:r + 1], None, n_class, NA, mode=0)
        assert torch.equal(a0[0].cpu(), rl_math.dqn_choose_action(yr, NA))
        a1, l1 = rl_ops.rollout_gather(ids[r:r + 1], probs[r:r + 1], n_class, NA, mode=1)
        wa, wl = rl_math.ppo_choose_action(yr, NA)
        assert torch.equal(a1[0].cpu(), wa) and (l1[0].cpu() - wl).abs().max().item() < TOL
    a2, l2 = rl_ops.rollout_gather(ids, probs, n_class, NA, mode=2)
    wa, wl = rl_math.ppo_select_update(ys, NA)
    assert torch.equal(a2[-1].cpu(), wa) and (l2[-1].cpu() - wl).abs().max().item() < TOL


@pytest.mark.parametrize("E,NA", [(30, 512), (30, 2048)])
def test_ppo_policy_loss_and_returns_at_config_shapes(cuda, E, NA):
    g = torch.Generator().manual_seed(E + NA)
    new = torch.randn(NA, 6, generator=g) * 0.3 - 0.5
    old_int = (torch.randn(E, NA, 6, generator=g) * 1.2 - 0.8).long()
    rewards, values = torch.rand(E, generator=g), torch.randn(E, 1, generator=g)
    wr = rl_math.ppo_returns([x for x in rewards], 0.99)
    wa = rl_math.ppo_advantages(wr, values)
    gr, ga = rl_ops.ppo_returns_adv(rewards.to(cuda), values.to(cuda), 0.99)
    assert (gr.cpu() - wr).abs().max().item() < TOL and (ga.cpu() - wa).abs().max().item() < TOL
    nr = new.double().requires_grad_(True)
    want = rl_math.ppo_policy_loss(nr, old_int, wa.double(), 0.2)
    want.backward()
    nd = new.to(cuda).requires_grad_(True)
    got = rl_ops.ppo_policy_loss(nd, old_int.to(cuda), ga, 0.2)
    got.backward()
    assert abs(got.item() - want.item()) < TOL
    assert (nd.grad.cpu().double() - nr.grad).abs().max().item() < 1e-6 + TOL * nr.grad.abs().max().item()


@pytest.mark.parametrize("W,E", [(1024, 30), (4096, 6)])
def test_ppo_choose_action_and_select_update_at_repo_dims(cuda, ppo_at_window, W, E):
    """PPO.choose_action on (R, W, 6) states and PPO.select_udpate on (E, W, 6) states through the product path
    (12-layer trunk at repo dims, fused heads, rollout_gather / logp_argmax) vs rl_math on the oracle's logits."""
    agent, P = ppo_at_window(W)
    NA = W // 2
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    ref = fill_params(cw_model.CWLinearTransformer(N_TOKEN, 512, 12, 8, variant="actor"), seed=71).eval()
    g = torch.Generator().manual_seed(W)
    R = 2
    x = _tokens(g, (R, W))
    ga, gl = agent.choose_action(x.to(cuda))
    assert ga.shape == (R, NA, 6) and gl.shape == (R, NA, 6)
    inside = total = 0
    for r in range(R):
        with torch.no_grad():
            ys = ref.forward_output(ref.forward_hidden(x[r:r + 1]))
        wa, wl = rl_math.ppo_choose_action(ys, NA)
        m = _margins(ys)                                            # (W, 6)
        rows = W - 1 - torch.arange(NA)                             # action row i <- position -(i+1)
        safe = m[rows] > MARGIN
        # log-prob of tempo / chord at row i reads the class chosen at position +(i+1): its margin matters too
        fwd = torch.arange(1, NA + 1)
        safe_lp = safe.clone()
        safe_lp[:, 0] &= m[fwd, 0] > MARGIN
        safe_lp[:, 1] &= m[fwd, 1] > MARGIN
        assert torch.equal(ga[r].cpu()[safe], wa[safe])
        assert (gl[r].cpu() - wl)[safe_lp].abs().max().item() < TOL
        inside += (~safe).sum().item()
        total += safe.numel()
    assert inside <= 0.01 * total, (inside, total)
    xs = _tokens(g, (E, W))
    a2, l2, v2 = agent.select_udpate(xs.to(cuda))
    assert a2.shape == (NA, 6) and l2.shape == (NA, 6) and v2.shape == (E, 1)
    with torch.no_grad():
        ys = ref.forward_output(ref.forward_hidden(xs[-1:]))        # only the LAST batch element is returned (:346)
    wa, wl = rl_math.ppo_select_update(ys, NA)
    safe = _margins(ys)[W - 1 - torch.arange(NA)] > MARGIN
    assert torch.equal(a2.cpu()[safe], wa[safe])
    assert (l2.detach().cpu() - wl)[safe].abs().max().item() < TOL
    assert (~safe).sum().item() <= 0.01 * safe.numel()
    # critic value of the last state vs the oracle critic
    cref = fill_params(cw_model.CWLinearTransformer(N_TOKEN, 512, 12, 8, variant="critic"), seed=72).eval()
    with torch.no_grad():
        wv = cref.value_produce(xs[-1:])
    assert (v2[-1].detach().cpu() - wv[0]).abs().max().item() < TOL * max(1.0, wv.abs().max().item())


@pytest.mark.parametrize("W,E", [(1024, 30), (4096, 6)])
def test_ppo_update_rollouts_grouping_at_config_shapes(cuda, ppo_at_window, monkeypatch, W, E):
    """PPO.update_rollouts at window 1024 / 4096, repo dims: stacking 8 rollouts per pass == one at a time."""
    agent, P = ppo_at_window(W)
    NA, R = W // 2, 3
    g = torch.Generator().manual_seed(W + 1)
    states = _tokens(g, (E, R, W)).to(cuda)
    expert = _tokens(g, (R, E + W + 3)).to(cuda)
    mask = torch.ones(R, E + W + 3)
    mask[1, W // 2:] = 0
    mask = mask.to(cuda)
    old_int = (-3 * torch.rand(E, R, NA, 6, generator=g)).long().to(cuda)
    advs = [torch.randn(E, generator=g).to(cuda) for _ in range(R)]
    rets = [torch.randn(E, generator=g).to(cuda) for _ in range(R)]
    for opt in (agent.actor_optim, agent.critic_optim):                # keep the gradients, skip the step
        monkeypatch.setattr(opt, "step", lambda *a, **k: None)

    def grads():
        return [p.grad.detach().clone() for net in (agent.actor_net, agent.critic_net) for p in net.parameters()
                if p.grad is not None]

    agent.update_rollouts(states, old_int, advs, rets, expert, mask, group=1)
    g1 = grads()
    agent.update_rollouts(states, old_int, advs, rets, expert, mask, group=8)
    g8 = grads()
    scale = max(t.abs().max().item() for t in g1)
    assert scale > 1e-5
    worst = max((a - b).abs().max().item() for a, b in zip(g1, g8))
    assert worst < 5e-5 * max(1.0, scale), (worst, scale)
    # and the by-hand recipe for rollout 0 alone reproduces its share: (policy + CE) / R and MSE / R
    agent.actor_sync.zero_grad()
    agent.critic_sync.zero_grad()
    for r in range(R):
        st = states[:, r]
        _, new_logp, value_pred = agent.select_udpate(st)
        pl = rl_ops.ppo_policy_loss(new_logp, old_int[:, r], advs[r], P.PPO_CLIP)
        ce = agent.actor_net.train_step(st, expert[r, :E + W].unfold(0, W, 1)[:E].permute(0, 2, 1),
                                        mask[r, :E + W].unfold(0, W, 1)[:E])
        ((pl + (ce[0] + ce[1] + ce[2] + ce[3] + ce[4] + ce[5]) / 6) / R).backward()
        (torch.nn.functional.mse_loss(rets[r], value_pred).sum() / R).backward()
    g0 = grads()
    worst = max((a - b).abs().max().item() for a, b in zip(g0, g8))
    assert worst < 5e-5 * max(1.0, scale), (worst, scale)


def test_select_pass_on_the_last_state_only_equals_the_literal_pass_over_all_states(cuda, ppo_at_window, monkeypatch):
    """`select_udpate` returns the rows of the LAST batch element only (ppo_train.py:346), so the product runs the actor on
    that state alone; with SELECT_ALL_STATES the actor sees all E states, as in the reference.  Same action rows, same
    log-probs, same critic values, and -- through `update_rollouts` -- the same gradients of both networks."""
    W, E = 1024, 30
    agent, P = ppo_at_window(W)
    NA, R = W // 2, 3
    g = torch.Generator().manual_seed(77)
    states = _tokens(g, (E, R, W)).to(cuda)
    expert = _tokens(g, (R, E + W + 3)).to(cuda)
    mask = torch.ones(R, E + W + 3).to(cuda)
    old_int = (-3 * torch.rand(E, R, NA, 6, generator=g)).long().to(cuda)
    advs = [torch.randn(E, generator=g).to(cuda) for _ in range(R)]
    rets = [torch.randn(E, generator=g).to(cuda) for _ in range(R)]
    for opt in (agent.actor_optim, agent.critic_optim):                # keep the gradients, skip the step
        monkeypatch.setattr(opt, "step", lambda *a, **k: None)
    out = {}
    for literal in (False, True):
        monkeypatch.setattr(P, "SELECT_ALL_STATES", literal)
        with torch.no_grad():
            a, lp, v = agent.select_udpate(states[:, 1])
        agent.update_rollouts(states, old_int, advs, rets, expert, mask, group=8)
        out[literal] = (a.clone(), lp.clone(), v.clone(),
                        [p.grad.detach().clone() for net in (agent.actor_net, agent.critic_net)
                         for p in net.parameters() if p.grad is not None])
    (a0, l0, v0, g0), (a1, l1, v1, g1) = out[False], out[True]
    assert torch.equal(a0, a1) and a0.shape == (NA, 6)
    assert (l0 - l1).abs().max().item() < 1e-5 and (v0 - v1).abs().max().item() < 1e-6
    scale = max(t.abs().max().item() for t in g1)
    assert scale > 1e-5
    worst = max((x - y).abs().max().item() for x, y in zip(g0, g1))
    assert worst < 5e-5 * max(1.0, scale), (worst, scale)


def test_ppo_rollout_step_at_config2_window(cuda, ppo_at_window):
    """One env step of configs[2] (window 1024) for 3 rollouts through PPO.rollout_step (graph replay) == the eager
    device function; next state = first half of the window + the action rows (ppo_train.py:483)."""
    agent, P = ppo_at_window(1024)
    W, NA, R = 1024, 512, 3
    g = torch.Generator().manual_seed(5)
    x = _tokens(g, (R, W)).to(cuda)
    m = torch.ones(R, W, device=cuda)
    a, lp, ns, v, rw = agent._rollout_step_device(x, m)
    a2, lp2, ns2, v2, rw2 = agent.rollout_step(x, m)
    assert torch.equal(a, a2) and torch.equal(ns, ns2)
    assert (lp - lp2).abs().max().item() < 1e-5 and (v - v2).abs().max().item() < 1e-5
    assert (rw - rw2).abs().max().item() < 1e-5
    assert torch.equal(ns[:, :NA], x[:, :NA]) and torch.equal(ns[:, NA:], a)
    assert v.shape == (R, 1) and rw.shape == (R, 1) and ((rw > 0) & (rw < 1)).all()
