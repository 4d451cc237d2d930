"""GPU: hipGraph replay of the launch-bound RL rollout steps (ops.GraphedCall) -- same results as the eager
launches, parameters read in place, fresh dropout masks on every replay through the device-side seed base."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401
from rlmg_amd import ops  # noqa: E402

pytestmark = pytest.mark.gpu


def _small(cfg_dict):
    old = dict(cfg_dict)
    cfg_dict.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    return old


def _same_trajectory(p0, p1, name):
    """Adam turns last-bit gradient differences (capturable vs default code path) of near-zero gradients into
    steps of up to ~lr, so a handful of elements may sit a few lr apart; everything else must agree closely."""
    d = (p0 - p1).abs()
    assert d.max().item() < 5e-3, name
    assert (d > 1e-4).float().mean().item() < 2e-2, name


def test_graphed_dropout_draws_fresh_masks_per_replay(cuda):
    x = torch.ones(4096, 512, device=cuda)
    call = ops.GraphedCall(lambda t: ops.posenc_dropout(t, None, 64, 0.5, 1234))
    a = call(x).clone()
    b = call(x).clone()
    c = call(2 * x).clone()
    for t, scale in ((a, 2.0), (b, 2.0), (c, 4.0)):
        kept = t != 0
        assert abs(kept.float().mean().item() - 0.5) < 0.01
        assert torch.all(t[kept] == scale)
    assert 0.4 < ((a != 0) ^ (b != 0)).float().mean().item() < 0.6      # independent masks
    assert 0.4 < ((b != 0) ^ (c != 0)).float().mean().item() < 0.6
    # eager launches do not use the seed base: same seed -> same mask, before and after replays
    e1 = ops.posenc_dropout(x, None, 64, 0.5, 1234)
    call(x)
    e2 = ops.posenc_dropout(x, None, 64, 0.5, 1234)
    assert torch.equal(e1, e2)


def test_graphed_dqn_choose_action_equals_eager_and_sees_weight_updates(cuda, monkeypatch):
    from rlmg_amd.dqn_policy import IRL_dqn_train as T, config
    old = _small(config.AgentConfig)
    try:
        n_class = [56, 135, 18, 87, 18, 25]
        agent = T.DQN(n_class, Pretrain=False)
        fill_params(agent.eval_net, seed=61)
        agent.eval_net.eval()
        g = torch.Generator().manual_seed(1)
        xs = [torch.stack([torch.randint(0, n, (1, 50), generator=g) for n in n_class], -1).to(cuda) for _ in range(3)]
        monkeypatch.setattr(ops, "GRAPHS_ENABLED", False)
        eager = [agent.choose_action(x) for x in xs]
        monkeypatch.setattr(ops, "GRAPHS_ENABLED", True)
        graphed = [agent.choose_action(x) for x in xs]
        assert all(torch.equal(a, b) for a, b in zip(eager, graphed))
        assert len(agent._graph_choose.graphs) == 1                       # one capture, three replays
        with torch.no_grad():                                             # an "optimizer step": in-place update
            for p in agent.eval_net.parameters():
                p.mul_(-1.0)
        monkeypatch.setattr(ops, "GRAPHS_ENABLED", False)
        eager2 = agent.choose_action(xs[0])
        monkeypatch.setattr(ops, "GRAPHS_ENABLED", True)
        graphed2 = agent.choose_action(xs[0])
        assert torch.equal(eager2, graphed2) and not torch.equal(eager2, eager[0])
        # a batch of rollouts is a second signature
        xb = torch.cat(xs, 0)
        got = agent.choose_action(xb)
        monkeypatch.setattr(ops, "GRAPHS_ENABLED", False)
        want = agent.choose_action(xb)
        assert got.shape == (3, 25, 6) and torch.equal(got, want) and torch.equal(got[0], eager2)
        assert len(agent._graph_choose.graphs) == 2
    finally:
        config.AgentConfig.update(old)


def test_graphed_ppo_rollout_step_equals_eager(cuda, monkeypatch):
    from rlmg_amd.ppo_policy import config, ppo_train as P
    old_a, old_d = _small(config.ActorConfig), _small(config.DiscriConfig)
    try:
        n_token = [49, 19, 19, 89, 67, 25]
        agent = P.PPO(n_token, Pretrain=False)
        for i, net in enumerate((agent.actor_net, agent.critic_net, agent.eval_net)):
            fill_params(net, seed=80 + i)
            net.eval()
        g = torch.Generator().manual_seed(2)
        x = torch.stack([torch.randint(0, n, (2, 50), generator=g) for n in n_token], -1).to(cuda)
        mask = torch.ones(2, 50, device=cuda)
        mask[1, 40:] = 0
        monkeypatch.setattr(ops, "GRAPHS_ENABLED", False)
        eager = agent.rollout_step(x, mask)
        monkeypatch.setattr(ops, "GRAPHS_ENABLED", True)
        for _ in range(2):
            graphed = agent.rollout_step(x, mask)
            for a, b in zip(eager, graphed):
                assert a.shape == b.shape
                assert torch.equal(a, b) if a.dtype == torch.int64 else (a - b).abs().max().item() < 1e-6
        action, logp, nxt, value, reward = graphed
        assert action.shape == (2, 25, 6) and logp.shape == (2, 25, 6) and nxt.shape == (2, 50, 6)
        assert torch.equal(nxt[:, :25], x[:, :25]) and torch.equal(nxt[:, 25:], action)
        assert value.shape == (2, 1) and reward.shape == (2, 1)
        # train() mode: dropout live, replays differ from each other (seed base advances)
        for net in (agent.actor_net, agent.critic_net, agent.eval_net):
            net.train()
        agent._graph_step = None
        v1 = agent.rollout_step(x, mask)[3]
        v2 = agent.rollout_step(x, mask)[3]
        assert torch.isfinite(v1).all() and not torch.equal(v1, v2)
    finally:
        config.ActorConfig.update(old_a)
        config.DiscriConfig.update(old_d)


def test_graphed_dqn_update_matches_eager_updates(cuda, tmp_path, monkeypatch):
    """Whole-step capture (forward + backward + Adam + LR tensor): five updates with the third one captured and the
    rest replayed follow the same parameter trajectory as five eager updates (dropout off)."""
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.dqn_policy import IRL_dqn_train as T, config
    old = _small(config.AgentConfig)
    try:
        n_class = [56, 135, 18, 87, 18, 25]
        g = torch.Generator().manual_seed(11)
        B = 6
        tok = lambda *s: torch.stack([torch.randint(0, n, s, generator=g) for n in n_class], -1).to(cuda)  # noqa: E731
        batches = [{"state": tok(B, 50), "action": tok(B, 25), "reward": torch.rand(B, 1, generator=g),
                    "nextstate": tok(B, 50), "done": torch.zeros(B, 1)} for _ in range(5)]
        m = torch.ones(B, 50, device=cuda)

        def run(graphs):
            monkeypatch.setattr(ops, "GRAPHS_ENABLED", graphs)
            monkeypatch.setattr(ops, "TRAIN_GRAPHS", graphs)
            agent = T.DQN(n_class, Pretrain=False)
            fill_params(agent.eval_net, seed=61)
            agent.eval_net.eval()
            agent.target_net.eval()
            agent.scheduler = torch.optim.lr_scheduler.MultiStepLR(agent.optim, milestones=[2, 4], gamma=0.5)
            losses = [agent.update(tr, dict(tr), m, False, 0) for tr in batches]
            return agent, losses

        eager, l0 = run(False)
        graphed, l1 = run(True)
        assert len(graphed._graph_update.graphs) == 1
        assert isinstance(graphed.optim.param_groups[0]["lr"], torch.Tensor)
        assert abs(float(graphed.optim.param_groups[0]["lr"]) - eager.optim.param_groups[0]["lr"]) < 1e-9      # f32 tensor
        for a, b in zip(l0, l1):
            assert all(abs(x - y) < 2e-4 * max(1.0, abs(x)) for x, y in zip(a, b))
        for (k, p0), (_, p1) in zip(eager.eval_net.named_parameters(), graphed.eval_net.named_parameters()):
            _same_trajectory(p0, p1, k)
    finally:
        config.AgentConfig.update(old)


def test_graphed_ppo_update_policy_matches_eager(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.ppo_policy import config, ppo_train as P
    old_a, old_d = _small(config.ActorConfig), _small(config.DiscriConfig)
    try:
        n_token = [49, 19, 19, 89, 67, 25]
        g = torch.Generator().manual_seed(12)
        E = P.BUFFER_SIZE
        tok = lambda *s: torch.stack([torch.randint(0, n, s, generator=g) for n in n_token], -1).to(cuda)  # noqa: E731
        states, exp_states = tok(E, 50), tok(E, 50)
        logp = -3 * torch.rand(E, 25, 6, generator=g).to(cuda)
        adv, ret = torch.randn(E, 1, generator=g).to(cuda), torch.randn(E, generator=g).to(cuda)

        def run(graphs):
            monkeypatch.setattr(ops, "GRAPHS_ENABLED", graphs)
            monkeypatch.setattr(ops, "TRAIN_GRAPHS", graphs)
            agent = P.PPO(n_token, Pretrain=False)
            fill_params(agent.actor_net, seed=71)
            fill_params(agent.critic_net, seed=72)
            agent.actor_net.eval()
            agent.critic_net.eval()
            P.AgentBuffer, P.ExpertBuffer = P.AgentMemory(), P.ExpertMemory()
            P.AgentBuffer.states_agent.copy_(states)
            P.AgentBuffer.log_actions_agent.copy_(logp)
            P.ExpertBuffer.states_exp.copy_(exp_states)
            P.ExpertBuffer.mask_state.fill_(1)
            loss = agent.update_policy(5, P.PPO_CLIP, adv, ret)
            return agent, loss

        eager, l0 = run(False)
        graphed, l1 = run(True)
        assert len(graphed._graph_ppo_step.graphs) == 1
        assert abs(l0 - l1) < 2e-4 * max(1.0, abs(l0))
        for net in ("actor_net", "critic_net"):
            for (k, p0), (_, p1) in zip(getattr(eager, net).named_parameters(), getattr(graphed, net).named_parameters()):
                _same_trajectory(p0, p1, (net, k))
    finally:
        config.ActorConfig.update(old_a)
        config.DiscriConfig.update(old_d)


def test_weight_shadows_follow_the_parameters(cuda, tmp_path, monkeypatch):
    """ops.ShadowSet (persistent bf16 copies of the master weights): an eager forward always computes with the current
    parameters -- after an in-place update, after load_state_dict, after a FUSED optimizer step (which does not bump
    autograd's version counters), after the replay of a captured optimizer step -- and two forwards before one
    backward (PPO's pattern) give the gradients of recomputed casts."""
    monkeypatch.chdir(tmp_path)
    from rlmg_amd.dqn_policy import IRL_dqn_train as T, config, model
    old = _small(config.AgentConfig)
    try:
        n_class = [56, 135, 18, 87, 18, 25]
        g = torch.Generator().manual_seed(4)
        x = torch.stack([torch.randint(0, n, (2, 32), generator=g) for n in n_class], -1).to(cuda)

        def fresh_forward(net):                     # the same net with every shadow dropped: casts recomputed
            enc = net.transformer_encoder
            keep = enc._shadow
            enc._shadow = None
            for p in net.parameters():
                p.__dict__.pop("_cwlt_shadow", None)
            with torch.no_grad():
                out = net.forward_hidden(x).float().clone()
            enc._shadow = keep
            return out

        net = fill_params(model.LinearTransformer(n_class), seed=5).to(cuda).eval()
        net.compute_dtype = torch.bfloat16
        with torch.no_grad():
            h0 = net.forward_hidden(x).float().clone()
            lin = net.transformer_encoder.layers[1].linear2
            lin.weight.mul_(1.5)                                            # in-place update: version bump
            h1 = net.forward_hidden(x).float().clone()
        assert (h1 - h0).abs().max().item() > 1e-3 and torch.equal(h1, fresh_forward(net))
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        sd["transformer_encoder.layers.0.attention.key_projection.weight"] *= 0.5
        net.load_state_dict(sd)
        with torch.no_grad():
            h2 = net.forward_hidden(x).float().clone()
        assert (h2 - h1).abs().max().item() > 1e-4 and torch.equal(h2, fresh_forward(net))

        # fused Adam moves the parameters without touching their version counters
        opt = torch.optim.Adam(net.parameters(), lr=1e-2, fused=True)
        net.train()
        tgt0 = torch.stack([torch.randint(0, n, (2, 32), generator=g) for n in n_class], -1).to(cuda)
        sum(net.train_step(x, tgt0, torch.ones(2, 32, device=cuda))).backward()
        v0 = net.in_linear.weight._version
        opt.step()
        net.eval()
        with torch.no_grad():
            h3 = net.forward_hidden(x).float().clone()
        assert (h3 - h2).abs().max().item() > 1e-3 and torch.equal(h3, fresh_forward(net)), net.in_linear.weight._version - v0
        # inside ops.frozen_weights() the caller vouches that nothing changes: copies are refreshed once and reused
        with ops.frozen_weights(), torch.no_grad():
            a1 = net.forward_hidden(x).float().clone()
            lin.weight.mul_(1.25)
            a2 = net.forward_hidden(x).float().clone()
        assert torch.equal(a1, a2)                                          # by contract: the stale copy
        with torch.no_grad():
            a3 = net.forward_hidden(x).float().clone()
        assert (a3 - a1).abs().max().item() > 1e-4 and torch.equal(a3, fresh_forward(net))

        # captured optimizer step: parameters move without a version bump
        monkeypatch.setattr(ops, "GRAPHS_ENABLED", True)
        monkeypatch.setattr(ops, "TRAIN_GRAPHS", True)
        agent = T.DQN(n_class, Pretrain=False)
        fill_params(agent.eval_net, seed=61)
        agent.eval_net.eval()
        agent.target_net.eval()
        B = 6
        tok = lambda *s: torch.stack([torch.randint(0, n, s, generator=g) for n in n_class], -1).to(cuda)  # noqa: E731
        m = torch.ones(B, 50, device=cuda)
        for i in range(5):
            tr = {"state": tok(B, 50), "action": tok(B, 25), "reward": torch.rand(B, 1, generator=g),
                  "nextstate": tok(B, 50), "done": torch.zeros(B, 1)}
            agent.update(tr, dict(tr), m, False, 0)
            with torch.no_grad():
                got = agent.eval_net.forward_hidden(x).float().clone()
            assert torch.equal(got, fresh_forward(agent.eval_net)), i
        assert len(agent._graph_update.graphs) == 1

        # two forwards, then the backward of the FIRST one: no autograd version error, gradients as with fresh casts
        net.train()
        tgt = torch.stack([torch.randint(0, n, (2, 32), generator=g) for n in n_class], -1).to(cuda)
        mask = torch.ones(2, 32, device=cuda)
        torch.manual_seed(3)
        l_first = sum(net.train_step(x, tgt, mask))
        sum(net.train_step(x, tgt, mask))
        net.zero_grad()
        l_first.backward()
        g_shadow = net.transformer_encoder.layers[0].linear1.weight.grad.clone()
        net.transformer_encoder._shadow = None
        torch.manual_seed(3)
        l_ref = sum(net.train_step(x, tgt, mask))
        net.zero_grad()
        l_ref.backward()
        assert torch.equal(g_shadow, net.transformer_encoder.layers[0].linear1.weight.grad)
    finally:
        config.AgentConfig.update(old)


def test_captures_are_counted_and_memset_nodes_are_rewritten_as_kernels(cuda):
    """GraphedCall counts the node kinds of what it captured (hipGraphGetNodes).  A captured hipMemsetAsync replays a wrong
    fill pattern on ROCm 7.2 from the second replay on (tools/probes/graph_memset_probe.py,
    profiles/r03_graph_memset_probe.txt), so every memset node is replaced by a fill-kernel node before the graph is
    instantiated (cwlt_graph_replace_memset_nodes): the replays then write the right bytes."""
    import ctypes
    import os
    from rlmg_amd import _lib
    x = torch.randn(64, 512, device=cuda).bfloat16()
    ok = ops.GraphedCall(lambda t: ops.colsum(t).clone())
    want = x.float().sum(0)
    for _ in range(3):
        got = ok(x)
        assert (got - want).abs().max().item() < 1e-2 * want.abs().max().item()
    assert ok.census.get("kernel", 0) >= 2 and ok.census.get("memset", 0) == 0 and ok.memsets_replaced == 0
    # a step that zeroes buffers with hipMemsetAsync (what PyTorch's reductions and hipBLASLt do inside a step): 1-, 2-
    # and 4-byte patterns, an odd length
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    buf = torch.ones(1 << 20, device=cuda)
    b16 = torch.ones(1001, device=cuda, dtype=torch.int16)
    b32 = torch.ones(777, device=cuda, dtype=torch.int32)

    def step(t):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(buf.numel() * 4), st) == 0
        assert hip.hipMemsetD16Async(ctypes.c_void_p(b16.data_ptr()), ctypes.c_ushort(0x1234), ctypes.c_size_t(1001), st) == 0
        assert hip.hipMemsetD32Async(ctypes.c_void_p(b32.data_ptr()), ctypes.c_int(-7), ctypes.c_size_t(777), st) == 0
        return buf[:4096] + t

    g = ops.GraphedCall(step)
    t = torch.randn(4096, device=cuda)
    for _ in range(5):
        buf.fill_(float("nan"))                          # poison: only a correct fill makes the result finite
        b16.fill_(-1)
        b32.fill_(5)
        torch.mm(torch.randn(2048, 2048, device=cuda), torch.randn(2048, 2048, device=cuda))   # work in between
        out = g(t).clone()
        assert torch.equal(out, t) and torch.count_nonzero(buf).item() == 0
        assert bool((b16 == 0x1234).all()) and bool((b32 == -7).all())
    assert g.census.get("memset", 0) == 3 and g.memsets_replaced == 3
    assert _lib.GRAPH_NODE_TYPES[2] == "memset"
