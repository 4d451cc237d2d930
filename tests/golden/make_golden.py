"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own classes.

Run in the build container only (needs /root/reference, which never travels to the GPU box):
    python tests/golden/make_golden.py

What runs: /root/reference/dqn_policy/model.py::LinearTransformer and
/root/reference/ppo_policy/model.py::{Actor_Transformer, Critic_Transformer}, imported unmodified
from their own directory.  Their top-level `from fast_transformers...` imports are satisfied by
oracle/ft_standin (a build-owned stand-in backed by oracle/ft_encoder.py), because the real
package (pytorch-fast-transformers==0.4.0) is a third-party dependency that is absent from the
reference checkout and from this image.  So these fixtures pin the reference's OWN code
(embeddings, in_linear, positional encoding, heads, CE loss, value heads) and are "parity unpinned"
for the encoder body, which comes from the oracle restatement.

To keep the fixtures small the reference classes are instantiated with their config dicts patched
to a small network (D_MODEL 128, N_LAYER 2, N_HEAD 2 -- BASELINE.json configs[0] dims); one fixture
uses the repo dims (512/12/8) with T=8 and stores outputs only.

Weights are not stored: tests/golden/fill.py overwrites every parameter with a deterministic
name-keyed fill, applied identically here (to the reference's modules) and in the tests.
Fixture = inputs + the reference's outputs / gradient slices (data only).
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from fill import fill_params  # noqa: E402
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def _import_reference(subdir):
    from oracle import ft_standin
    ft_standin.install()
    for m in ("model", "config"):
        sys.modules.pop(m, None)
    path = os.path.join(REF, subdir)
    sys.path.insert(0, path)
    try:
        config = importlib.import_module("config")
        model = importlib.import_module("model")
    finally:
        sys.path.remove(path)
    return config, model


def _transformers_placeholders():
    """AIRL_model.py:10 imports two names that are gone from the transformers release in this image and that
    nothing uses; give the (lazy, self-replacing) transformers module placeholders for them."""
    from transformers import LongformerModel  # noqa: F401  -- settles the lazy module object first
    tf = sys.modules["transformers"]
    for name in ("TrajectoryTransformerConfig", "TrajectoryTransformerModel"):
        if not hasattr(tf, name):
            setattr(tf, name, type(name, (), {}))


def _np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items() if k != "pos_emb.pe"}


def _tokens(gen, shape, n_class):
    return torch.stack([torch.randint(0, n, shape, generator=gen) for n in n_class], -1)


def dqn_small():
    config, model = _import_reference("dqn_policy")
    config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    n_class = [56, 135, 18, 87, 18, 25]            # IRL_dqn_train.py:403
    net = fill_params(model.LinearTransformer(n_class), seed=11).eval()
    gen = torch.Generator().manual_seed(1234)
    x, y = _tokens(gen, (2, 16), n_class), _tokens(gen, (2, 16), n_class)
    mask = torch.ones(2, 16)
    mask[1, 12:] = 0
    h = net.forward_hidden(x)
    logits = net.forward_output(h, y)
    losses = net.train_step(x, y, mask)
    loss = sum(losses) / 6
    net.zero_grad()
    loss.backward()
    out = {"x": x.numpy(), "y": y.numpy(), "mask": mask.numpy(), "h": h.detach().numpy(),
           "losses": np.array([l.item() for l in losses], dtype=np.float64),
           "grad.in_linear.weight": net.in_linear.weight.grad[:16, ::19].numpy(),
           "grad.q0": net.transformer_encoder.layers[0].attention.query_projection.weight.grad[::8, ::8].numpy(),
           "grad.k1": net.transformer_encoder.layers[1].attention.key_projection.weight.grad[::8, ::8].numpy(),
           "grad.lin1": net.transformer_encoder.layers[1].linear1.weight.grad[::64, ::8].numpy(),
           "grad.norm1": net.transformer_encoder.layers[0].norm1.weight.grad.numpy(),
           "grad.emb_pitch": net.word_emb_pitch.lut.weight.grad[:, ::16].numpy(),
           "grad.proj_chord.bias": net.proj_chord.bias.grad.numpy(),
           "n_class": np.array(n_class)}
    for i, l in enumerate(logits):
        out["logits%d" % i] = l.detach().numpy()
    out["keys"] = np.array(sorted(net.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, "dqn_small.npz"), **out)
    config.AgentConfig.update({"D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8})
    return net


def dqn_repo_dims():
    """Repo dims (512/12/8), short T."""
    config, model = _import_reference("dqn_policy")
    n_class = [56, 135, 18, 87, 18, 25]
    net = fill_params(model.LinearTransformer(n_class), seed=12).eval()
    gen = torch.Generator().manual_seed(99)
    x = _tokens(gen, (1, 8), n_class)
    with torch.no_grad():
        h = net.forward_hidden(x)
        logits = net.forward_output(h, None)
    out = {"x": x.numpy(), "h": h.numpy(), "n_params": np.array(model.network_paras(net)),
           "keys": np.array(sorted(net.state_dict().keys()))}
    for i, l in enumerate(logits):
        out["logits%d" % i] = l.numpy()
    np.savez_compressed(os.path.join(HERE, "dqn_repo_dims.npz"), **out)


def ppo_small():
    config, model = _import_reference("ppo_policy")
    config.ActorConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    n_token = [49, 19, 19, 89, 67, 25]             # ppo_policy/prepare_data.py:247-291
    actor = fill_params(model.Actor_Transformer(n_token), seed=21).eval()
    critic = fill_params(model.Critic_Transformer(n_token), seed=22).eval()
    gen = torch.Generator().manual_seed(4321)
    x, y = _tokens(gen, (3, 50), n_token), _tokens(gen, (3, 50), n_token)
    mask = torch.ones(3, 50, dtype=torch.long)       # ppo passes an int64 mask (ppo_train.py:207,398)
    with torch.no_grad():
        h = actor.forward_hidden(x)
        logits = actor.forward_output(h)
        value_fn = actor.value_funtion(h[0])
        losses = actor.train_step(x, y, mask)
        v = critic.value_produce(x)
    out = {"x": x.numpy(), "y": y.numpy(), "mask": mask.numpy(), "h": h.numpy(), "value_funtion": value_fn.numpy(),
           "losses": np.array([l.item() for l in losses], dtype=np.float64), "critic_value": v.numpy(),
           "n_token": np.array(n_token),
           "actor_keys": np.array(sorted(actor.state_dict().keys())),
           "critic_keys": np.array(sorted(critic.state_dict().keys()))}
    for i, l in enumerate(logits):
        out["logits%d" % i] = l.numpy()
    np.savez_compressed(os.path.join(HERE, "ppo_small.npz"), **out)
    config.ActorConfig.update({"D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8})


def ppo_reward_small():
    """ppo_policy/model.py::LongFormer (frozen reward model), eval mode, small dims."""
    config, model = _import_reference("ppo_policy")
    config.DiscriConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    n_token = [49, 19, 19, 89, 67, 25]
    net = fill_params(model.LongFormer(n_token), seed=31).eval()
    gen = torch.Generator().manual_seed(77)
    x = _tokens(gen, (3, 50), n_token)
    mask = torch.ones(3, 50, dtype=torch.long)
    mask[1, 40:] = 0
    with torch.no_grad():
        r = net.token_forward(x, None, mask)
    np.savez_compressed(os.path.join(HERE, "ppo_reward_small.npz"), x=x.numpy(), mask=mask.numpy(), reward=r.numpy(),
                        n_token=np.array(n_token), keys=np.array(sorted(net.state_dict().keys())))
    config.DiscriConfig.update({"D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8})


def airl_small():
    """dqn_policy/AIRL_model.py::LongFormer (AIRL discriminator), eval mode.  Its module-level constants are
    patched to a small net after import; line 10's dead import of TrajectoryTransformer* (gone from the
    transformers release in this image) is satisfied with placeholder names that nothing uses."""
    _transformers_placeholders()
    from oracle import ft_standin
    ft_standin.install()
    sys.modules.pop("AIRL_model", None)
    path = os.path.join(REF, "dqn_policy")
    sys.path.insert(0, path)
    try:
        am = importlib.import_module("AIRL_model")
    finally:
        sys.path.remove(path)
    am.D_MODEL, am.N_LAYER, am.N_HEAD = 128, 2, 2
    n_class = [56, 135, 18, 87, 18, 25]
    net = fill_params(am.LongFormer(n_class), seed=41).eval()
    with torch.no_grad():      # non-trivial BatchNorm running statistics
        net.score_classifier[1].running_mean.copy_(torch.linspace(-0.2, 0.2, 128))
        net.score_classifier[1].running_var.copy_(torch.linspace(0.5, 1.5, 128))
    gen = torch.Generator().manual_seed(78)
    x = _tokens(gen, (4, 50), n_class)
    mask = torch.ones(4, 50, dtype=torch.long)
    mask[2, 45:] = 0
    with torch.no_grad():
        score = net(x, mask)
    np.savez_compressed(os.path.join(HERE, "airl_small.npz"), x=x.numpy(), mask=mask.numpy(), score=score.numpy(),
                        n_class=np.array(n_class), keys=np.array(sorted(net.state_dict().keys())))


def airl_grads_small():
    """Gradients of the discriminator's training loss (dqn_policy/AIRL.py:150-170: BCE(expert, 1) + BCE(agent, 0) +
    token CE of the agent windows vs the expert windows) through the reference's own AIRL_model.LongFormer + HF
    Longformer, eval mode (dropout off, BatchNorm running statistics) so that it is deterministic.  Stored: the
    three loss terms, the L2 norm of every parameter's gradient, and the gradient itself (first 8 rows of
    tensors with more than 4096 elements) for every parameter."""
    _transformers_placeholders()
    from oracle import ft_standin
    ft_standin.install()
    sys.modules.pop("AIRL_model", None)
    path = os.path.join(REF, "dqn_policy")
    sys.path.insert(0, path)
    try:
        am = importlib.import_module("AIRL_model")
    finally:
        sys.path.remove(path)
    am.D_MODEL, am.N_LAYER, am.N_HEAD = 128, 2, 2
    n_class = [56, 135, 18, 87, 18, 25]
    net = fill_params(am.LongFormer(n_class), seed=43).eval()
    with torch.no_grad():
        net.score_classifier[1].running_mean.copy_(torch.linspace(-0.2, 0.2, 128))
        net.score_classifier[1].running_var.copy_(torch.linspace(0.5, 1.5, 128))
    gen = torch.Generator().manual_seed(79)
    x_exp = _tokens(gen, (3, 50), n_class)
    x_agent = _tokens(gen, (3, 50), n_class)
    mask = torch.ones(3, 50, dtype=torch.long)
    mask[1, 44:] = 0
    bce = torch.nn.BCELoss()
    exp_l = bce(net(x_exp, mask), torch.ones(3, 1))
    ce_l = net.token_forward(x_agent, x_exp, mask)
    agent_l = bce(net(x_agent, mask), torch.zeros(3, 1))
    (exp_l + (agent_l + ce_l)).backward()
    out = {"x_exp": x_exp.numpy(), "x_agent": x_agent.numpy(), "mask": mask.numpy(), "n_class": np.array(n_class),
           "losses": np.array([exp_l.item(), agent_l.item(), ce_l.item()], dtype=np.float64)}
    names, norms = [], []
    for k, p_ in net.named_parameters():
        if p_.grad is None:
            continue
        g = p_.grad.detach()
        names.append(k)
        norms.append(g.double().norm().item())
        out["grad." + k] = (g[:8] if g.numel() > 4096 else g).numpy()
    out["names"] = np.array(names)
    out["norms"] = np.array(norms)
    np.savez_compressed(os.path.join(HERE, "airl_grads_small.npz"), **out)


def dqn_generation_small():
    """Generation (dqn_policy/testing-no-type-cp.py:126-179 loop, driven here on the CPU): the reference's own
    LinearTransformer(is_training=False) -- forward_hidden(x, memory, is_training=False) one token at a time from
    the Bar token, then ITS forward_output_sampling (its numpy samplers, seeded np.random).  Stored: the token
    stream, and per step the hidden row and the six logit vectors the samplers saw."""
    config, model = _import_reference("dqn_policy")
    config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    n_class = [56, 135, 18, 87, 18, 25]
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        net = fill_params(model.LinearTransformer(n_class, is_training=False), seed=81).eval()
    steps = 48
    np.random.seed(20240)
    tokens, hs, logits = [np.array([0, 0, 1, 0, 0, 0])], [], []
    with torch.no_grad():
        h, memory = net.forward_hidden(torch.from_numpy(tokens[0]).long().view(1, 1, 6), None, is_training=False)
        for _ in range(steps):
            hs.append(h.numpy().reshape(-1))
            logits.append(np.concatenate([y.numpy().reshape(-1) for y in net.forward_output(h, None)]))
            nxt = net.forward_output_sampling(h)
            tokens.append(np.asarray(nxt))
            h, memory = net.forward_hidden(torch.from_numpy(np.asarray(nxt)).long().view(1, 1, 6), memory,
                                           is_training=False)
    out = {"tokens": np.stack(tokens).astype(np.int64), "h": np.stack(hs), "logits": np.stack(logits),
           "n_class": np.array(n_class), "np_seed": np.array(20240), "fill_seed": np.array(81)}
    np.savez_compressed(os.path.join(HERE, "dqn_generation_small.npz"), **out)
    config.AgentConfig.update({"D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8})


def _import_reference_ppo_train():
    """/root/reference/ppo_policy/ppo_train.py, unmodified, on CPU (`config.device` falls back to cpu, config.py:11).
    Beside `model` / `config` it imports wandb (absent here; every call to it is commented out in the file), tqdm and
    matplotlib (both present): wandb gets an empty placeholder module that exists only inside this generator."""
    import types
    config, model = _import_reference("ppo_policy")
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    sys.modules.pop("ppo_train", None)
    path = os.path.join(REF, "ppo_policy")
    sys.path.insert(0, path)
    try:
        pt = importlib.import_module("ppo_train")
    finally:
        sys.path.remove(path)
    return config, model, pt


def ppo_rl_small():
    """The reference's own PPO class and buffers (ppo_policy/ppo_train.py:69-417) driven exactly as its main loop
    drives them (:460-506) for one song: 30 env steps (choose_action -> next_state = cat(state[:25], action) ->
    critic value -> reward model -> store_transition x2), then calculate_returns / calculate_advantages,
    select_udpate, and ONE inner step of update_policy.  Small nets (128 / 2 / 2), name-keyed fills, all three nets in
    eval() so that the record is deterministic (the reference rolls out with dropout live).  Also: ring overwrite and
    seeded sampling of AgentMemory / ExpertMemory with BUFFER_SIZE patched to 8."""
    config, model, pt = _import_reference_ppo_train()
    small = {"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2}
    config.ActorConfig.update(small)
    config.CriticConfig.update(small)
    config.DiscriConfig.update(small)
    n_token = [49, 19, 19, 89, 67, 25]
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        agent = pt.PPO(n_token, Pretrain=False)
    fill_params(agent.actor_net, seed=21).eval()
    fill_params(agent.critic_net, seed=22).eval()
    fill_params(agent.eval_net, seed=31).eval()
    dev = pt.device
    gen = torch.Generator().manual_seed(2468)
    E, W, NA = pt.EPISODES, pt.WINDOW_SIZE, pt.N_ACTIONS
    L = E + 50 + W + 4
    train_x = _tokens(gen, (W,), n_token)
    expert_x = _tokens(gen, (L,), n_token)
    train_mask = torch.ones(L)
    train_mask[70:] = 0
    out = {"n_token": np.array(n_token), "state0": train_x.numpy(), "expert_x": expert_x.numpy(),
           "train_mask": train_mask.numpy()}
    # ---- rollout, ppo_train.py:467-496 ----
    pt.AgentBuffer, pt.ExpertBuffer, pt.Agent = pt.AgentMemory(), pt.ExpertMemory(), agent
    state_x = train_x
    acts, logps, vals, rews = [], [], [], []
    with torch.no_grad():
        for num in range(E):
            Expert_state = expert_x[num: num + W]
            Expert_next_state = expert_x[num + 50: num + 50 + W]
            Expert_reward = torch.tensor(1.0).float().to(dev)
            Expert_done = torch.tensor(0).long().to(dev)
            Expert_mask_state = train_mask[num: num + W]
            Expert_mask_nextstate = train_mask[num + 1: num + 1 + W]
            done = torch.tensor(0).long().to(dev)
            action, log_prob_res = agent.choose_action(state_x.unsqueeze(0))
            next_state = torch.cat((state_x[:NA, :], action), dim=0)
            state_x = next_state
            value_state = agent.critic_net.value_produce(state_x.unsqueeze(0))
            agent_reward = agent.eval_net.token_forward(state_x.unsqueeze(0), Expert_state, Expert_mask_state.unsqueeze(0))
            pt.AgentBuffer.store_transition(state_x, action, log_prob_res, value_state, agent_reward, next_state, done)
            pt.ExpertBuffer.store_transition(Expert_state, action, Expert_reward, Expert_next_state, Expert_done,
                                             Expert_mask_state, Expert_mask_nextstate)
            acts.append(action.numpy()); logps.append(log_prob_res.numpy())
            vals.append(value_state.numpy().reshape(())); rews.append(agent_reward.numpy().reshape(()))
    out.update(actions=np.stack(acts), logps=np.stack(logps), values=np.stack(vals), rewards=np.stack(rews))
    agent_all = pt.AgentBuffer.get()
    expert_all = pt.ExpertBuffer.get()
    for k, v in agent_all.items():
        out["agent_get." + k] = v.numpy()
    for k, v in expert_all.items():
        out["expert_get." + k] = v.numpy()
    np.random.seed(97)
    for i, t in enumerate(pt.AgentBuffer.sampling(6)):
        out["agent_sample.%d" % i] = t.numpy()
    for i, t in enumerate(pt.ExpertBuffer.sampling(6)):
        out["expert_sample.%d" % i] = t.numpy()
    # ---- returns / advantages, :498-503 ----
    returns = agent.calculate_returns(agent_all["rewards"], pt.DISCOUNT_FACTOR)
    advantages = agent.calculate_advantages(returns, agent_all["values"])
    out.update(returns=returns.numpy(), advantages=advantages.numpy(),
               returns_raw=agent.calculate_returns(agent_all["rewards"], pt.DISCOUNT_FACTOR, normalize=False).numpy())
    # ---- select_udpate, :293-346 ----
    with torch.no_grad():
        sa, sl, sv = agent.select_udpate(agent_all["states"])
    out.update(select_action=sa.numpy(), select_logp=sl.numpy(), select_value=sv.numpy())
    # ---- one inner step of update_policy, :365-417 (the optimizer steps are real; gradients stay in .grad) ----
    mse_calls = []
    real_mse = pt.F.mse_loss

    def spy_mse(a, b, *args, **kw):
        r = real_mse(a, b, *args, **kw)
        mse_calls.append(float(r.detach().sum()))
        return r

    pt.F.mse_loss = spy_mse
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            actor_loss = agent.update_policy(1, pt.PPO_CLIP, advantages, returns)
    finally:
        pt.F.mse_loss = real_mse
    out.update(update_actor_loss=np.array(actor_loss, dtype=np.float64),
               update_value_loss=np.array(mse_calls[0], dtype=np.float64))
    pick = {"actor": ["in_linear.weight", "transformer_encoder.layers.0.attention.query_projection.weight",
                      "transformer_encoder.layers.1.linear2.weight", "transformer_encoder.norm.weight",
                      "proj_pitch.weight", "proj_tempo.bias", "word_emb_chord.lut.weight"],
            "critic": ["in_linear.weight", "transformer_encoder.layers.1.attention.value_projection.weight",
                       "proj_duration.weight", "pitch_value.weight", "velocity_value.bias"]}
    for who, net in (("actor", agent.actor_net), ("critic", agent.critic_net)):
        ps = dict(net.named_parameters())
        for k in pick[who]:
            g = ps[k].grad
            out["grad.%s.%s" % (who, k)] = (g[:8] if g.numel() > 4096 else g).numpy()
        out["gradnorm." + who] = np.array([ps[k].grad.double().norm().item() if ps[k].grad is not None else -1.0
                                           for k in sorted(ps)])
        out["gradnames." + who] = np.array(sorted(ps))
    # ---- ring overwrite + seeded sampling with a buffer of 8 slots and 11 stores ----
    pt.BUFFER_SIZE = 8
    try:
        ab, eb = pt.AgentMemory(), pt.ExpertMemory()
        g2 = torch.Generator().manual_seed(1357)
        stored = {k: [] for k in ("state", "action", "logp", "value", "reward", "next", "done", "mstate", "mnext")}
        for i in range(11):
            st, nx = _tokens(g2, (W,), n_token), _tokens(g2, (W,), n_token)
            ac = _tokens(g2, (NA,), n_token)
            lp = -3 * torch.rand(NA, 6, generator=g2)
            va, rw = torch.randn(1, 1, generator=g2), torch.rand(1, 1, generator=g2)
            dn = torch.tensor(i % 2).long()
            ms, mn = (torch.rand(W, generator=g2) > 0.3).float(), (torch.rand(W, generator=g2) > 0.3).float()
            ab.store_transition(st, ac, lp, va, rw, nx, dn)
            eb.store_transition(st, ac, rw.reshape(()), nx, dn, ms, mn)
            for k, v in zip(stored, (st, ac, lp, va, rw, nx, dn, ms, mn)):
                stored[k].append(v.numpy())
        for k, v in stored.items():
            out["ring.in." + k] = np.stack(v)
        for k, v in ab.get().items():
            out["ring.agent_get." + k] = v.numpy()
        for k, v in eb.get().items():
            out["ring.expert_get." + k] = v.numpy()
        out["ring.counter"] = np.array([ab.memory_counter, eb.memory_counter])
        np.random.seed(4242)
        for i, t in enumerate(ab.sampling(5)):
            out["ring.agent_sample.%d" % i] = t.numpy()
        for i, t in enumerate(eb.sampling(5)):
            out["ring.expert_sample.%d" % i] = t.numpy()
    finally:
        pt.BUFFER_SIZE = pt.EPISODES
    np.savez_compressed(os.path.join(HERE, "ppo_rl_small.npz"), **out)
    big = {"D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8}
    config.ActorConfig.update(big)
    config.CriticConfig.update(big)
    config.DiscriConfig.update(big)


class _reference_dqn_side:
    """Context in which /root/reference/dqn_policy/{IRL_dqn_train,AIRL,AIRL_model}.py import and run on the CPU,
    unmodified.  What the files need and this image lacks exists only inside the `with` block:
      * `miditoolkit` (absent; IRL_dqn_train.py:16-17 imports it and never uses it), `wandb` (absent; `log` / `save` /
        `init` are calls whose results nothing reads) and `utils` (IRL_dqn_train.py:21 / AIRL.py:15 import three plot
        helpers from a module the reference does not contain): empty placeholder modules with no-op callables;
      * `.cuda()` (DQN.__init__, the buffers' sampling / get, RewardDiscri): there is no GPU here, so Tensor.cuda and
        Module.cuda return the object itself -- the arithmetic that follows is the reference's own, on the CPU;
      * fast_transformers -> oracle/ft_standin (as for every other fixture: the encoder BODY is the restatement),
        TrajectoryTransformer* placeholders for AIRL_model.py:10."""

    def __enter__(self):
        import types
        _transformers_placeholders()
        self.config, self.model = _import_reference("dqn_policy")
        self.saved_mods = {k: sys.modules.get(k) for k in ("wandb", "miditoolkit", "miditoolkit.midi",
                                                            "miditoolkit.midi.containers", "utils", "AIRL", "AIRL_model",
                                                            "IRL_dqn_train")}
        wb = types.ModuleType("wandb")
        wb.log = wb.save = wb.init = lambda *a, **k: None
        mt, mm, mc = (types.ModuleType(n) for n in ("miditoolkit", "miditoolkit.midi", "miditoolkit.midi.containers"))
        for n in ("Marker", "Instrument", "TempoChange", "Note"):
            setattr(mc, n, type(n, (), {}))
        mt.midi, mm.containers = mm, mc
        ut = types.ModuleType("utils")
        ut.bi_loss_plot = ut.tri_loss_plot = ut.score_plotting = lambda *a, **k: None
        sys.modules.update({"wandb": wb, "miditoolkit": mt, "miditoolkit.midi": mm, "miditoolkit.midi.containers": mc,
                            "utils": ut})
        for m in ("AIRL", "AIRL_model", "IRL_dqn_train"):
            sys.modules.pop(m, None)
        self.cuda = (torch.Tensor.cuda, torch.nn.Module.cuda)
        torch.Tensor.cuda = lambda self_, *a, **k: self_
        torch.nn.Module.cuda = lambda self_, *a, **k: self_
        self.path = os.path.join(REF, "dqn_policy")
        sys.path.insert(0, self.path)
        self.am = importlib.import_module("AIRL_model")
        self.airl = importlib.import_module("AIRL")
        self.am.D_MODEL, self.am.N_LAYER, self.am.N_HEAD = 128, 2, 2
        self.config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
        return self

    def train_module(self):
        return importlib.import_module("IRL_dqn_train")

    def __exit__(self, *exc):
        torch.Tensor.cuda, torch.nn.Module.cuda = self.cuda
        sys.path.remove(self.path)
        for k, v in self.saved_mods.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        self.config.AgentConfig.update({"D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8})
        return False


def _no_dropout(net):
    """The reference scores and trains with dropout live (`all_forward` forces train(), AIRL.py:63); a record needs it
    off: every nn.Dropout gets p = 0 and HF's Longformer self-attention its `dropout` probability 0.  BatchNorm stays as
    the reference has it (train mode: batch statistics)."""
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(getattr(m, "dropout", None), float):
            m.dropout = 0.0
    return net


def dqn_rl_small():
    """The reference's own DQN-side classes (dqn_policy/IRL_dqn_train.py:78-345, dqn_policy/AIRL.py:33-91,121-236), small
    nets (128 / 2 / 2), name-keyed fills, eval() (dropout off) so that the record is deterministic:
      choose.*   DQN.choose_action on a fixed 50-token state (rows are positions [0, 49, 48, ...]: `-0 == 0`);
      update.*   ONE DQN.update on a fixed batch of 30: MSE / CE / total loss, gradient slices of eval_net, a weight slice
                 after the Adam step, the target sync; then 51 more updates: learning rate after every update
                 (MultiStepLR [20, 40] stepped per update) and the second target sync at update 51;
      ring.*     AgentMemory / ExpertMemory with BUFFER_SIZE 8: 11 stores (ring overwrite), get(), seeded sampling;
      reward.*   RewardDiscri.update_disc(train=False) -> calculate_reward x 2 from a saved ./ckpt/disc_IRL.pt, 11 windows
                 in batches of 4 (the 3-window tail keeps the initial 1.0), dropout p = 0, BatchNorm batch statistics."""
    import contextlib
    import io
    import shutil
    import tempfile
    n_class = [56, 135, 18, 87, 18, 25]
    out = {"n_class": np.array(n_class)}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    with _reference_dqn_side() as ref:
        os.chdir(tmp)
        try:
            os.makedirs("ckpt")
            os.makedirs("exp")
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                T = ref.train_module()
                T.num = 0                                        # the module-level loop variable update() prints
                T.first_loss, T.sec_loss, T.global_loss = [], [], []
                agent = T.DQN(n_class, Pretrain=False)
            fill_params(agent.eval_net, seed=61).eval()
            fill_params(agent.target_net, seed=99).eval()        # different on purpose: update() must sync it
            gen = torch.Generator().manual_seed(1357)
            # ---- choose_action, :240-264 ----
            x = _tokens(gen, (1, 50), n_class)
            with torch.no_grad():
                action = agent.choose_action(x, x)
            out.update({"choose.x": x.numpy(), "choose.action": action.numpy()})
            # ---- update, :267-345 ----
            B = T.batch_size
            st, ns, ex = _tokens(gen, (B, 50), n_class), _tokens(gen, (B, 50), n_class), _tokens(gen, (B, 50), n_class)
            ac = _tokens(gen, (B, T.N_ACTIONS), n_class)
            rw = torch.rand(B, 1, generator=gen)
            dn = torch.randint(0, 2, (B, 1), generator=gen)
            mask = (torch.rand(B, 50, generator=gen) > 0.2).float()
            agent_tr = {"state": st, "action": ac, "reward": rw, "nextstate": ns, "done": dn}
            expert_tr = {"state": st, "action": ac, "reward": rw, "nextstate": ex, "done": dn}
            before = {k: v.clone() for k, v in agent.eval_net.state_dict().items()}
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                agent.update(agent_tr, expert_tr, mask, False, 0)
            out.update({"update.state": st.numpy(), "update.nextstate": ns.numpy(), "update.expert_next": ex.numpy(),
                        "update.action": ac.numpy(), "update.reward": rw.numpy(), "update.done": dn.numpy(),
                        "update.mask": mask.numpy(),
                        "update.losses": np.array([agent.mse_val, agent.ce_val, agent.total_val], dtype=np.float64),
                        "update.target_synced": np.array(all(torch.equal(v, before[k]) for k, v in
                                                             agent.target_net.state_dict().items()))})
            ps = dict(agent.eval_net.named_parameters())
            pick = ["in_linear.weight", "word_emb_pitch.lut.weight", "proj_chord.weight", "proj_tempo.bias",
                    "transformer_encoder.layers.0.attention.query_projection.weight",
                    "transformer_encoder.layers.0.attention.out_projection.bias",
                    "transformer_encoder.layers.1.linear1.weight", "transformer_encoder.layers.1.linear2.weight",
                    "transformer_encoder.layers.1.norm2.weight", "transformer_encoder.norm.bias"]
            for k in pick:
                g = ps[k].grad
                out["update.grad." + k] = (g[:8] if g.numel() > 4096 else g).numpy()
                w_ = ps[k].detach()
                out["update.after." + k] = (w_[:8] if w_.numel() > 4096 else w_).numpy().copy()
            names = sorted(k for k in ps if ps[k].grad is not None)
            out["update.gradnames"] = np.array(names)
            out["update.gradnorm"] = np.array([ps[k].grad.double().norm().item() for k in names])
            lrs = [float(agent.optim.param_groups[0]["lr"])]
            synced51 = None
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                for i in range(2, 53):                          # updates 2 .. 52
                    if i == 51:
                        pre = {k: v.clone() for k, v in agent.eval_net.state_dict().items()}
                    agent.update(agent_tr, expert_tr, mask, False, 0)
                    lrs.append(float(agent.optim.param_groups[0]["lr"]))
                    if i == 50:                                  # target_count 49 -> no sync at update 50
                        tgt50 = {k: v.clone() for k, v in agent.target_net.state_dict().items()}
                    if i == 51:                                  # target_count 50 -> sync BEFORE the step of update 51
                        synced51 = all(torch.equal(v, pre[k]) for k, v in agent.target_net.state_dict().items())
            out["update.lr_after"] = np.array(lrs)               # lr_after[k] = learning rate after k + 1 updates
            out["update.synced_at_51"] = np.array(bool(synced51))
            out["update.target_unchanged_2_to_50"] = np.array(all(torch.equal(v, before[k]) for k, v in tgt50.items()))
            out["update.counters"] = np.array([agent.target_count, agent.cnt_update])
            # ---- ring overwrite + seeded sampling with a buffer of 8 slots and 11 stores, :78-204 ----
            T.BUFFER_SIZE = 8
            ab, eb = T.AgentMemory(), T.ExpertMemory()
            g2 = torch.Generator().manual_seed(2468)
            stored = {k: [] for k in ("state", "action", "reward", "next", "done", "mstate", "mnext")}
            for i in range(11):
                s_, n_ = _tokens(g2, (50,), n_class), _tokens(g2, (50,), n_class)
                a_ = _tokens(g2, (T.N_ACTIONS,), n_class)
                r_ = torch.rand((), generator=g2)
                d_ = torch.tensor(i % 2).long()
                ms, mn = (torch.rand(50, generator=g2) > 0.3).float(), (torch.rand(50, generator=g2) > 0.3).float()
                ab.store_transition(s_, a_, r_, n_, d_)
                eb.store_transition(s_, a_, r_, n_, d_, ms, mn)
                for k, v in zip(stored, (s_, a_, r_, n_, d_, ms, mn)):
                    stored[k].append(v.numpy())
            for k, v in stored.items():
                out["ring.in." + k] = np.stack(v)
            for i, t in enumerate(ab.get()):
                out["ring.agent_get.%d" % i] = t.numpy()
            for i, t in enumerate(eb.get()):
                out["ring.expert_get.%d" % i] = t.numpy()
            out["ring.counter"] = np.array([ab.memory_counter, eb.memory_counter])
            np.random.seed(4242)
            for i, t in enumerate(ab.sampling(5)):
                out["ring.agent_sample.%d" % i] = t.numpy()
            for i, t in enumerate(eb.sampling(5)):
                out["ring.expert_sample.%d" % i] = t.numpy()
            # ---- RewardDiscri.update_disc(train=False) -> calculate_reward, AIRL.py:69-91,121-236 ----
            with contextlib.redirect_stdout(io.StringIO()):
                rd = ref.airl.RewardDiscri(n_class, Pretrain=False)
            fill_params(rd.disc_model, seed=41)
            with torch.no_grad():
                rd.disc_model.score_classifier[1].running_mean.copy_(torch.linspace(-0.2, 0.2, 128))
                rd.disc_model.score_classifier[1].running_var.copy_(torch.linspace(0.5, 1.5, 128))
            _no_dropout(rd.disc_model)
            torch.save({"epoch": 0, "model_state_dict": rd.disc_model.state_dict()}, rd.IRL_ckpt_path)
            rd.batch_size = 4
            g3 = torch.Generator().manual_seed(8642)
            n_win = 11
            a_states, a_next = _tokens(g3, (n_win, 50), n_class), _tokens(g3, (n_win, 50), n_class)
            e_states, e_next = _tokens(g3, (n_win, 50), n_class), _tokens(g3, (n_win, 50), n_class)
            dones = torch.zeros(n_win, 1).long()
            m_states = torch.ones(n_win, 50)
            m_states[2, 45:] = 0
            m_states[6, 30:] = 0
            m_next = torch.ones(n_win, 50)
            agent_ep = (a_states, None, None, a_next, dones)
            expert_ep = (e_states, None, None, e_next, dones, m_states, m_next)
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                traj, answer = rd.update_disc(agent_ep, expert_ep, train=False)
            out.update({"reward.agent_states": a_states.numpy(), "reward.expert_states": e_states.numpy(),
                        "reward.mask_states": m_states.numpy(), "reward.batch_size": np.array(4),
                        "reward.traj": traj.numpy(), "reward.answer": answer.numpy()})
            assert float(traj[8:].min()) == 1.0 and float(traj[:8].max()) < 1.0     # the tail keeps the initial 1.0
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, "dqn_rl_small.npz"), **out)


def dqn_loop_small():
    """The reference's own rollout loop -- the `__main__` block of dqn_policy/IRL_dqn_train.py:386-497 -- run on the CPU.
    The file is parsed, five module constants are replaced IN MEMORY (NUM_SONGS 2, BUFFER_SIZE 60, the three absolute
    paths -> a scratch directory holding a synthetic train_data_linear.npz / dictionary.pkl of the documented schema and a
    `trainloss_13.pt`-style checkpoint of a small filled net) and one call is inserted behind `Rewarder = RewardDiscri(...)`
    that puts both agent nets in eval(), switches the discriminator's dropout off and sets its scoring batch to 16
    (so that 60 buffered windows give 3 scored batches and a 12-window tail); everything else executes as written:
    two songs x 50 environment steps, and from the 61st step on (memory_counter > BUFFER_SIZE) re-scoring of both
    buffers, the overwrite of every stored reward, the two `sampling` calls and `Agent.update`.
    Recorded: every action, the rewards of every update_disc call, what every update was given and the three losses it
    printed (6 decimals), gene_reward, and both buffers at the end."""
    import ast
    import contextlib
    import io
    import pickle
    import re
    import shutil
    import tempfile
    disk = [56, 135, 18, 3, 87, 18, 25]
    n_class = [56, 135, 18, 87, 18, 25]
    keys = ["tempo", "chord", "bar-beat", "type", "pitch", "duration", "velocity"]
    songs, L, BUF = 2, 200, 60
    gen = torch.Generator().manual_seed(31337)
    x = torch.stack([torch.randint(0, n, (songs, L), generator=gen) for n in disk], -1).numpy()
    y = torch.stack([torch.randint(0, n, (songs, L), generator=gen) for n in disk], -1).numpy()
    mask = (torch.rand(songs, L, generator=gen) > 0.25).float().numpy()
    out = {"x": x.astype(np.int16), "y": y.astype(np.int16), "mask": mask, "n_class": np.array(n_class),
           "buffer_size": np.array(BUF), "score_batch": np.array(16), "np_seed": np.array(777)}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    rec = {"actions": [], "rewards": [], "upd": []}
    with _reference_dqn_side() as ref:
        os.chdir(tmp)
        try:
            os.makedirs("ckpt")
            os.makedirs("exp")
            np.savez(os.path.join(tmp, "train_data_linear.npz"), x=x, y=y, mask=mask)
            e2w = {k: {"%s_%d" % (k, i): i for i in range(n)} for k, n in zip(keys, disk)}
            w2e = {k: {i: e for e, i in v.items()} for k, v in e2w.items()}
            with open(os.path.join(tmp, "dictionary.pkl"), "wb") as f:
                pickle.dump((e2w, w2e), f)
            with contextlib.redirect_stdout(io.StringIO()):
                pre = fill_params(ref.model.LinearTransformer(n_class), seed=61)
                disc = fill_params(ref.am.LongFormer(n_class), seed=41)
            torch.save({"epoch": 0, "model_state_dict": pre.state_dict()}, os.path.join(tmp, "pretrain.pt"))
            with torch.no_grad():
                disc.score_classifier[1].running_mean.copy_(torch.linspace(-0.2, 0.2, 128))
                disc.score_classifier[1].running_var.copy_(torch.linspace(0.5, 1.5, 128))
            torch.save({"epoch": 0, "model_state_dict": disc.state_dict()}, "./ckpt/disc_IRL.pt")

            def hook(Agent, Rewarder):
                Agent.eval_net.eval()
                Agent.target_net.eval()
                _no_dropout(Rewarder.disc_model)
                Rewarder.batch_size = 16
                choose, update, upd_disc = Agent.choose_action, Agent.update, Rewarder.update_disc

                def choose_rec(x_, target):
                    a = choose(x_, target)
                    rec["actions"].append(a.numpy().copy())
                    return a

                def update_rec(agent_transition, expert_transition, mask_next_states, update_flag, epoch):
                    rec["upd"].append({"state": agent_transition["state"].numpy().copy(),
                                       "action": agent_transition["action"].numpy().copy(),
                                       "reward": agent_transition["reward"].numpy().copy(),
                                       "nextstate": agent_transition["nextstate"].numpy().copy(),
                                       "done": agent_transition["done"].numpy().copy(),
                                       "e_nextstate": expert_transition["nextstate"].numpy().copy(),
                                       "e_done": expert_transition["done"].numpy().copy(),
                                       "mask": mask_next_states.numpy().copy(), "flag": bool(update_flag), "epoch": epoch,
                                       "lr": float(Agent.optim.param_groups[0]["lr"])})
                    return update(agent_transition, expert_transition, mask_next_states, update_flag, epoch)

                def upd_disc_rec(agent_traj, expert_traj, train=True):
                    r = upd_disc(agent_traj, expert_traj, train=train)
                    rec["rewards"].append((r[0].numpy().copy(), r[1].numpy().copy(), bool(train)))
                    return r

                Agent.choose_action, Agent.update, Rewarder.update_disc = choose_rec, update_rec, upd_disc_rec

            src = open(os.path.join(REF, "dqn_policy", "IRL_dqn_train.py")).read()
            tree = ast.parse(src)
            consts = {"NUM_SONGS": songs, "BUFFER_SIZE": BUF, "path_train_data": os.path.join(tmp, "train_data_linear.npz"),
                      "path_dictionary": os.path.join(tmp, "dictionary.pkl"),
                      "Pretrain_ckpt": os.path.join(tmp, "pretrain.pt")}
            seen = set()
            for node in tree.body:
                if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                        and node.targets[0].id in consts:
                    node.value = ast.Constant(consts[node.targets[0].id])
                    seen.add(node.targets[0].id)
            assert seen == set(consts), seen
            main_if = [n for n in tree.body if isinstance(n, ast.If)][-1]
            at = [i for i, n in enumerate(main_if.body) if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name)
                  and n.targets[0].id == "Rewarder"]
            assert len(at) == 1
            call = ast.Expr(ast.Call(ast.Name("_golden_hook", ast.Load()),
                                     [ast.Name("Agent", ast.Load()), ast.Name("Rewarder", ast.Load())], []))
            main_if.body.insert(at[0] + 1, call)
            ast.fix_missing_locations(tree)
            code = compile(tree, "<reference dqn_policy/IRL_dqn_train.py, constants patched in memory>", "exec")
            ns = {"__name__": "__main__", "_golden_hook": hook}
            np.random.seed(777)
            log = io.StringIO()
            with contextlib.redirect_stdout(log), contextlib.redirect_stderr(io.StringIO()):
                exec(code, ns)
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmp, ignore_errors=True)
    losses = [[float(v) for v in m] for m in
              re.findall(r"MSE_Loss: ([-\d.naninf]+)\| CE_Loss: ([-\d.naninf]+)\| TD_Loss: ([-\d.naninf]+)", log.getvalue())]
    n_upd = len(rec["upd"])
    assert len(rec["actions"]) == songs * 50 and n_upd == songs * 50 - BUF and len(losses) == n_upd == len(rec["rewards"])
    out["actions"] = np.stack(rec["actions"]).astype(np.int16)
    out["traj_reward"] = np.stack([r[0] for r in rec["rewards"]])[:, :, 0]
    out["answer_reward"] = np.stack([r[1] for r in rec["rewards"]])[:, :, 0]
    out["update.losses"] = np.array(losses, dtype=np.float64)
    out["update.lr_before"] = np.array([u["lr"] for u in rec["upd"]])
    out["update.epoch"] = np.array([u["epoch"] for u in rec["upd"]])
    for k in ("reward", "done", "e_done"):
        out["update." + k] = np.stack([u[k] for u in rec["upd"]])
    for k in ("state", "action", "nextstate", "e_nextstate", "mask"):
        out["update3." + k] = np.stack([u[k] for u in rec["upd"][:3]]).astype(np.int16 if k != "mask" else np.float32)
        out["update.sum." + k] = np.array([u[k].astype(np.float64).sum() for u in rec["upd"]])
    ab, eb = ns["AgentBuffer"], ns["ExpertBuffer"]
    out["gene_reward"] = np.array(ns["gene_reward"], dtype=np.float64)
    out["final.agent_states"] = ab.states_agent.astype(np.int16)
    out["final.agent_actions"] = ab.actions_agent.astype(np.int16)
    out["final.agent_rewards"] = ab.rewards_agent.astype(np.float32)
    out["final.agent_next"] = ab.next_states_agent.astype(np.int16)
    out["final.expert_states"] = eb.states_exp.astype(np.int16)
    out["final.expert_next"] = eb.next_states_exp.astype(np.int16)
    out["final.expert_rewards"] = eb.rewards_exp.astype(np.float32)
    out["final.mask_state"] = eb.mask_state.numpy()
    out["final.mask_next_state"] = eb.mask_next_state.numpy()
    out["final.counters"] = np.array([ab.memory_counter, eb.memory_counter])
    np.savez_compressed(os.path.join(HERE, "dqn_loop_small.npz"), **out)


def ppo_reward_grads_small():
    """Gradients THROUGH the PPO reward model (ppo_policy/model.py:459-495 `LongFormer.token_forward`, the function
    `my_pretrain.py --reward_pretrain` would train): d(sum of scores * w) / d(every parameter), reference's own class
    + HF Longformer backward, eval mode (dropout off).  Stored like airl_grads_small: norms of all gradients, the
    gradients themselves (first 8 rows of large tensors)."""
    config, model = _import_reference("ppo_policy")
    config.DiscriConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    n_token = [49, 19, 19, 89, 67, 25]
    net = fill_params(model.LongFormer(n_token), seed=33).eval()
    gen = torch.Generator().manual_seed(80)
    x = _tokens(gen, (3, 50), n_token)
    mask = torch.ones(3, 50, dtype=torch.long)
    mask[2, 41:] = 0
    w = torch.tensor([[1.0], [-2.0], [0.5]])
    score = net.token_forward(x, None, mask)
    (score * w).sum().backward()
    out = {"x": x.numpy(), "mask": mask.numpy(), "w": w.numpy(), "score": score.detach().numpy(),
           "n_token": np.array(n_token)}
    names, norms = [], []
    for k, p_ in net.named_parameters():
        if p_.grad is None:
            continue
        g = p_.grad.detach()
        names.append(k)
        norms.append(g.double().norm().item())
        out["grad." + k] = (g[:8] if g.numel() > 4096 else g).numpy()
    out["names"] = np.array(names)
    out["norms"] = np.array(norms)
    np.savez_compressed(os.path.join(HERE, "ppo_reward_grads_small.npz"), **out)
    config.DiscriConfig.update({"D_MODEL": 512, "N_LAYER": 12, "N_HEAD": 8})


def ppo_dataset_files():
    """The PPO pipeline's on-disk files written by the REFERENCE's own writers (data, not source):
      tests/golden/ppo_dataset/dictionary.pickle   <- ppo_policy/prepare_data.py::construct_dict  (:239-302)
      tests/golden/ppo_dataset/worded_data.pickle   a hand-made input: 4 word sequences of lengths 1300/700/1200/90
      tests/golden/ppo_dataset/our_dataset.pickle  <- ppo_policy/preprocess.py::process_data      (:10-72), np seed 5
    prepare_data.py does `import miditoolkit` (absent here) at module level and never touches it inside construct_dict:
    it gets an empty placeholder module inside this generator only.  process_data reads ./dataset/* relative to the
    working directory, so it runs in a scratch directory."""
    import pickle
    import shutil
    import tempfile
    import types
    out_dir = os.path.join(HERE, "ppo_dataset")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(REF, "ppo_policy")
    for m in ("miditoolkit",):
        sys.modules.setdefault(m, types.ModuleType(m))
    for m in ("prepare_data", "preprocess", "utils", "chord_recognition", "config"):
        sys.modules.pop(m, None)
    sys.path.insert(0, path)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    try:
        import contextlib
        import io
        prepare_data = importlib.import_module("prepare_data")
        os.chdir(tmp)
        os.makedirs("dataset")
        with contextlib.redirect_stdout(io.StringIO()):
            prepare_data.construct_dict(os.path.join("dataset", "dictionary.pickle"))
        with open(os.path.join("dataset", "dictionary.pickle"), "rb") as f:
            event2word, _ = pickle.load(f)
        n_tok = [len(event2word[k]) for k in event2word]
        rng = np.random.default_rng(17)
        worded = [[[int(rng.integers(0, n - 3)) for n in n_tok] for _ in range(L)] for L in (1300, 700, 1200, 90)]
        with open(os.path.join("dataset", "worded_data.pickle"), "wb") as f:
            pickle.dump(worded, f)
        shutil.copy(os.path.join("dataset", "worded_data.pickle"), os.path.join(out_dir, "worded_data.pickle"))
        preprocess = importlib.import_module("preprocess")
        np.random.seed(5)
        with contextlib.redirect_stdout(io.StringIO()):
            preprocess.process_data()
        for name in ("dictionary.pickle", "our_dataset.pickle"):
            shutil.copy(os.path.join("dataset", name), os.path.join(out_dir, name))
    finally:
        os.chdir(cwd)
        sys.path.remove(path)
        shutil.rmtree(tmp, ignore_errors=True)
        sys.modules.pop("config", None)


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1:                      # python make_golden.py <fixture function> ...: only those
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    dqn_small()
    dqn_repo_dims()
    ppo_small()
    ppo_reward_small()
    airl_small()
    airl_grads_small()
    dqn_generation_small()
    ppo_rl_small()
    dqn_rl_small()
    dqn_loop_small()
    ppo_reward_grads_small()
    ppo_dataset_files()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
