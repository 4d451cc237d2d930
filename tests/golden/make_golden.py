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


if __name__ == "__main__":
    torch.set_num_threads(4)
    dqn_small()
    dqn_repo_dims()
    ppo_small()
    ppo_reward_small()
    airl_small()
    airl_grads_small()
    dqn_generation_small()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
