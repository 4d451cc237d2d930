"""GPU: the product model classes (drop-in model.py surfaces on libcwlt kernels) against the golden
fixtures recorded from the REFERENCE's own classes, and against the CPU oracle on fresh inputs."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401
from oracle import cw_model  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north_star: logits within 1e-4 fp32


def _load(name):
    return np.load(os.path.join(HERE, "golden", name), allow_pickle=False)


def _t(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def _err(a, b):
    return (a.detach().double().cpu() - torch.as_tensor(np.asarray(b)).double()).abs().max().item()


def _dqn_model(dims, n_class, seed, cuda):
    from rlmg_amd.dqn_policy import config, model
    old = dict(config.AgentConfig)
    config.AgentConfig.update({"D_MODEL": dims[0], "N_LAYER": dims[1], "N_HEAD": dims[2]})
    try:
        net = model.LinearTransformer(n_class)
    finally:
        config.AgentConfig.update(old)
    return fill_params(net, seed=seed).to(cuda).eval()


def _ppo_models(dims, n_token, cuda):
    from rlmg_amd.ppo_policy import config, model
    old = dict(config.ActorConfig)
    config.ActorConfig.update({"D_MODEL": dims[0], "N_LAYER": dims[1], "N_HEAD": dims[2]})
    try:
        actor, critic = model.Actor_Transformer(n_token), model.Critic_Transformer(n_token)
    finally:
        config.ActorConfig.update(old)
    return fill_params(actor, seed=21).to(cuda).eval(), fill_params(critic, seed=22).to(cuda).eval()


def test_dqn_small_matches_reference_fixture(cuda):
    fx = _load("dqn_small.npz")
    net = _dqn_model((128, 2, 2), fx["n_class"].tolist(), 11, cuda)
    assert sorted(net.state_dict().keys()) == fx["keys"].tolist()
    x, y, mask = _t(fx["x"], cuda), _t(fx["y"], cuda), _t(fx["mask"], cuda)
    h = net.forward_hidden(x)
    assert _err(h, fx["h"]) < TOL
    for i, l in enumerate(net.forward_output(h, y)):
        assert _err(l, fx["logits%d" % i]) < TOL
    losses = net.train_step(x, y, mask)
    assert np.allclose([l.item() for l in losses], fx["losses"], rtol=0, atol=TOL)
    (sum(losses) / 6).backward()
    enc = net.transformer_encoder
    got = {
        "grad.in_linear.weight": net.in_linear.weight.grad[:16, ::19],
        "grad.q0": enc.layers[0].attention.query_projection.weight.grad[::8, ::8],
        "grad.k1": enc.layers[1].attention.key_projection.weight.grad[::8, ::8],
        "grad.lin1": enc.layers[1].linear1.weight.grad[::64, ::8],
        "grad.norm1": enc.layers[0].norm1.weight.grad,
        "grad.emb_pitch": net.word_emb_pitch.lut.weight.grad[:, ::16],
        "grad.proj_chord.bias": net.proj_chord.bias.grad,
    }
    for k, v in got.items():
        assert _err(v, fx[k]) < TOL, k


def test_dqn_repo_dims_matches_reference_fixture(cuda):
    fx = _load("dqn_repo_dims.npz")
    net = _dqn_model((512, 12, 8), [56, 135, 18, 87, 18, 25], 12, cuda)
    assert sum(p.numel() for p in net.parameters() if p.requires_grad) == int(fx["n_params"])
    assert sorted(net.state_dict().keys()) == fx["keys"].tolist()
    with torch.no_grad():
        h = net.forward_hidden(_t(fx["x"], cuda))
        assert _err(h, fx["h"]) < TOL
        for i, l in enumerate(net.forward_output(h, None)):
            assert _err(l, fx["logits%d" % i]) < TOL


def test_ppo_small_matches_reference_fixture(cuda):
    fx = _load("ppo_small.npz")
    actor, critic = _ppo_models((128, 2, 2), fx["n_token"].tolist(), cuda)
    assert sorted(actor.state_dict().keys()) == fx["actor_keys"].tolist()
    assert sorted(critic.state_dict().keys()) == fx["critic_keys"].tolist()
    x, y, mask = _t(fx["x"], cuda), _t(fx["y"], cuda), _t(fx["mask"], cuda)
    with torch.no_grad():
        h = actor.forward_hidden(x)
        assert _err(h, fx["h"]) < TOL
        for i, l in enumerate(actor.forward_output(h)):
            assert _err(l, fx["logits%d" % i]) < TOL
        assert _err(actor.value_funtion(h[0]), fx["value_funtion"]) < TOL
        losses = actor.train_step(x, y, mask)           # int64 mask, as ppo_train.py passes it
        assert np.allclose([l.item() for l in losses], fx["losses"], rtol=0, atol=TOL)
        assert _err(critic.value_produce(x), fx["critic_value"]) < TOL


def test_repo_dims_train_step_grads_match_oracle(cuda):
    """Full 12-layer fwd+bwd at repo dims vs the CPU oracle on the same seeded inputs (B=2, T=96)."""
    n_class = [56, 135, 18, 87, 18, 25]
    net = _dqn_model((512, 12, 8), n_class, 31, cuda)
    ref = fill_params(cw_model.CWLinearTransformer(n_class, 512, 12, 8, variant="dqn"), seed=31).eval()
    g = torch.Generator().manual_seed(8)
    x = torch.stack([torch.randint(0, n, (2, 96), generator=g) for n in n_class], -1)
    y = torch.stack([torch.randint(0, n, (2, 96), generator=g) for n in n_class], -1)
    mask = torch.ones(2, 96)
    mask[1, 80:] = 0
    lr = ref.train_step(x, y, mask)
    (sum(lr) / 6).backward()
    lg = net.train_step(x.to(cuda), y.to(cuda), mask.to(cuda))
    (sum(lg) / 6).backward()
    assert np.allclose([l.item() for l in lg], [l.item() for l in lr], rtol=0, atol=TOL)
    pr = dict(ref.named_parameters())
    worst = 0.0
    for name, p in net.named_parameters():
        if name.startswith("project_concat_type"):
            assert p.grad is None
            continue
        e = _err(p.grad, pr[name].grad.numpy())
        scale = max(1e-3, pr[name].grad.abs().max().item())
        worst = max(worst, e / scale)
        assert e <= 2e-3 * scale + 1e-6, (name, e, scale)
    print("worst relative grad error", worst)


def test_greedy_ids_bit_exact_vs_oracle(cuda):
    """Greedy (softmax -> argmax) token ids from the GPU path equal the CPU oracle's."""
    from rlmg_amd import ops
    n_class = [56, 135, 18, 87, 18, 25]
    net = _dqn_model((512, 12, 8), n_class, 41, cuda)
    ref = fill_params(cw_model.CWLinearTransformer(n_class, 512, 12, 8, variant="dqn"), seed=41).eval()
    g = torch.Generator().manual_seed(9)
    x = torch.stack([torch.randint(0, n, (4, 50), generator=g) for n in n_class], -1)
    with torch.no_grad():
        ys = ref.forward_output(ref.forward_hidden(x))
        ids_ref = torch.stack([torch.softmax(t, -1).argmax(-1) for t in ys], -1)
        logits = net.fused_logits(net.forward_hidden(x.to(cuda)))
        ids = ops.heads_forward(logits, n_class, want_argmax=True)["argmax"].view(4, 50, 6)
    assert torch.equal(ids.cpu(), ids_ref)


# bf16 mode against the fp32 oracle ---------------------------------------------------------------------------------
# Error model for the bound below.  In bf16 mode every activation tensor that crosses HBM is rounded to 8 significant
# bits: relative error uniform in +-2^-9, rms u = 2^-9 / sqrt(3) = 1.13e-3 per rounding.  One encoder layer stores 10
# such tensors on the forward path (qkv, attention out, out-proj, s1, x1, h, g, ffn out, s2, layer out) and about as
# many on the backward path; 12 layers + embedding / in_linear / positional encoding / final norm / logits give
# n ~ 2 * (12 * 10 + 5) = 250 roundings between the tokens and a parameter gradient, plus the bf16 rounding of the
# GEMM weights (same u, once per use).  Independent relative perturbations of rms u accumulate as a random walk:
# expected norm-wise relative error of a gradient tensor ~ sqrt(n) * u = sqrt(250) * 1.13e-3 = 1.8e-2 (first order:
# post-LN renormalises every layer, so the perturbations do not grow geometrically).  The test allows 4x that in
# the L2 norm per tensor -- BF16_GRAD_REL = 7e-2 -- relative to the tensor's OWN norm, with no additive term (round 2
# added the model-wide rms norm; measured errors are 1-5 % of the own norm for every tensor, so it was never needed).
BF16_GRAD_REL = 4 * (2 * (12 * 10 + 5)) ** 0.5 * 2.0 ** -9 / 3 ** 0.5


_ORACLE_CACHE = {}


@pytest.mark.parametrize("schedule", ["library", "bench"])
def test_repo_dims_bf16_train_step_grads_match_fp32_oracle(cuda, monkeypatch, schedule):
    """BASELINE configs[1]'s own shapes in the benched dtype: repo dims, T = 1024, B = 2, bf16 storage; losses and
    EVERY parameter gradient against the fp32 CPU oracle on the same tokens (dropout off so both are deterministic).
    "library": what the library picks for 16 (sequence, head) streams -- segmented scans, the dkdv + dq backward pair;
    "bench": whole-sequence scans as at B = 512 -- the forward's final-state hand-over and the one-sweep backward.
    Both run the one-kernel FFN forward and the fused FFN backward."""
    if schedule == "bench":
        monkeypatch.setenv("CWLT_SCAN_SEGMENTS", "1")
    n_class = [56, 135, 18, 87, 18, 25]
    net = _dqn_model((512, 12, 8), n_class, 51, cuda)
    B, T = 2, 1024
    g = torch.Generator().manual_seed(10)
    x = torch.stack([torch.randint(0, n, (B, T), generator=g) for n in n_class], -1)
    y = torch.stack([torch.randint(0, n, (B, T), generator=g) for n in n_class], -1)
    mask = torch.ones(B, T)
    mask[1, 900:] = 0
    if "ref" not in _ORACLE_CACHE:                       # the fp32 oracle pass is the slow part: once for both schedules
        ref = fill_params(cw_model.CWLinearTransformer(n_class, 512, 12, 8, variant="dqn"), seed=51).eval()
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
        lr = ref.train_step(x, y, mask)
        (sum(lr) / 6).backward()
        _ORACLE_CACHE["ref"] = (ref, [l.detach() for l in lr])
    ref, lr = _ORACLE_CACHE["ref"]
    net.compute_dtype = torch.bfloat16
    lg = net.train_step(x.to(cuda), y.to(cuda), mask.to(cuda))
    (sum(lg) / 6).backward()
    l16, l32 = np.array([l.item() for l in lg]), np.array([l.item() for l in lr])
    # a loss is a mean over 2 000 tokens of nll values that each carry the logits' relative error
    assert np.abs(l16 - l32).max() <= BF16_GRAD_REL * np.abs(l32).max(), (l16, l32)
    pr = dict(ref.named_parameters())
    norms = {n_: pr[n_].grad.double().norm().item() for n_, p in net.named_parameters() if p.grad is not None}
    rms_norm = (sum(v * v for v in norms.values()) / len(norms)) ** 0.5
    rows, worst = [], 0.0
    for name, p in net.named_parameters():
        if name.startswith("project_concat_type"):
            assert p.grad is None
            continue
        assert p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), name
        d = (p.grad.detach().double().cpu() - pr[name].grad.double()).norm().item()
        rel = d / max(norms[name], 1e-30)
        cos = torch.nn.functional.cosine_similarity(p.grad.detach().double().cpu().flatten(),
                                                    pr[name].grad.double().flatten(), dim=0).item()
        rows.append((name, norms[name], d, rel, cos))
        # The derived bound is relative to the tensor's OWN norm -- for every tensor of this model, down to the
        # key-projection biases whose true gradient nearly cancels (own norm 5e-6 against an rms tensor norm of 5e-2).  The
        # floor of 1e-4 of the rms norm only keeps a gradient that is exactly zero in the oracle from dividing by nothing.
        floor = 1e-4 * rms_norm
        bound = BF16_GRAD_REL * max(norms[name], floor)
        worst = max(worst, d / max(norms[name], floor))
        assert d <= bound, (name, d, norms[name], rms_norm)
    out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "bf16_grad_errors_%s.txt" % schedule), "w") as f:
            f.write("bound %.4f  rms tensor norm %.4e  losses bf16 %s fp32 %s\n" % (BF16_GRAD_REL, rms_norm, l16, l32))
            for r in rows:
                f.write("%-70s |g| %.4e  |d| %.4e  rel %.4e  cos %.6f\n" % r)
    print("worst bf16 gradient error / own norm: %.4f (bound %.4f)" % (worst, BF16_GRAD_REL))


def test_one_kernel_ffn_forward_equals_gemm_plus_activation_in_the_model(cuda, monkeypatch):
    """Training mode, dropout ON, bf16: the same seeds give the same masks whether linear1 + bias + GELU + dropout run as
    ONE kernel (cwlt_gemm_nt_bias_gelu_dropout) or as hipBLASLt GEMM + cwlt_bias_gelu_dropout_fwd; losses and every
    parameter gradient agree to what one differing bf16 rounding of the pre-activation per layer can do."""
    from rlmg_amd import encoder
    n_class = [56, 135, 18, 87, 18, 25]
    B, T = 2, 256
    g = torch.Generator().manual_seed(12)
    x = torch.stack([torch.randint(0, n, (B, T), generator=g) for n in n_class], -1).to(cuda)
    y = torch.stack([torch.randint(0, n, (B, T), generator=g) for n in n_class], -1).to(cuda)
    mask = torch.ones(B, T, device=cuda)
    runs = {}
    for fused in (True, False):
        monkeypatch.setattr(encoder, "FUSED_FFN_FWD", fused)
        net = _dqn_model((512, 4, 8), n_class, 53, cuda).train()
        net.compute_dtype = torch.bfloat16
        torch.manual_seed(77)                                   # ops.next_seed() draws from torch's CPU generator
        losses = net.train_step(x, y, mask)
        (sum(losses) / 6).backward()
        runs[fused] = ([l.item() for l in losses], {n_: p.grad.detach().double().cpu() for n_, p in net.named_parameters()
                                                   if p.grad is not None})
    la, lb = np.array(runs[True][0]), np.array(runs[False][0])
    assert np.abs(la - lb).max() <= 2e-3 * np.abs(lb).max(), (la, lb)
    ga, gb = runs[True][1], runs[False][1]
    assert ga.keys() == gb.keys()
    rms = (sum(v.norm().item() ** 2 for v in gb.values()) / len(gb)) ** 0.5
    for k in gb:
        d = (ga[k] - gb[k]).norm().item()
        assert d <= 2e-2 * (gb[k].norm().item() + rms), (k, d, gb[k].norm().item())
