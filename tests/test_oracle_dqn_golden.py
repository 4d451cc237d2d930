"""CPU: oracle/rl_math.py, oracle/dqn_loop.py and oracle/discriminator.py against tests/golden/dqn_rl_small.npz, which
was recorded from the REFERENCE's own DQN-side classes -- `DQN`, `AgentMemory`, `ExpertMemory`
(dqn_policy/IRL_dqn_train.py:78-345) and `RewardDiscri` (dqn_policy/AIRL.py:33-91,121-236), imported unmodified by
tests/golden/make_golden.py::dqn_rl_small (placeholders for wandb / miditoolkit / the missing `utils`, `.cuda()` a no-op
on the CPU).  After this test the DQN restatement no longer rests on a reading of the reference: A9, A10, A11 (DQN side)
and A13 of SURVEY section 8 are pinned by reference-generated vectors (the encoder body under them stays the
restatement of fast_transformers 0.4.0: DESIGN.md section 5)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

from oracle import cw_model, discriminator as odisc, dqn_loop, rl_math  # noqa: E402

FX = np.load(os.path.join(HERE, "golden", "dqn_rl_small.npz"), allow_pickle=False)
N_CLASS = FX["n_class"].tolist()
GAMMA, NA = 0.95, 25                      # IRL_dqn_train.py:48,57


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _net(seed=61):
    return fill_params(cw_model.CWLinearTransformer(N_CLASS, 128, 2, 2, variant="dqn"), seed=seed).eval()


def test_choose_action_rows_are_positions_0_then_49_downwards():
    net = _net()
    x = _t(FX["choose.x"])
    with torch.no_grad():
        logits = net.forward_output(net.forward_hidden(x))
        action = rl_math.dqn_choose_action(logits, NA)
        ids = torch.stack([torch.argmax(torch.softmax(y, -1), -1)[0] for y in logits], -1)       # (50, 6)
    assert torch.equal(action, _t(FX["choose.action"]))
    assert torch.equal(_t(FX["choose.action"])[0], ids[0])                 # `-0 == 0`: the first row is position 0
    assert torch.equal(_t(FX["choose.action"])[1:], ids.flip(0)[:NA - 1])  # then 49, 48, ..., 26


def test_one_update_losses_gradients_adam_step_and_schedule():
    net = _net()
    st, ns, ex = _t(FX["update.state"]), _t(FX["update.nextstate"]), _t(FX["update.expert_next"])
    ac, rw, dn, mask = _t(FX["update.action"]), _t(FX["update.reward"]), _t(FX["update.done"]), _t(FX["update.mask"])
    assert bool(FX["update.target_synced"])                     # target_count 0: the target net is the eval net
    y = net.forward_output(net.forward_hidden(st))
    yt = net.forward_output(net.forward_hidden(ns))             # the reference leaves the target pass under autograd
    mse, _ = rl_math.dqn_td_loss(y, yt, ac, rw, dn, GAMMA, NA)
    ce = sum(net.train_step(st, ex, mask)) / 6
    total = 0.3 * mse + 0.7 * ce
    want = FX["update.losses"]
    for got, w in zip((mse, ce, total), want):
        assert abs(got.item() - w) <= 2e-5 * max(1.0, abs(w)), (got.item(), w)
    opt = torch.optim.Adam(net.parameters(), lr=0.01)
    opt.zero_grad()
    total.backward()
    ps = dict(net.named_parameters())
    # the target pass is a second use of the same weights in the restatement (yt depends on them); in the reference it
    # runs on target_net, whose gradients nobody reads: detach it for the gradient comparison
    net.zero_grad()
    yt_d = [t.detach() for t in yt]
    mse_d, _ = rl_math.dqn_td_loss(net.forward_output(net.forward_hidden(st)), yt_d, ac, rw, dn, GAMMA, NA)
    (0.3 * mse_d + 0.7 * sum(net.train_step(st, ex, mask)) / 6).backward()
    names = FX["update.gradnames"].tolist()
    norms = dict(zip(names, FX["update.gradnorm"].tolist()))
    for k in names:
        g = ps[k].grad
        assert g is not None, k
        assert abs(g.double().norm().item() - norms[k]) <= 1e-4 * max(norms[k], 1e-6), k
    for key in FX.files:
        if key.startswith("update.grad."):
            k = key[len("update.grad."):]
            g = ps[k].grad
            g = g[:8] if g.numel() > 4096 else g
            w = _t(FX[key])
            assert (g - w).abs().max().item() <= 1e-4 * max(1e-4, w.abs().max().item()), k
    opt.step()
    for key in FX.files:
        if key.startswith("update.after."):
            k = key[len("update.after."):]
            v = ps[k].detach()
            v = v[:8] if v.numel() > 4096 else v
            # Adam's first step is lr * g / (|g| + 1e-8): where |g| is not far above that epsilon the step amplifies
            # the last bits of g -- compare where the gradient is well-defined
            solid = _t(FX["update.grad." + k]).abs() > 1e-6
            assert solid.float().mean().item() > 0.05, k
            assert (v - _t(FX[key]))[solid].abs().max().item() <= 2e-5, k
    # MultiStepLR([20, 40], 0.1) stepped once per UPDATE (:344-345); the target net is re-synced every 50 updates
    lr = FX["update.lr_after"]
    assert np.allclose(lr[:19], 1e-2) and np.allclose(lr[19:39], 1e-3) and np.allclose(lr[39:], 1e-4) and len(lr) == 52
    assert bool(FX["update.target_unchanged_2_to_50"]) and bool(FX["update.synced_at_51"])
    assert FX["update.counters"].tolist() == [52, 52]


def test_ring_buffers_get_and_seeded_sampling():
    ab, eb = dqn_loop.RefAgentMemory(8), dqn_loop.RefExpertMemory(8)
    for i in range(11):
        s_, a_, n_ = _t(FX["ring.in.state"][i]), _t(FX["ring.in.action"][i]), _t(FX["ring.in.next"][i])
        r_, d_ = _t(FX["ring.in.reward"][i]), _t(FX["ring.in.done"][i])
        ab.store_transition(s_, a_, r_, n_, d_)
        eb.store_transition(s_, a_, r_, n_, d_, _t(FX["ring.in.mstate"][i]), _t(FX["ring.in.mnext"][i]))
    assert [ab.memory_counter, eb.memory_counter] == FX["ring.counter"].tolist()
    for i, t in enumerate(ab.get()):
        assert np.array_equal(t.numpy(), FX["ring.agent_get.%d" % i]) and str(t.numpy().dtype) == str(FX["ring.agent_get.%d" % i].dtype), i
    for i, t in enumerate(eb.get()):
        assert np.array_equal(t.numpy(), FX["ring.expert_get.%d" % i]) and str(t.numpy().dtype) == str(FX["ring.expert_get.%d" % i].dtype), i
    # slots 0..2 were overwritten by stores 8..10
    assert np.array_equal(FX["ring.agent_get.0"][:3], FX["ring.in.state"][8:11])
    np.random.seed(4242)
    for i, t in enumerate(ab.sampling(5)):
        assert np.array_equal(t.numpy(), FX["ring.agent_sample.%d" % i]), i
    for i, t in enumerate(eb.sampling(5)):
        assert np.array_equal(t.numpy(), FX["ring.expert_sample.%d" % i]), i


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


def test_calculate_reward_scores_whole_batches_and_the_tail_keeps_one():
    sd = _disc_sd()
    bs = int(FX["reward.batch_size"])
    mask = _t(FX["reward.mask_states"])
    with torch.no_grad():
        traj = odisc.calculate_reward(sd, _t(FX["reward.agent_states"]), mask, bs, 2, 2, 50)
        answer = odisc.calculate_reward(sd, _t(FX["reward.expert_states"]), mask, bs, 2, 2, 50)
    assert (traj - _t(FX["reward.traj"])).abs().max().item() < 1e-5
    assert (answer - _t(FX["reward.answer"])).abs().max().item() < 1e-5
    assert FX["reward.traj"][8:].tolist() == [[1.0]] * 3 and FX["reward.traj"][:8].max() < 1.0
    # the BatchNorm statistics are those of each batch of 4: scoring the first 8 windows as ONE batch gives other values
    with torch.no_grad():
        one = odisc.airl_forward(sd, _t(FX["reward.agent_states"])[:8], mask[:8].long(), 2, 2, 50, batch_stats=True)
    assert (one - _t(FX["reward.traj"])[:8]).abs().max().item() > 1e-4
