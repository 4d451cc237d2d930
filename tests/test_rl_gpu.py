"""GPU: RL arithmetic kernels (through the C-ABI) against the literal CPU restatement of the reference
(oracle/rl_math.py), quirks included."""
import pytest
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import ops, rl_ops
from oracle import rl_math

pytestmark = pytest.mark.gpu
N_CLASS = (56, 135, 18, 87, 18, 25)


def _logits(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, T, n, generator=g) * 2 for n in N_CLASS]


def _fused(ys, cuda):
    W = sum(N_CLASS) + (-sum(N_CLASS)) % 64
    B, T = ys[0].shape[:2]
    out = torch.zeros(B, T, W)
    o = 0
    for y in ys:
        out[..., o:o + y.shape[-1]] = y
        o += y.shape[-1]
    return out.to(cuda)


def _heads(fused):
    B, T, W = fused.shape
    res = ops.heads_forward(fused.view(B * T, W), N_CLASS, want_argmax=True, want_probs=True)
    return res["argmax"].view(B, T, 6), res["probs"].view(B, T, -1)


def test_dqn_choose_action_rows_bit_exact(cuda):
    ys = _logits(1, 50, 1)
    want = rl_math.dqn_choose_action(ys, 25)
    ids, _ = _heads(_fused(ys, cuda))
    got, _ = rl_ops.rollout_gather(ids, None, N_CLASS, 25, mode=0)
    assert torch.equal(got[0].cpu(), want)
    assert torch.equal(got[0, 0].cpu(), ids[0, 0].cpu()) and torch.equal(got[0, 1].cpu(), ids[0, 49].cpu())  # -0 == 0


def test_ppo_choose_action_and_logp_quirk(cuda):
    ys = _logits(1, 50, 2)
    want_a, want_lp = rl_math.ppo_choose_action(ys, 25)
    ids, probs = _heads(_fused(ys, cuda))
    got_a, got_lp = rl_ops.rollout_gather(ids, probs, N_CLASS, 25, mode=1)
    assert torch.equal(got_a[0].cpu(), want_a)
    assert (got_lp[0].cpu() - want_lp).abs().max().item() < 1e-4


def test_ppo_select_update_last_batch_element(cuda):
    ys = _logits(4, 50, 3)
    want_a, want_lp = rl_math.ppo_select_update(ys, 25)
    ids, probs = _heads(_fused(ys, cuda))
    got_a, got_lp = rl_ops.rollout_gather(ids, probs, N_CLASS, 25, mode=2)
    assert torch.equal(got_a[-1].cpu(), want_a)
    assert (got_lp[-1].cpu() - want_lp).abs().max().item() < 1e-4


def test_logp_argmax_gradient(cuda):
    g = torch.Generator().manual_seed(4)
    W = 384
    x = torch.randn(25, W, generator=g)
    up = torch.randn(25, 6, generator=g)
    xr = x.double().requires_grad_(True)
    lps, o = [], 0
    for n in N_CLASS:
        ls = torch.log_softmax(xr[:, o:o + n], -1)
        lps.append(ls.max(-1).values)
        o += n
    ref = torch.stack(lps, -1)
    (ref * up.double()).sum().backward()
    xd = x.to(cuda).requires_grad_(True)
    lp, ids = rl_ops.logp_argmax(xd, N_CLASS)
    assert (lp.cpu().double() - ref.detach()).abs().max().item() < 1e-5
    (lp * up.to(cuda)).sum().backward()
    assert (xd.grad.cpu().double() - xr.grad).abs().max().item() < 1e-5


def test_ppo_returns_advantages_forward_order_quirk(cuda):
    r = torch.tensor([1.0, 2.0, 3.0])
    want = rl_math.ppo_returns([x for x in r], 0.5, normalize=False)
    assert want.flatten().tolist() == [4.25, 2.5, 1.0]
    got, _ = rl_ops.ppo_returns_adv(r.to(cuda), torch.zeros(3, device=cuda), 0.5, normalize=False)
    assert got.flatten().tolist() == [4.25, 2.5, 1.0]
    g = torch.Generator().manual_seed(5)
    rewards, values = torch.rand(30, generator=g), torch.randn(30, 1, generator=g)
    wr = rl_math.ppo_returns([x for x in rewards], 0.99)
    wa = rl_math.ppo_advantages(wr, values)
    gr, ga = rl_ops.ppo_returns_adv(rewards.to(cuda), values.to(cuda), 0.99)
    assert (gr.cpu() - wr).abs().max().item() < 1e-4 and (ga.cpu() - wa).abs().max().item() < 1e-4


def test_ppo_policy_loss_and_grad_with_int_truncation(cuda):
    g = torch.Generator().manual_seed(6)
    new = (torch.randn(25, 6, generator=g) * 0.3 - 0.5)
    old = (torch.randn(30, 25, 6, generator=g) * 1.2 - 0.8)
    old_int = old.long()                       # [-0.3, -1.7, -2.0] -> [0, -1, -2]
    assert torch.tensor([-0.3, -1.7, -2.0]).long().tolist() == [0, -1, -2]
    adv = torch.randn(30, 1, generator=g)
    nr = new.double().requires_grad_(True)
    want = rl_math.ppo_policy_loss(nr, old_int, adv.double(), 0.2)
    want.backward()
    nd = new.to(cuda).requires_grad_(True)
    got = rl_ops.ppo_policy_loss(nd, old_int.to(cuda), adv.to(cuda), 0.2)
    got.backward()
    assert abs(got.item() - want.item()) < 1e-5
    assert (nd.grad.cpu().double() - nr.grad).abs().max().item() < 1e-5


def test_dqn_td_loss_and_grad_with_batch0_gather_quirk(cuda):
    B, T, NA = 30, 50, 25
    y, yt = _logits(B, T, 7), _logits(B, T, 8)
    g = torch.Generator().manual_seed(9)
    action = torch.stack([torch.randint(0, n, (B, NA), generator=g) for n in N_CLASS], -1)
    reward, done = torch.rand(B, 1, generator=g), torch.randint(0, 2, (B, 1), generator=g)
    yr = [t.double().requires_grad_(True) for t in y]
    want, per = rl_math.dqn_td_loss(yr, [t.double() for t in yt], action, reward.double(), done, 0.95, NA)
    want.backward()
    fy = _fused(y, cuda).requires_grad_(True)
    mse = rl_ops.dqn_td_mse(fy, _fused(yt, cuda), action.to(cuda), reward.to(cuda), done.to(cuda), N_CLASS, 0.95)
    loss = mse.sum() / 6
    loss.backward()
    assert abs(loss.item() - want.item()) < 1e-4 * max(1.0, abs(want.item()))
    assert (mse.cpu().double() - torch.stack([p.detach() for p in per])).abs().max().item() < 1e-4
    o = 0
    for f, n in enumerate(N_CLASS):
        gg = fy.grad[..., o:o + n].cpu().double()
        assert (gg - yr[f].grad).abs().max().item() < 1e-5
        assert gg[1:].abs().sum().item() == 0            # only batch element 0 receives gradient
        o += n
