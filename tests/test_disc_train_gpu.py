"""GPU: discriminator TRAINING path -- band-attention backward, the differentiable Longformer schedule and
`RewardDiscri.update_disc(train=True)` -- vs torch autograd of the oracle and the gradients recorded from the
reference's own AIRL_model.LongFormer + HF Longformer (tests/golden/airl_grads_small.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401
from rlmg_amd import ops  # noqa: E402
from oracle import longformer as olf  # noqa: E402

pytestmark = pytest.mark.gpu


def _qkv(B, L, H, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, L, 3, H, 64, generator=g)


def _ref_grads(qkv, mask, w, dout, keep=None, keep_scale=1.0):
    """Dense f64 autograd reference: olf.band_attention when no dropout, else explicit masked softmax."""
    x = qkv.double().requires_grad_(True)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    if keep is None:
        out = olf.band_attention(q, k, v, mask, w)
    else:
        B, L, H, D = q.shape
        s = torch.einsum("blhd,bmhd->bhlm", q / 8.0, k)
        i = torch.arange(L)
        band = (i[:, None] - i[None, :]).abs() <= w
        ok = band[None, None] & (mask[:, None, None, :] != 0 if mask is not None else True)
        p = torch.softmax(s.masked_fill(~ok, float("-inf")), -1)
        p = torch.nan_to_num(p) * keep.double() * keep_scale
        out = torch.einsum("bhlm,bmhd->blhd", p, v).reshape(B, L, H * D)
        if mask is not None:
            out = out * (mask[:, :, None] != 0)
    out.backward(dout.double())
    return out.detach(), x.grad


@pytest.mark.parametrize("B,L,H,w", [(2, 50, 2, 25), (1, 50, 1, 256), (2, 70, 1, 8), (1, 200, 2, 25), (1, 1, 1, 3),
                                     (1, 130, 1, 64)])
def test_band_attention_backward_matches_autograd(cuda, B, L, H, w):
    qkv = _qkv(B, L, H, 7 * L + w)
    mask = torch.ones(B, L)
    if L > 10:
        mask[0, L - 5:] = 0
        mask[-1, 2] = 0
    dout = torch.randn(B, L, H * 64, generator=torch.Generator().manual_seed(5))
    for m in (mask, None):
        ref_out, ref_g = _ref_grads(qkv, m, w, dout)
        x = qkv.to(cuda).requires_grad_(True)
        out = ops.BandAttentionFn.apply(x, None if m is None else m.to(cuda), w, 0.0, 0)
        out.backward(dout.to(cuda))
        assert (out.detach().cpu().double() - ref_out).abs().max().item() < 1e-4
        assert (x.grad.cpu().double() - ref_g).abs().max().item() < 2e-4


def test_band_attention_backward_bf16(cuda):
    B, L, H, w = 2, 100, 2, 25
    qkv = _qkv(B, L, H, 3).bfloat16()
    dout = torch.randn(B, L, H * 64, generator=torch.Generator().manual_seed(6)).bfloat16()
    _, ref_g = _ref_grads(qkv.float(), None, w, dout.float())
    x = qkv.to(cuda).requires_grad_(True)
    ops.BandAttentionFn.apply(x, None, w, 0.0, 0).backward(dout.to(cuda))
    err = (x.grad.float().cpu().double() - ref_g).abs().max().item()
    assert err < 0.03 * ref_g.abs().max().item() + 1e-2


def test_band_attention_backward_regenerates_the_dropout_mask(cuda):
    """The keep pattern depends only on (seed, b, h, i, j): read it off a forward with uniform probabilities and
    one-hot values, then check forward and backward of random inputs against autograd with that explicit mask."""
    B, L, H, w, p, seed = 2, 60, 2, 20, 0.25, 991
    probe = torch.zeros(B, L, 3, H, 64, device=cuda)
    probe[:, :, 2] = torch.eye(64, device=cuda)[:L][None, :, None, :]          # v_j = e_j
    pd = ops.band_attention(probe[:, :, 0], probe[:, :, 1], probe[:, :, 2], None, w, p, seed).view(B, L, H, 64)
    keep = (pd[..., :L] != 0).permute(0, 2, 1, 3).cpu()                         # (B, H, i, j)
    i = torch.arange(L)
    band = ((i[:, None] - i[None, :]).abs() <= w)
    frac = keep[:, :, band].float().mean().item()
    assert abs(frac - (1 - p)) < 0.02 and not keep[:, :, ~band].any()
    qkv = _qkv(B, L, H, 12)
    dout = torch.randn(B, L, H * 64, generator=torch.Generator().manual_seed(8))
    ref_out, ref_g = _ref_grads(qkv, None, w, dout, keep=keep, keep_scale=1.0 / (1 - p))
    x = qkv.to(cuda).requires_grad_(True)
    out = ops.BandAttentionFn.apply(x, None, w, p, seed)
    out.backward(dout.to(cuda))
    assert (out.detach().cpu().double() - ref_out).abs().max().item() < 2e-3    # 16-bit keep threshold: scale ~1/(1-p)
    assert (x.grad.cpu().double() - ref_g).abs().max().item() < 2e-3


def _small_airl(seed, n_class):
    from rlmg_amd.dqn_policy import AIRL_model
    old = (AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD)
    AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = 128, 2, 2
    try:
        net = fill_params(AIRL_model.LongFormer(n_class), seed=seed)
    finally:
        AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = old
    with torch.no_grad():
        net.score_classifier[1].running_mean.copy_(torch.linspace(-0.2, 0.2, 128))
        net.score_classifier[1].running_var.copy_(torch.linspace(0.5, 1.5, 128))
    return net


def test_discriminator_gradients_match_reference_fixture(cuda):
    fx = np.load(os.path.join(HERE, "golden", "airl_grads_small.npz"), allow_pickle=False)
    net = _small_airl(43, fx["n_class"].tolist()).to(cuda).eval()
    x_exp, x_ag, mask = (torch.from_numpy(fx[k]).to(cuda) for k in ("x_exp", "x_agent", "mask"))
    bce = torch.nn.BCELoss()
    e = bce(net(x_exp, mask), torch.ones(3, 1, device=cuda))
    c = net.token_forward(x_ag, x_exp, mask)
    a = bce(net(x_ag, mask), torch.zeros(3, 1, device=cuda))
    assert np.allclose([e.item(), a.item(), c.item()], fx["losses"], atol=1e-4)
    (e + (a + c)).backward()
    params = dict(net.named_parameters())
    for k, want_norm in zip(fx["names"].tolist(), fx["norms"]):
        g = params[k].grad
        assert g is not None, k
        assert abs(g.double().norm().item() - want_norm) < 1e-4 + 1e-3 * want_norm, k
        want = torch.from_numpy(fx["grad." + k])
        got = (g[:8] if g.numel() > 4096 else g).cpu()
        assert (got - want).abs().max().item() < 1e-4, k
    # scoring schedule (autograd off) gives the same scores as the training schedule
    with torch.no_grad():
        s0 = net(x_exp, mask)
    assert (s0 - net(x_exp, mask).detach()).abs().max().item() < 1e-5


def test_update_disc_trains_and_scores(cuda, tmp_path, monkeypatch):
    """A separable toy problem: expert windows use low token ids, agent windows high ones.  After
    update_disc(train=True) the BCE terms must have dropped and the checkpoint / reward pickle exist."""
    from rlmg_amd.dqn_policy import AIRL, AIRL_model
    monkeypatch.chdir(tmp_path)
    n_class = [56, 135, 18, 87, 18, 25]
    monkeypatch.setattr(AIRL_model, "D_MODEL", 128)
    monkeypatch.setattr(AIRL_model, "N_LAYER", 2)
    monkeypatch.setattr(AIRL_model, "N_HEAD", 2)
    torch.manual_seed(3)
    disc = AIRL.RewardDiscri(n_class, Pretrain=False)
    disc.batch_size, disc.epoch_disc = 20, 3
    g = torch.Generator().manual_seed(4)
    n, W = 60, 50
    exp = torch.stack([torch.randint(0, c // 2, (n, W), generator=g) for c in n_class], -1)
    ag = torch.stack([torch.randint(c // 2, c, (n, W), generator=g) for c in n_class], -1)
    done = torch.zeros(n, 1)
    mask = torch.ones(n, W)
    before = {k: v.clone() for k, v in disc.disc_model.state_dict().items()}
    traj, ans = disc.update_disc((ag, None, None, ag, done), (exp, None, None, exp, done, mask, mask), train=True)
    assert traj.shape == (n, 1) and ans.shape == (n, 1)
    assert os.path.exists("./ckpt/disc_IRL.pt") and os.path.exists("./exp/IRL_reward.pickle")
    ls = disc.last_losses
    assert len(ls) == 3 and all(np.isfinite(list(l.values())).all() for l in ls)
    assert ls[-1]["expert"] + ls[-1]["agent"] < ls[0]["expert"] + ls[0]["agent"]
    assert ls[-1]["ce"] < ls[0]["ce"]
    after = disc.disc_model.state_dict()
    changed = [k for k in before if before[k].dtype.is_floating_point and not torch.equal(before[k], after[k])]
    assert any("attention.self.query.weight" in k for k in changed) and any("word_emb_pitch" in k for k in changed)
    # the reference reloads the epoch-0 checkpoint before scoring: parameters now equal the file's (BatchNorm's
    # running statistics moved on, because scoring runs in train() mode)
    sd = torch.load("./ckpt/disc_IRL.pt")["model_state_dict"]
    assert all(torch.equal(sd[k].to(after[k].device), after[k]) for k, _ in disc.disc_model.named_parameters())


def test_score_in_groups_equals_batch_by_batch_forward(cuda):
    """Grouped scoring (Longformer over many windows, classifier per group of `group`) == forward() per batch,
    including the BatchNorm running statistics left behind (train mode, dropout probabilities zeroed)."""
    import copy
    n_class = [56, 135, 18, 87, 18, 25]
    net = _small_airl(47, n_class).to(cuda).train()
    net.longformer.p_hidden = net.longformer.p_attn = 0.0
    g = torch.Generator().manual_seed(9)
    x = torch.stack([torch.randint(0, c, (60, 50), generator=g) for c in n_class], -1).to(cuda)
    mask = torch.ones(60, 50, dtype=torch.long, device=cuda)
    mask[7, 41:] = 0
    ref_net = copy.deepcopy(net)
    with torch.no_grad():
        want = torch.cat([ref_net(x[s:s + 20], mask[s:s + 20]) for s in range(0, 60, 20)], 0)
        got = net.score_in_groups(x, mask, 20, windows_per_pass=50)          # passes of 40 + 20 windows
    assert (got - want).abs().max().item() < 1e-5
    bn, bn_ref = net.score_classifier[1], ref_net.score_classifier[1]
    assert (bn.running_mean - bn_ref.running_mean).abs().max().item() < 1e-6
    assert (bn.running_var - bn_ref.running_var).abs().max().item() < 1e-6
    assert int(bn.num_batches_tracked) == int(bn_ref.num_batches_tracked) == 3


def test_calculate_reward_tail_keeps_one_and_missing_checkpoint_keeps_weights(cuda, tmp_path, monkeypatch):
    """`RewardDiscri.calculate_reward` (/root/reference/dqn_policy/AIRL.py:69-91): whole batches of `batch_size` windows
    are scored (train() mode, as `all_forward` forces), a tail shorter than a batch keeps the initial 1.0; fewer windows
    than one batch: all ones.  No checkpoint on disk: the current weights score (differences)."""
    from rlmg_amd.dqn_policy import AIRL, AIRL_model
    monkeypatch.chdir(tmp_path)
    n_class = [56, 135, 18, 87, 18, 25]
    monkeypatch.setattr(AIRL_model, "D_MODEL", 128)
    monkeypatch.setattr(AIRL_model, "N_LAYER", 2)
    monkeypatch.setattr(AIRL_model, "N_HEAD", 2)
    torch.manual_seed(5)
    disc = AIRL.RewardDiscri(n_class, Pretrain=False)
    disc.batch_size = 20
    disc.disc_model.longformer.p_hidden = disc.disc_model.longformer.p_attn = 0.0      # deterministic scores
    g = torch.Generator().manual_seed(6)
    n, W = 53, 50
    x = torch.stack([torch.randint(0, c, (n, W), generator=g) for c in n_class], -1)
    done, mask = torch.zeros(n, 1), torch.ones(n, W)
    assert not os.path.exists("./ckpt/disc_IRL.pt")
    import copy
    ref = copy.deepcopy(disc.disc_model).train()
    pred = disc.calculate_reward(x, done, x, mask, mask)
    assert pred.shape == (n, 1) and pred.device.type == "cpu"
    assert torch.equal(pred[40:], torch.ones(13, 1))                    # the tail of 13 windows was never scored
    with torch.no_grad():
        want = torch.cat([ref(x[s:s + 20].long().to(cuda), mask[s:s + 20].long().to(cuda)) for s in (0, 20)], 0)
    assert (pred[:40] - want.float().cpu()).abs().max().item() < 1e-5
    assert (pred[:40] != 1.0).all()
    few = disc.calculate_reward(x[:7], done[:7], x[:7], mask[:7], mask[:7])
    assert torch.equal(few, torch.ones(7, 1))
