"""GPU: recurrent (generation) form of the encoder on the libcwlt kernels: token-by-token it reproduces the
CPU oracle's recurrent encoder; checkpoints interchange between the parallel and recurrent forms."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401
from rlmg_amd import ops  # noqa: E402
from oracle import cla as ocla, cw_model  # noqa: E402

pytestmark = pytest.mark.gpu


def test_recurrent_step_kernel_matches_oracle(cuda):
    N, H, T = 3, 8, 20
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(T, N, 3 * H * 64, generator=g)
    S = torch.zeros(N, H, 64, 64, device=cuda)
    Z = torch.zeros(N, H, 64, device=cuda)
    state = None
    for t in range(T):
        q, k, v = (qkv[t, :, i * H * 64:(i + 1) * H * 64].view(N, H, 64) for i in range(3))
        want, state = ocla.cla_recurrent_step(q.double(), k.double(), v.double(), state)
        got = ops.recurrent_cla_step(qkv[t].to(cuda), S, Z, H)
        assert (got.cpu().double().view(N, H, 64) - want).abs().max().item() < 1e-4
    assert (S.cpu().double() - state[0]).abs().max().item() < 1e-3
    assert (Z.cpu().double() - state[1]).abs().max().item() < 1e-4


def test_recurrent_scan_equals_parallel_scan(cuda):
    """Feeding the tokens one by one through the recurrent kernel gives the chunked training kernel's output."""
    N, H, T = 2, 4, 70
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(N, T, 3, H, 64, generator=g).to(cuda)
    par = ops.causal_linear_attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
    S = torch.zeros(N, H, 64, 64, device=cuda)
    Z = torch.zeros(N, H, 64, device=cuda)
    for t in range(T):
        got = ops.recurrent_cla_step(qkv[:, t].reshape(N, 3 * H * 64), S, Z, H)
        assert (got.view(N, H, 64) - par[:, t]).abs().max().item() < 1e-4


def test_recurrent_model_matches_oracle_and_shares_checkpoints(cuda):
    from rlmg_amd.dqn_policy import config, model
    n_class = [56, 135, 18, 87, 18, 25]
    old = dict(config.AgentConfig)
    config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    try:
        par = fill_params(model.LinearTransformer(n_class, is_training=True), seed=81)
        rec = model.LinearTransformer(n_class, is_training=False).to(cuda).eval()
    finally:
        config.AgentConfig.update(old)
    rec.load_state_dict(par.state_dict())                 # same keys in both forms
    ref = fill_params(cw_model.CWLinearTransformer(n_class, 128, 2, 2, variant="dqn", recurrent=True), seed=81).eval()
    g = torch.Generator().manual_seed(3)
    x = torch.stack([torch.randint(0, n, (1, 24), generator=g) for n in n_class], -1)
    with torch.no_grad():
        mem, mem_ref = None, None
        for t in range(24):
            # one token per call, positions do not advance (pos_emb sees T = 1): testing-no-type-cp.py:150-166
            h_t, mem = rec.forward_hidden(x[:, t:t + 1].to(cuda), memory=mem, is_training=False)
            r_t, mem_ref = ref.forward_hidden(x[:, t:t + 1], memory=mem_ref, is_training=False)
            assert (h_t.cpu() - r_t).abs().max().item() < 1e-4
        nxt = rec.forward_output_sampling(h_t)
    assert nxt.shape == (6,) and all(0 <= int(v) < n for v, n in zip(nxt, n_class))
