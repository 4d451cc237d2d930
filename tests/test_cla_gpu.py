"""Parity of the HIP causal-linear-attention kernels (through the C-ABI) against the CPU oracle."""
import pytest
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import ops
from oracle import cla as ocla

pytestmark = pytest.mark.gpu

TOL = 1e-4  # BASELINE.json north_star: logits within 1e-4 fp32


def _rand(N, L, H, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(N, L, H, 64, generator=g) * scale for _ in range(4)]


@pytest.mark.parametrize("N,L,H", [(1, 1, 1), (2, 50, 8), (1, 32, 2), (3, 33, 1), (2, 257, 4), (1, 1024, 8)])
def test_cla_fwd_bwd_fp32_matches_oracle(cuda, N, L, H):
    q, k, v, g = _rand(N, L, H, seed=100 + L)
    ref_out, ref_dq, ref_dk, ref_dv = ocla.cla_grads(q.double(), k.double(), v.double(), g.double())
    qd, kd, vd = (t.to(cuda).requires_grad_(True) for t in (q, k, v))
    out = ops.causal_linear_attention(qd, kd, vd)
    out.backward(g.to(cuda))
    torch.cuda.synchronize()
    for name, got, ref in (("out", out, ref_out), ("dq", qd.grad, ref_dq), ("dk", kd.grad, ref_dk),
                           ("dv", vd.grad, ref_dv)):
        err = (got.detach().cpu().double() - ref).abs().max().item()
        scale = max(1.0, ref.abs().max().item())
        assert err <= TOL * scale, "%s: max abs err %.3e (scale %.2f) at N=%d L=%d H=%d" % (name, err, scale, N, L, H)


def test_cla_strided_qkv_views(cuda):
    """q, k, v as column slices of one fused (N*L, 3*H*64) projection buffer (no copies)."""
    N, L, H = 2, 70, 8
    g = torch.Generator().manual_seed(7)
    qkv = torch.randn(N, L, 3, H, 64, generator=g)
    dout = torch.randn(N, L, H, 64, generator=g)
    ref = ocla.cla_reference(qkv[:, :, 0].double(), qkv[:, :, 1].double(), qkv[:, :, 2].double())
    x = qkv.to(cuda)
    out = ops.causal_linear_attention(x[:, :, 0], x[:, :, 1], x[:, :, 2])
    assert (out.cpu().double() - ref).abs().max().item() < TOL


def test_cla_causality(cuda):
    """Perturbing token t must leave outputs before t bit-identical."""
    N, L, H = 1, 96, 2
    q, k, v, _ = _rand(N, L, H, seed=3)
    a = ops.causal_linear_attention(q.to(cuda), k.to(cuda), v.to(cuda)).cpu()
    k2, v2 = k.clone(), v.clone()
    k2[:, 40:] += 1.0
    v2[:, 40:] -= 2.0
    b = ops.causal_linear_attention(q.to(cuda), k2.to(cuda), v2.to(cuda)).cpu()
    assert torch.equal(a[:, :40], b[:, :40])
    assert not torch.equal(a[:, 40:], b[:, 40:])


@pytest.mark.parametrize("N,L,H", [(2, 130, 8), (1, 1, 1), (1, 64, 2), (3, 65, 1), (1, 50, 8), (2, 1024, 2)])
def test_cla_bf16_io(cuda, N, L, H):
    """bf16 storage + bf16 MFMA (f32 running states): compare against the oracle run on the bf16-rounded inputs."""
    q, k, v, g = (t.bfloat16() for t in _rand(N, L, H, seed=11))
    ref_out, ref_dq, ref_dk, ref_dv = ocla.cla_grads(q.double(), k.double(), v.double(), g.double())
    qd, kd, vd = (t.to(cuda).requires_grad_(True) for t in (q, k, v))
    out = ops.causal_linear_attention(qd, kd, vd)
    out.backward(g.to(cuda))
    for got, ref in ((out, ref_out), (qd.grad, ref_dq), (kd.grad, ref_dk), (vd.grad, ref_dv)):
        err = (got.detach().cpu().double() - ref).abs().max().item()
        assert err <= 2.0 ** -7 * max(1.0, ref.abs().max().item())  # one bf16 rounding of the result
