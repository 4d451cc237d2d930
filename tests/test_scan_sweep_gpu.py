"""GPU: the one-sweep backward of the bf16 scan (cwlt_causal_linear_bwd_sweep) -- dQ, dK, dV from a single reverse pass
that takes the dQ scan's prefix state from the forward's final state by subtraction -- against the f64 oracle, the
dkdv + dq kernel pair, and itself across batch sizes.  Reference: the backward of fast_transformers'
causal_dot_product reached from dqn_policy/model.py:128-137."""
import pytest
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import ops
from oracle import cla as ocla

pytestmark = pytest.mark.gpu
BF16_TOL = 2.0 ** -7          # one bf16 rounding of a result of the tensor's scale, as in test_cla_gpu.py


def _pair(qd, kd, vd, gd):
    _, _, _, out, zinv = ops.cla_fwd(qd, kd, vd)
    return ops.cla_bwd(qd, kd, vd, out, zinv, gd, want_colsum=True)


def _sweep(qd, kd, vd, gd):
    _, _, _, out, zinv, fin = ops.cla_fwd(qd, kd, vd, final_state=True)
    assert fin is not None, "the forward did not hand over its final state"
    return ops.cla_bwd(qd, kd, vd, out, zinv, gd, want_colsum=True, final_state=fin), fin


@pytest.mark.parametrize("N,L,H", [(1, 64, 1), (2, 128, 2), (1, 100, 1), (3, 200, 2), (1, 1024, 2), (2, 1000, 1),
                                   (1, 4096, 1), (1, 1, 1), (1, 65, 3), (2, 192, 8),
                                   (8, 130, 3), (16, 64, 2)])      # N % 8 == 0: whole sequences dealt to XCDs
def test_sweep_matches_oracle_and_pair(cuda, monkeypatch, N, L, H):
    monkeypatch.setenv("CWLT_SCAN_SEGMENTS", "1")
    g0 = torch.Generator().manual_seed(7 * L + H)
    q, k, v, g = (torch.randn(N, L, H, 64, generator=g0).bfloat16() for _ in range(4))
    qd, kd, vd, gd = (t.to(cuda) for t in (q, k, v, g))
    (d1, b1), fin = _sweep(qd, kd, vd, gd)
    d2, b2 = _pair(qd, kd, vd, gd)
    torch.cuda.synchronize()
    # final state = sum_j phi(k_j) v_j^T (transposed) | sum_j phi(k_j), per (sequence, head)
    kf = torch.nn.functional.elu(k.double()) + 1
    S = torch.einsum("nlhe,nlhm->nhme", kf.bfloat16().double(), v.double())      # [m][e]
    fin = fin.view(N, H, 65, 64).double().cpu()
    assert (fin[:, :, :64] - S).abs().max().item() <= 1e-4 * max(1.0, S.abs().max().item())
    ks = kf.bfloat16().double().sum(1)
    assert (fin[:, :, 64] - ks).abs().max().item() <= 1e-4 * max(1.0, ks.abs().max().item())
    d1c, d2c = d1.float().cpu(), d2.float().cpu()
    for i in range(3):
        a, b = d1c[:, :, i], d2c[:, :, i]
        assert (a - b).abs().max().item() <= BF16_TOL * max(1.0, b.abs().max().item()), "dq dk dv"[3 * i:3 * i + 2]
    assert (b1.cpu() - b2.cpu()).abs().max().item() <= 2e-2 * max(1.0, b2.abs().max().item())
    if N * L * H <= 8192:
        ref = ocla.cla_grads(q.double(), k.double(), v.double(), g.double())
        for i, name in enumerate(("dq", "dk", "dv")):
            r = ref[1 + i]
            err = (d1c[:, :, i].double() - r).abs().max().item()
            assert err <= BF16_TOL * max(1.0, r.abs().max().item()), (name, err)
        # column sums: those of the stored (bf16) gradients
        want = d1c.double().sum((0, 1)).reshape(-1)
        assert (b1.double().cpu() - want).abs().max().item() <= 1e-3 * max(1.0, want.abs().max().item())


def test_sweep_through_strided_qkv_and_autograd(cuda, monkeypatch):
    """q, k, v as views of one (N, L, 3, H, 64) projection (row stride 3*H*64), through the autograd Function."""
    monkeypatch.setenv("CWLT_SCAN_SEGMENTS", "1")
    g0 = torch.Generator().manual_seed(5)
    N, L, H = 2, 320, 4
    qkv = torch.randn(N, L, 3, H, 64, generator=g0).bfloat16().to(cuda).requires_grad_(True)
    g = torch.randn(N, L, H, 64, generator=g0).bfloat16().to(cuda)
    out = ops.causal_linear_attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
    out.backward(g)
    ref = ocla.cla_grads(*(qkv.detach()[:, :, i].double().cpu() for i in range(3)), g.double().cpu())
    assert (out.detach().double().cpu() - ref[0]).abs().max().item() <= BF16_TOL * max(1.0, ref[0].abs().max().item())
    for i in range(3):
        err = (qkv.grad[:, :, i].double().cpu() - ref[1 + i]).abs().max().item()
        assert err <= BF16_TOL * max(1.0, ref[1 + i].abs().max().item())


def test_backward_takes_a_dout_view_with_an_odd_row_stride(cuda, monkeypatch):
    """dout as a strided view (row stride 132 elements: not a multiple of 8) is a legal autograd input of the public
    causal_linear_attention(); the forward has already handed over its final state for the one-sweep backward, which
    cannot take such a view -- the backward then drops the final state and runs the dkdv + dq pair (any stride)."""
    monkeypatch.setenv("CWLT_SCAN_SEGMENTS", "1")
    N, L, H = 2, 200, 2
    g0 = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(N, L, H, 64, generator=g0).bfloat16() for _ in range(3))
    big = torch.randn(N, L, H * 64 + 4, generator=g0).bfloat16()
    ref = ocla.cla_grads(q.double(), k.double(), v.double(), big[:, :, :H * 64].reshape(N, L, H, 64).double())
    qd, kd, vd = (t.to(cuda).requires_grad_(True) for t in (q, k, v))
    out = ops.causal_linear_attention(qd, kd, vd)
    gview = big.to(cuda)[:, :, :H * 64].view(N, L, H, 64)
    assert gview.stride(1) == H * 64 + 4 and not gview.is_contiguous()
    out.backward(gview)
    for got, r in zip((out, qd.grad, kd.grad, vd.grad), ref):
        err = (got.detach().cpu().double() - r).abs().max().item()
        assert err <= BF16_TOL * max(1.0, r.abs().max().item()), err
    # the raw wrapper: same inputs, final state passed explicitly
    _, _, _, o2, zinv, fin = ops.cla_fwd(qd.detach(), kd.detach(), vd.detach(), final_state=True)
    d2 = ops.cla_bwd(qd.detach(), kd.detach(), vd.detach(), o2, zinv, gview, final_state=fin)
    assert torch.equal(d2[:, :, 0], qd.grad) and torch.equal(d2[:, :, 2], vd.grad)


def test_sweep_is_batch_independent_and_deterministic_at_bench_size(cuda, monkeypatch):
    """(B, 1024, 8, 64): every stream is one workgroup, so a sequence's gradients do not depend on its neighbours,
    and two runs agree bit for bit."""
    monkeypatch.setenv("CWLT_SCAN_SEGMENTS", "1")      # the 8-sequence sub-batch would otherwise be cut into segments
    g0 = torch.Generator().manual_seed(11)
    N, L, H = 48, 1024, 8
    q, k, v, g = (torch.randn(N, L, H, 64, generator=g0).bfloat16().to(cuda) for _ in range(4))
    (d1, b1), _ = _sweep(q, k, v, g)
    (d2, b2), _ = _sweep(q, k, v, g)
    assert torch.equal(d1, d2) and torch.equal(b1, b2)
    (d3, _), _ = _sweep(q[40:], k[40:], v[40:], g[40:])
    assert torch.equal(d1[40:], d3)
    dp, bp = _pair(q, k, v, g)
    for i in range(3):
        a, b = d1[:, :, i].float(), dp[:, :, i].float()
        assert (a - b).abs().max().item() <= BF16_TOL * max(1.0, b.abs().max().item())
    assert (b1 - bp).abs().max().item() <= 2e-2 * max(1.0, bp.abs().max().item())


def test_sweep_entry_rejects_what_it_cannot_run(cuda):
    from rlmg_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1, 64, 1, 64, device=cuda)
    z = torch.zeros(1, 64, 1, device=cuda)
    fin = torch.zeros(65 * 64, device=cuda)
    p = lambda t: t.data_ptr()
    args = lambda code, ld: (p(x), p(x), p(x), p(x), p(z), p(x), p(fin), p(x), p(x), p(x), None, None, None,
                             1, 1, 64, 64, ld, ld, ld, ld, ld, ld, ld, ld, code, None)
    assert lib.cwlt_causal_linear_bwd_sweep(*args(0, 64)) == 1002          # f32: CWLT_ERR_DTYPE
    assert lib.cwlt_causal_linear_bwd_sweep(*args(1, 68)) == 1001          # row stride not a multiple of 8
    assert lib.cwlt_scan_final_state_floats(4, 8) == 4 * 8 * 65 * 64
