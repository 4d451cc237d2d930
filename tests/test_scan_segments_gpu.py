"""GPU: the few-stream (segmented, "T-split") schedule of the bf16 scan kernels -- every sequence cut into runs of
whole chunks, one workgroup each, with a state-increment pass and a prefix pass in front -- against the one-pass
schedule and the f64 oracle.  The reference reaches this kernel with 4 sequences x 8 heads = 32 streams
(dqn_policy/agent_pretrain.py:48,524-526: batch 4, T = 3584)."""
import pytest
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import ops
from oracle import cla as ocla

pytestmark = pytest.mark.gpu
BF16_TOL = 2.0 ** -7


def _run(q, k, v, g, cuda):
    qd, kd, vd = (t.to(cuda) for t in (q, k, v))
    _, _, _, out, zinv = ops.cla_fwd(qd, kd, vd)
    dqkv, dbias = ops.cla_bwd(qd, kd, vd, out, zinv, g.to(cuda), want_colsum=True)
    return out.float().cpu(), dqkv.float().cpu(), dbias.cpu(), zinv.cpu()


@pytest.mark.parametrize("N,L,H", [(1, 1024, 2), (2, 3584, 1), (1, 4096, 8), (1, 257, 1), (3, 200, 2), (1, 130, 1)])
@pytest.mark.parametrize("segs", [2, 3, 8, 16])
def test_segmented_scan_equals_one_pass_and_oracle(cuda, monkeypatch, N, L, H, segs):
    g0 = torch.Generator().manual_seed(L + segs)
    q, k, v, g = (torch.randn(N, L, H, 64, generator=g0).bfloat16() for _ in range(4))
    monkeypatch.setenv("CWLT_SCAN_SEGMENTS", "1")
    assert ops.scan_segments(N, H, L, torch.bfloat16) == 1
    o1, d1, b1, z1 = _run(q, k, v, g, cuda)
    monkeypatch.setenv("CWLT_SCAN_SEGMENTS", str(segs))
    P = ops.scan_segments(N, H, L, torch.bfloat16)
    assert 1 <= P <= min(segs, (L + 63) // 64)
    oP, dP, bP, zP = _run(q, k, v, g, cuda)
    # same arithmetic up to the order in which chunk contributions enter the f32 states (and one extra bf16
    # rounding of nothing): results agree to a fraction of a bf16 ulp of the tensor's scale
    for a, b in ((o1, oP), (d1, dP)):
        assert (a - b).abs().max().item() <= 0.5 * BF16_TOL * max(1.0, a.abs().max().item())
    assert (z1 - zP).abs().max().item() <= 1e-5 * max(1.0, z1.abs().max().item())
    assert (b1 - bP).abs().max().item() <= 2e-2 * max(1.0, b1.abs().max().item())
    if N * L * H <= 8192:
        ref = ocla.cla_grads(q.double(), k.double(), v.double(), g.double())
        got = (oP, dP[:, :, 0], dP[:, :, 1], dP[:, :, 2])
        for a, r in zip(got, ref):
            assert (a.double() - r).abs().max().item() <= BF16_TOL * max(1.0, r.abs().max().item())


def test_library_picks_segments_only_for_few_streams(cuda, monkeypatch):
    monkeypatch.delenv("CWLT_SCAN_SEGMENTS", raising=False)
    bf = torch.bfloat16
    assert ops.scan_segments(512, 8, 1024, bf) == 1            # bench shape: the streams fill the chip
    assert ops.scan_segments(64, 8, 1024, bf) == 1
    P = ops.scan_segments(4, 8, 3584, bf)                      # the reference's own pretrain batch
    assert P == 14 and ops.scan_segments(4, 8, 3584, torch.float32) == 1   # 56 chunks -> 14 runs of 4: 448 workgroups
    assert ops.scan_segments(1, 8, 50, bf) == 1                # one chunk: nothing to cut
    assert 2 <= ops.scan_segments(30, 8, 1024, bf) <= 8


def test_segmented_default_path_through_the_autograd_function(cuda, monkeypatch):
    """No env override: (2, 2048, 2) is few-stream, so the autograd Function takes the segmented schedule."""
    monkeypatch.delenv("CWLT_SCAN_SEGMENTS", raising=False)
    assert ops.scan_segments(2, 2, 2048, torch.bfloat16) > 1
    g0 = torch.Generator().manual_seed(5)
    q, k, v, g = (torch.randn(2, 2048, 2, 64, generator=g0).bfloat16() for _ in range(4))
    ref = ocla.cla_grads(q.double(), k.double(), v.double(), g.double())
    qd, kd, vd = (t.to(cuda).requires_grad_(True) for t in (q, k, v))
    out = ops.causal_linear_attention(qd, kd, vd)
    out.backward(g.to(cuda))
    for got, r in zip((out, qd.grad, kd.grad, vd.grad), ref):
        assert (got.detach().cpu().double() - r).abs().max().item() <= BF16_TOL * max(1.0, r.abs().max().item())
