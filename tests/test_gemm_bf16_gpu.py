"""GPU: cwlt_gemm_bf16 -- the projection GEMM C (M, N) [+]= A (M, K) . W (N, K)^T [+ bias] of the encoder layer
(query / key / value / out projection, linear1, linear2 and the six heads as one projection:
/root/reference/dqn_policy/model.py:128-137,156-161,241-249) in its forward and input-gradient forms -- against the f64
product of the same bf16 operands, at every shape the layer uses, with ragged row tiles, a partial column tile, strided
operands, the accumulate form, every schedule variant, and at the bench's 524 288 rows.
Tolerance: the result is ONE rounding of the f32 accumulator to bf16: 2^-8 x |value| + the f32 accumulation error, taken
as 2^-7 x the largest |value| of the row block (as in test_ops_bf16_gpu.py)."""
import pytest
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import _lib, ops

pytestmark = pytest.mark.gpu
BF16_TOL = 2.0 ** -7


@pytest.fixture(autouse=True)
def _all_rows(monkeypatch):
    monkeypatch.setattr(ops, "GEMM_BF16_MIN_ROWS", 0)
    yield
    _lib.load().cwlt_gemm_bf16_tune(-1, None)


def _operands(M, N, K, seed, cuda, lda=None, ldw=None):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, lda or K, generator=g).bfloat16()[:, :K]
    w = (torch.randn(N, ldw or K, generator=g) * (2.0 / K ** 0.5)).bfloat16()[:, :K]
    bias = torch.randn(N, generator=g) * 0.3
    c0 = torch.randn(M, N, generator=g).bfloat16()
    return a, w, bias, c0


def _check(out, ref):
    sc = max(1.0, ref.abs().max().item())
    err = (out.double().cpu() - ref).abs().max().item()
    assert err <= BF16_TOL * sc, (err, sc)


# the layer's shapes (N, K) x forms, at row counts that end inside a tile, inside a 16-row MFMA block, and on a boundary
@pytest.mark.parametrize("M", [1, 255, 256, 1000, 4096 + 17])
@pytest.mark.parametrize("N,K", [(1536, 512), (512, 2048), (512, 512), (2048, 512), (512, 1536), (384, 512)])
def test_gemm_bf16_matches_the_f64_product(cuda, M, N, K):
    a, w, bias, c0 = _operands(M, N, K, 7 * M + N + K, cuda)
    ad, wd = a.to(cuda), w.to(cuda)
    prod = a.double() @ w.double().t()
    out = ops.gemm_bf16(ad, wd)
    assert out.dtype == torch.bfloat16 and out.shape == (M, N)
    _check(out, prod)
    _check(ops.gemm_bf16(ad, wd, bias.to(cuda)), prod + bias.double())
    acc = c0.to(cuda)
    res = ops.gemm_bf16(ad, wd, out=acc, accumulate=True)
    assert res.data_ptr() == acc.data_ptr()
    _check(acc, prod + c0.double())
    acc2 = c0.to(cuda)
    ops.gemm_bf16(ad, wd, bias.to(cuda), out=acc2, accumulate=True)
    _check(acc2, prod + c0.double() + bias.double())


@pytest.mark.parametrize("variant", [0, 1, 0 | (1 << 8), 1 | (1 << 8), (2 << 1) | (1 << 8)])
@pytest.mark.parametrize("K", [128, 192, 256, 320, 2048])
def test_every_schedule_variant_and_short_reductions(cuda, variant, K):
    """K = 128 is two K-tiles (the peeled first and last K-tile alone), 192 / 320 an odd count (the buffer a tile starts in
    alternates from tile to tile): the slot parity and the counted waits of the first and the last two K-tiles are exercised
    at every length, with the next tile's operands requested from inside the last K-tile (bit 0 clear, the default) and
    after the main loop (bit 0 set), with one tile per workgroup and -- the grid held to 8 workgroups (bit 8) -- five to
    six tiles per workgroup, padding tiles of the 8-XCD deal among them, and with a start stagger."""
    M, N = 5000, 512
    a, w, bias, c0 = _operands(M, N, K, 100 + K, cuda)
    _lib.load().cwlt_gemm_bf16_tune(variant, None)
    ad, wd = a.to(cuda), w.to(cuda)
    prod = a.double() @ w.double().t()
    _check(ops.gemm_bf16(ad, wd, bias.to(cuda)), prod + bias.double())
    acc = c0.to(cuda)
    ops.gemm_bf16(ad, wd, out=acc, accumulate=True)
    _check(acc, prod + c0.double())


def test_strided_operands_and_output_view(cuda):
    """q / k / v are column blocks of the (R, 3 D) projection, the output may be a column block too: row strides are
    arguments.  Everything outside the output view must be left untouched."""
    M, N, K = 1500, 512, 512
    a, w, bias, _ = _operands(M, N, K, 5, cuda, lda=3 * K, ldw=K + 64)
    big_a = torch.zeros(M, 3 * K, device=cuda, dtype=torch.bfloat16)
    big_a[:, K:2 * K] = a.to(cuda)
    av = big_a[:, K:2 * K]
    big_w = torch.zeros(N, K + 64, device=cuda, dtype=torch.bfloat16)
    big_w[:, :K] = w.to(cuda)
    wv = big_w[:, :K]
    big_c = torch.full((M, 3 * N), 3.0, device=cuda, dtype=torch.bfloat16)
    cv = big_c[:, N:2 * N]
    assert av.stride(0) == 3 * K and wv.stride(0) == K + 64 and cv.stride(0) == 3 * N
    ops.gemm_bf16(av, wv, bias.to(cuda), out=cv)
    _check(cv, a.double() @ w.double().t() + bias.double())
    assert torch.all(big_c[:, :N] == 3.0) and torch.all(big_c[:, 2 * N:] == 3.0)


def test_exact_integers_and_an_asymmetric_weight(cuda):
    """Small-integer operands make every product and sum exact: any swapped row / column or k-order mistake in the
    fragment maps shows as a wrong integer (an identity-like A with an asymmetric W would hide nothing here)."""
    M, N, K = 512, 512, 256
    g = torch.Generator().manual_seed(11)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    w[:, 0] = torch.arange(N).float() % 7 - 3        # column-dependent, not symmetric in (n, k)
    out = ops.gemm_bf16(a.bfloat16().to(cuda), w.bfloat16().to(cuda))
    ref = (a.double() @ w.double().t())
    exact = ref.abs() <= 256                            # integers up to 256 are exact in bf16
    assert torch.equal(out.double().cpu()[exact], ref[exact])


def test_rejects_what_it_cannot_run(cuda):
    a = torch.randn(256, 512, device=cuda).bfloat16()
    w = torch.randn(512, 512, device=cuda).bfloat16()
    with pytest.raises(RuntimeError):
        ops.gemm_bf16(a[:, :96], w[:, :96])                       # K % 64
    with pytest.raises(RuntimeError):
        ops.gemm_bf16(a[:, :64], w[:, :64])                       # K < 128
    with pytest.raises(RuntimeError):
        ops.gemm_bf16(a, w[:508])                                  # N % 8
    with pytest.raises(ValueError):
        ops.gemm_bf16(a, w, accumulate=True)                       # nothing to add onto
    assert not ops.gemm_bf16_supported(a.float(), w.float())


def test_bench_rows(cuda):
    """R = 524 288 (B = 512, T = 1024): the four (N, K) of the step, checked on three row slabs (first tile, a middle
    XCD-dealt tile, the last rows) against the f64 product; accumulate form on linear1's input gradient."""
    M = 524288
    for N, K, acc in ((1536, 512, False), (512, 2048, True), (512, 512, False)):
        g = torch.Generator(device=cuda).manual_seed(N + K)
        a = torch.randn(M, K, device=cuda, generator=g).bfloat16()
        w = (torch.randn(N, K, device=cuda, generator=g) * (2.0 / K ** 0.5)).bfloat16()
        bias = torch.randn(N, device=cuda, generator=g) * 0.3
        c0 = torch.randn(M, N, device=cuda, generator=g).bfloat16() if acc else None
        out = c0.clone() if acc else None
        out = ops.gemm_bf16(a, w, bias, out=out, accumulate=acc)
        for lo in (0, 256 * 1031, M - 300):
            sl = slice(lo, lo + 300)
            ref = a[sl].double() @ w.double().t() + bias.double()
            if acc:
                ref = ref + c0[sl].double()
            sc = max(1.0, ref.abs().max().item())
            assert (out[sl].double() - ref).abs().max().item() <= BF16_TOL * sc
        del a, w, out, c0


def test_encoder_on_the_hand_written_projections_equals_the_library_path(cuda, monkeypatch):
    """Training mode, dropout ON, bf16, d_model 512, 3 layers: same seeds -> same masks whether the plain projections of a
    layer (QKV forward, linear2 forward, the three input gradients, the heads) run on cwlt_gemm_bf16 or on hipBLASLt;
    losses and every parameter gradient agree to what a differing bf16 rounding of the products can do."""
    import os
    import sys
    import numpy as np
    HERE = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from fill import fill_params
    from rlmg_amd.dqn_policy import config, model
    n_class = [56, 135, 18, 87, 18, 25]
    B, T = 2, 256
    g = torch.Generator().manual_seed(12)
    x = torch.stack([torch.randint(0, n, (B, T), generator=g) for n in n_class], -1).to(cuda)
    y = torch.stack([torch.randint(0, n, (B, T), generator=g) for n in n_class], -1).to(cuda)
    mask = torch.ones(B, T, device=cuda)
    monkeypatch.setattr(ops, "LAYER_C", False)                  # the per-op layer (512 rows would take the one-call path)
    monkeypatch.setattr(ops, "LINEAR_LN_MIN_ROWS", 1 << 40)    # the residual blocks as GEMM + LayerNorm kernel: their GEMMs
    runs = {}                                                   # (out-projection, linear2) then go through this switch too
    old = dict(config.AgentConfig)
    config.AgentConfig.update({"D_MODEL": 512, "N_LAYER": 3, "N_HEAD": 8})
    try:
        for hand in (True, False):
            monkeypatch.setattr(ops, "GEMM_BF16", hand)
            net = fill_params(model.LinearTransformer(n_class), seed=57).to(cuda).train()
            net.compute_dtype = torch.bfloat16
            torch.manual_seed(77)                               # ops.next_seed() draws from torch's CPU generator
            calls = []
            real = ops.gemm_bf16
            monkeypatch.setattr(ops, "gemm_bf16", lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1])
            losses = net.train_step(x, y, mask)
            (sum(losses) / 6).backward()
            monkeypatch.setattr(ops, "gemm_bf16", real)
            # per layer: QKV, out-projection, linear2 forward; linear1 / out-projection / QKV input gradients; + the heads
            # forward and the heads' input gradient
            # (+ in_linear's forward and input gradient where the front is not the one-pass embed kernel)
            assert len(calls) in ((3 * 6 + 2, 3 * 6 + 4) if hand else (0,)), len(calls)
            runs[hand] = ([l.item() for l in losses],
                          {n_: p.grad.detach().double().cpu() for n_, p in net.named_parameters() if p.grad is not None})
    finally:
        config.AgentConfig.update(old)
    la, lb = np.array(runs[True][0]), np.array(runs[False][0])
    assert np.abs(la - lb).max() <= 2e-3 * np.abs(lb).max(), (la, lb)
    ga, gb = runs[True][1], runs[False][1]
    assert ga.keys() == gb.keys()
    # two bf16 schedules of the same step (test_gemm_ln_gpu.py): 7.1 % of the tensor's own norm, with a floor of 2e-4 of
    # the rms tensor norm for gradients that are rounding noise in both (key_projection.bias: norm 7e-6 beside 1e-2 .. 1
    # for the rest; its two noise draws differ by ~10 % of that)
    rms = float(np.sqrt(np.mean([gb[k].norm().item() ** 2 for k in gb])))
    for k in gb:
        d = (ga[k] - gb[k]).norm().item()
        assert d <= 7.1e-2 * max(gb[k].norm().item(), 2e-4 * rms), (k, d, gb[k].norm().item(), rms)
