"""GPU: the one-call encoder layer (csrc/layer.hip: cwlt_encoder_layer_fwd / _bwd) and the kernels it adds
(cwlt_gemm_bf16_small, cwlt_transpose_bf16_many) -- fast_transformers' post-LN TransformerEncoderLayer with causal linear
attention as built at /root/reference/dqn_policy/model.py:128-137 and called at :232, at the row counts of the reference's
own RL updates (30 windows x 50 tokens, dqn_policy/IRL_dqn_train.py:267-345).

* the small projection GEMM against the f64 product of the same bf16 operands (one rounding: 2^-7 x the largest value);
* the layer call against the per-op path running the SAME kernels (CWLT_GEMM_SMALL_PER_OP): outputs, input gradient and
  every weight gradient BIT FOR BIT, dropout on and off, with and without the scan's one-sweep backward (few long
  sequences are cut into segments); the Q/K/V bias gradients are summed over sequences in a different (fixed) order;
* the layer call against the library-GEMM per-op path (hipBLASLt): to what a differing bf16 rounding of the products does.
"""
import pytest
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import _lib, encoder, ops

pytestmark = pytest.mark.gpu
BF16_TOL = 2.0 ** -7


def _operands(M, N, K, seed, lda=None, ldw=None):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, lda or K, generator=g).bfloat16()[:, :K]
    w = (torch.randn(N, ldw or K, generator=g) * (2.0 / K ** 0.5)).bfloat16()[:, :K]
    bias = torch.randn(N, generator=g) * 0.3
    c0 = torch.randn(M, N, generator=g).bfloat16()
    return a, w, bias, c0


@pytest.mark.parametrize("M", [1, 37, 64, 1500])
@pytest.mark.parametrize("N,K", [(1536, 512), (512, 512), (512, 2048), (512, 1536), (2048, 512), (384, 512), (24, 96),
                                 (72, 320)])
def test_gemm_small_matches_the_f64_product(cuda, M, N, K):
    a, w, bias, c0 = _operands(M, N, K, 1000 * M + N + K)
    ad, wd, bd = a.to(cuda), w.to(cuda), bias.to(cuda)
    for use_bias in (False, True):
        for acc in (False, True):
            out = c0.to(cuda).clone() if acc else None
            got = ops.gemm_bf16_small(ad, wd, bd if use_bias else None, out=out, accumulate=acc)
            ref = a.double() @ w.double().t()
            if use_bias:
                ref = ref + bias.double()
            if acc:
                ref = ref + c0.double()
            sc = max(1.0, ref.abs().max().item())
            err = (got.double().cpu() - ref).abs().max().item()
            assert err <= BF16_TOL * sc, (use_bias, acc, err, sc)


def test_gemm_small_strided_and_untouched_neighbours(cuda):
    M, N, K = 203, 136, 640
    g = torch.Generator().manual_seed(5)
    ad = torch.randn(M, K + 24, generator=g).bfloat16().to(cuda)[:, :K]
    wd = (torch.randn(N, K + 8, generator=g) * (2.0 / K ** 0.5)).bfloat16().to(cuda)[:, :K]
    bias = torch.randn(N, generator=g) * 0.3
    a, w = ad.cpu(), wd.cpu()
    assert ad.stride(0) == K + 24 and wd.stride(0) == K + 8
    big = torch.full((M + 2, N + 16), 7.0, dtype=torch.bfloat16, device=cuda)
    out = big[1:M + 1, 8:8 + N]
    ops.gemm_bf16_small(ad, wd, bias.to(cuda), out=out)
    ref = a.double() @ w.double().t() + bias.double()
    assert (out.double().cpu() - ref).abs().max().item() <= BF16_TOL * max(1.0, ref.abs().max().item())
    big2 = big.clone()
    big2[1:M + 1, 8:8 + N] = 7.0
    assert (big2 == 7.0).all()                                  # nothing outside the view was written
    lib = _lib.load()
    p = lambda t: _lib.dev(t)
    assert lib.cwlt_gemm_bf16_small(p(ad), p(wd), None, p(big), M, N, 48, K + 24, K + 8, N + 16, 0, None) == 1001   # K % 32
    assert lib.cwlt_gemm_bf16_small(p(ad), p(wd), None, p(big), M, 12, K, K + 24, K + 8, N + 16, 0, None) == 1001   # N % 8


@pytest.mark.parametrize("M", [1, 50, 255])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_small_ffn_forward_equals_the_two_kernel_chain_bit_for_bit(cuda, M, p):
    """cwlt_gemm_bf16_small_gelu (linear1 + bias + GELU + dropout, and the backward's factor gd, on the split-K small tiles)
    against cwlt_gemm_bf16_small followed by cwlt_bias_gelu_dropout_fwd: the same rounded product, the same dropout
    stream -- and against the f64 GELU of that product where nothing is dropped."""
    a, w, bias, _ = _operands(M, 2048, 512, 31 * M + 7)
    ad, wd, bd = a.to(cuda), w.to(cuda), bias.to(cuda)
    g, gd = ops.gemm_bf16_small_gelu(ad, wd, bd, p, 1234)
    h = ops.gemm_bf16_small(ad, wd)
    hd = h.double().cpu()
    g_ref = ops.gelu_fwd(h, bd, p, 1234, gd_inplace=True)       # h now holds gd
    assert torch.equal(g, g_ref) and torch.equal(gd, h)
    g_only, none = ops.gemm_bf16_small_gelu(ad, wd, bd, p, 1234, want_gd=False)
    assert none is None and torch.equal(g_only, g)
    x = hd + bias.double()
    exact = 0.5 * x * (1.0 + torch.erf(x / 2.0 ** 0.5)) / (1.0 - p)
    kept = (g != 0).cpu() | (exact.abs() < 1e-3)
    assert ((g.double().cpu() - exact).abs()[kept] <= 2.0 ** -7 * exact.abs().clamp(min=1.0)[kept]).all()
    if p > 0:
        frac = 1.0 - (g != 0).float().mean().item()
        assert M < 50 or abs(frac - p) < 0.02, frac


def test_transpose_many(cuda):
    g = torch.Generator().manual_seed(3)
    mats = [torch.randn(r, c, generator=g).bfloat16().to(cuda) for r, c in ((1536, 512), (512, 512), (2048, 512),
                                                                            (512, 2048), (70, 130), (64, 8))]
    cache = ops.LayerCache(mats, owner=None)
    cache.refresh_transposed()
    for m, v in zip(mats, cache.views):
        assert v.shape == (m.shape[1], m.shape[0])
        assert torch.equal(v, m.t())


def test_cast_many_equals_the_multi_tensor_copy(cuda, monkeypatch):
    """ops.ShadowSet's one-launch refresh (cwlt_cast_bf16_many) against torch._foreach_copy_: the same bf16 values, odd
    sizes and unaligned slices included; follows a parameter whose storage moved."""
    g = torch.Generator().manual_seed(2)
    ps = [torch.nn.Parameter(torch.randn(sh, generator=g).to(cuda)) for sh in ((512, 512), (1536,), (7, 3), (2048, 512), (5,),
                                                                                (513,), (1, 1))]
    groups = [(ps[0],), (ps[1],), (ps[2], ps[2]), (ps[3],), (ps[4], ps[5], ps[4]), (ps[6],)]
    fast = ops.ShadowSet(groups, torch.bfloat16)
    bufs = [b.clone() for b in fast.refresh()]
    assert fast._fast[1] is not None                            # the one-launch path was taken
    monkeypatch.setattr(ops, "CAST_MANY", False)
    slow = ops.ShadowSet(groups, torch.bfloat16)
    for a, b in zip(bufs, slow.refresh()):
        assert torch.equal(a, b)
    monkeypatch.setattr(ops, "CAST_MANY", True)
    with torch.no_grad():
        ps[3].data = torch.randn(2048, 512, generator=g).to(cuda)   # new storage
        ps[0].mul_(2.0)
    got = fast.refresh()
    assert torch.equal(got[3], ps[3].detach().bfloat16()) and torch.equal(got[0], ps[0].detach().bfloat16())


def _encoder(cuda, n_layers, p, seed):
    enc = encoder.TransformerEncoderBuilder.from_kwargs(
        n_layers=n_layers, n_heads=8, query_dimensions=64, value_dimensions=64, feed_forward_dimensions=2048,
        activation="gelu", dropout=p, attention_type="causal-linear").get()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for q in enc.parameters():
            q.copy_(torch.randn(q.shape, generator=g) * (1.0 / q.shape[1] ** 0.5 if q.dim() == 2 else 0.2))
        for layer in enc.layers:                                # LayerNorm weights around one
            layer.norm1.weight.add_(1.0)
            layer.norm2.weight.add_(1.0)
        enc.norm.weight.add_(1.0)
    return enc.to(cuda)


def _run(enc, x, dy, mode, monkeypatch, train=True):
    """mode: 'stack' one call per encoder pass; 'c' one call per layer; 'small' per-op on the same kernels; 'lib' per-op
    on hipBLASLt."""
    monkeypatch.setattr(ops, "LAYER_C", mode in ("c", "stack"))
    monkeypatch.setattr(ops, "LAYER_C_STACK", mode == "stack")
    monkeypatch.setattr(ops, "GEMM_SMALL_PER_OP", mode == "small")
    enc.train(train)
    for q in enc.parameters():
        q.grad = None
    torch.manual_seed(99)                                       # ops.next_seed(): the dropout streams
    xin = x.clone().requires_grad_(True)
    y = enc(xin, attn_mask=encoder.TriangularCausalMask(x.shape[1], device=x.device))
    y.backward(dy)
    torch.cuda.synchronize()
    return y.detach(), xin.grad.detach(), {n: q.grad.detach().clone() for n, q in enc.named_parameters()}


@pytest.mark.parametrize("N,L", [(30, 50), (5, 77), (2, 300), (1, 1)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_layer_call_equals_the_per_op_path_bit_for_bit(cuda, monkeypatch, N, L, p):
    enc = _encoder(cuda, 2, p, seed=21)
    g = torch.Generator().manual_seed(N * 1000 + L)
    x = torch.randn(N, L, 512, generator=g).bfloat16().to(cuda)
    dy = (torch.randn(N, L, 512, generator=g) * 0.1).bfloat16().to(cuda)
    calls = []
    real, real_stack = ops.encoder_layer_fwd, ops.encoder_fwd
    monkeypatch.setattr(ops, "encoder_layer_fwd", lambda st: (calls.append("layer"), real(st))[1])
    monkeypatch.setattr(ops, "encoder_fwd", lambda arr, n: (calls.append("stack"), real_stack(arr, n))[1])
    yc, dxc, gc = _run(enc, x, dy, "c", monkeypatch)
    assert calls == ["layer", "layer"]                          # the layers did go through the one-call path
    yk, dxk, gk = _run(enc, x, dy, "stack", monkeypatch)
    assert calls == ["layer", "layer", "stack"]
    ys, dxs, gs = _run(enc, x, dy, "small", monkeypatch)
    assert len(calls) == 3
    for y_, dx_, g_ in ((yc, dxc, gc), (yk, dxk, gk)):
        assert torch.equal(y_, ys)
        assert torch.equal(dx_, dxs)
        for n in gs:
            if n.endswith(("query_projection.bias", "key_projection.bias", "value_projection.bias")) and N > 1:
                # column sums over sequences: ascending order in C, torch's reduction per-op
                assert torch.allclose(g_[n], gs[n], rtol=1e-5, atol=1e-6 * max(1.0, gs[n].abs().max().item())), n
            else:
                assert torch.equal(g_[n], gs[n]), n
    for n in gs:
        assert torch.equal(gc[n], gk[n]), n                     # per-layer and per-stack calls: the same launches


def test_layer_call_without_a_backward_equals_the_per_op_path(cuda, monkeypatch):
    """Inference / rollout passes (torch.no_grad, eval mode): no saved state is kept, the output is the same to bf16
    rounding."""
    enc = _encoder(cuda, 2, 0.1, seed=4).eval()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(30, 50, 512, generator=g).bfloat16().to(cuda)
    outs = {}
    for mode in ("c", "stack", "small"):
        monkeypatch.setattr(ops, "LAYER_C", mode in ("c", "stack"))
        monkeypatch.setattr(ops, "LAYER_C_STACK", mode == "stack")
        monkeypatch.setattr(ops, "GEMM_SMALL_PER_OP", mode == "small")
        with torch.no_grad():
            outs[mode] = enc(x, attn_mask=encoder.TriangularCausalMask(50, device=cuda))
    assert torch.equal(outs["c"], outs["stack"])
    # without a backward the per-op path takes linear1 as a plain product followed by the activation kernel, the layer
    # call keeps the one-kernel form: the two sum K in different orders before the same rounding
    assert (outs["c"].float() - outs["small"].float()).abs().max().item() <= 2e-2 * outs["small"].float().abs().max().item()


def test_layer_call_close_to_the_library_gemm_path(cuda, monkeypatch):
    """hipBLASLt's products round differently: outputs to 2 % of their scale, gradients to 7.1 % of each tensor's norm (the
    bound of two bf16 schedules of one step, tests/test_gemm_ln_gpu.py)."""
    enc = _encoder(cuda, 3, 0.1, seed=11)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(30, 50, 512, generator=g).bfloat16().to(cuda)
    dy = (torch.randn(30, 50, 512, generator=g) * 0.1).bfloat16().to(cuda)
    yc, dxc, gc = _run(enc, x, dy, "stack", monkeypatch)
    yl, dxl, gl = _run(enc, x, dy, "lib", monkeypatch)
    assert (yc.float() - yl.float()).abs().max().item() <= 2e-2 * yl.float().abs().max().item()
    assert (dxc.float() - dxl.float()).norm().item() <= 7.1e-2 * dxl.float().norm().item()
    rms = (sum(v.double().norm().item() ** 2 for v in gl.values()) / len(gl)) ** 0.5
    for n in gl:
        d = (gc[n].double() - gl[n].double()).norm().item()
        assert d <= 7.1e-2 * max(gl[n].double().norm().item(), 2e-4 * rms), (n, d)


def test_plan_and_refusals(cuda):
    lib = _lib.load()
    import ctypes
    plan = _lib.EncoderLayerPlan()
    assert lib.cwlt_encoder_layer_plan(30, 50, 512, 2048, 8, 0.1, 1, ctypes.byref(plan)) == 0
    R = 1500
    assert plan.saved_bytes >= R * (3 * 512 + 4 * 512 + 2 * 2048) * 2
    assert plan.grad_floats >= 3 * 512 * 512 + 512 * 512 + 2 * 2048 * 512
    offs = list(plan.grad_off)
    assert len(set(offs)) == 12 and all(o % 4 == 0 for o in offs)
    assert lib.cwlt_encoder_layer_plan(30, 50, 256, 2048, 8, 0.1, 1, ctypes.byref(plan)) == 1001      # d_model
    assert lib.cwlt_encoder_layer_plan(30, 50, 512, 2000, 8, 0.1, 1, ctypes.byref(plan)) == 1001      # d_ff % 256
    st = _lib.EncoderLayer(n_seq=30, len=50, d_model=512, d_ff=2048, n_heads=8, want_backward=1, p_drop=0.1)
    assert lib.cwlt_encoder_layer_fwd(ctypes.byref(st), None) == 1001                                  # null pointers
    assert lib.cwlt_encoder_layer_bwd(ctypes.byref(st), None) == 1001


def test_transposed_weights_once_per_forward_equal_the_per_use_copies(cuda, monkeypatch):
    """Training sizes (per-op layer, >= 16 384 token rows): the input-gradient products read transposed weight copies made
    by ONE launch per forward (ops.LayerCache.refresh_transposed) instead of a copy kernel per use: same bytes, so the
    same results bit for bit."""
    enc = _encoder(cuda, 2, 0.1, seed=9)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(16, 1024, 512, generator=g).bfloat16().to(cuda)
    dy = (torch.randn(16, 1024, 512, generator=g) * 0.1).bfloat16().to(cuda)
    monkeypatch.setattr(ops, "DGRAD_WT_CACHE", True)
    ya, dxa, ga = _run(enc, x, dy, "lib", monkeypatch)
    assert enc._tcache is not None
    monkeypatch.setattr(ops, "DGRAD_WT_CACHE", False)
    yb, dxb, gb = _run(enc, x, dy, "lib", monkeypatch)
    assert torch.equal(ya, yb) and torch.equal(dxa, dxb)
    for n in gb:
        assert torch.equal(ga[n], gb[n]), n

