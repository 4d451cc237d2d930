"""The one-pass input front (cwlt_cw_embed_proj_fwd / _bwd through ops.embed_proj) against the chain it replaces --
six embeddings * sqrt(d), cat, in_linear, + pe, dropout (dqn_policy/model.py:206-223, 90-92) -- and against the
library's own unfused kernels (same dropout stream for the same seed)."""
import math

import pytest
import torch
import torch.nn.functional as F

import rlmg_amd  # noqa: F401
from rlmg_amd import ops

pytestmark = pytest.mark.gpu
WIDTHS = (128, 256, 64, 512, 128, 128)
NROWS = (56, 135, 18, 87, 18, 25)


def _setup(shape, seed, D=512, widths=WIDTHS, nrows=NROWS, max_len=300):
    g = torch.Generator().manual_seed(seed)
    tabs = [torch.randn(n, w, generator=g) for n, w in zip(nrows, widths)]
    tok = torch.stack([torch.randint(0, n, shape, generator=g) for n in nrows], -1)
    w_in = torch.randn(D, sum(widths), generator=g) / math.sqrt(sum(widths))
    b_in = torch.randn(D, generator=g)
    pe = torch.randn(1, max_len, D, generator=g)
    dout = torch.randn(*shape, D, generator=g)
    return tabs, tok, w_in, b_in, pe, dout


def _reference(tabs, tok, w_in, b_in, pe, dout, keep=None, p=0.0):
    """f64 chain with autograd; `keep` (bool mask) stands in for the dropout draw."""
    tr = [t.double().requires_grad_(True) for t in tabs]
    wr, br = w_in.double().requires_grad_(True), b_in.double().requires_grad_(True)
    emb = torch.cat([F.embedding(tok[..., i], tr[i]) * math.sqrt(tr[i].shape[1]) for i in range(len(tr))], -1)
    y = F.linear(emb, wr, br) + pe[:, :tok.shape[1]].double()
    if keep is not None:
        y = y * keep.double() / (1 - p)
    y.backward(dout.double())
    return y.detach(), [t.grad for t in tr], wr.grad, br.grad


def _run(tabs, tok, w_in, b_in, pe, dout, p, seed, dtype, dev):
    td = [t.to(dev).requires_grad_(True) for t in tabs]
    wd, bd = w_in.to(dev).requires_grad_(True), b_in.to(dev).requires_grad_(True)
    y = ops.embed_proj(tok.to(dev), td, wd, bd, pe.to(dev), p, seed, dtype)
    y.backward(dout.to(dev).to(dtype))
    return y, [t.grad for t in td], wd.grad, bd.grad


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("shape", [(1, 1), (3, 50), (2, 257)])
def test_f32_front_equals_the_reference_chain(cuda, shape):
    args = _setup(shape, 11 + shape[1])
    y_ref, dt_ref, dw_ref, db_ref = _reference(*args)
    y, dt, dw, db = _run(*args, 0.0, 0, torch.float32, cuda)
    assert y.dtype == torch.float32 and y.shape == (*shape, 512)
    assert (y.double().cpu() - y_ref).abs().max().item() < 1e-4 * max(1.0, y_ref.abs().max().item())
    assert _rel(dw, dw_ref) < 1e-5 and _rel(db, db_ref) < 1e-5
    for a, b in zip(dt, dt_ref):
        assert _rel(a, b) < 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2.0 ** -7)])
def test_dropout_stream_is_the_unfused_kernels_and_gradients_follow_the_mask(cuda, dtype, tol):
    shape, p, seed = (4, 96), 0.25, 1234
    args = _setup(shape, 5)
    tabs, tok, w_in, b_in, pe, dout = args
    y, dt, dw, db = _run(*args, p, seed, dtype, cuda)
    # the library's own unfused chain with the same seed: embedding -> in_linear -> PositionalEncoding kernel
    emb = ops.cw_embed(tok.to(cuda), [t.to(cuda) for t in tabs], dtype)
    lin = F.linear(emb, w_in.to(cuda).to(dtype), b_in.to(cuda).to(dtype))
    y_unf = ops.PosEncDropoutFn.apply(lin, pe.to(cuda), p, seed)
    keep = y != 0
    assert torch.equal(keep, y_unf != 0)                          # same draw, element for element
    assert abs(keep.float().mean().item() - (1 - p)) < 0.01
    scale = y_unf.float().abs().max().item()
    assert (y.float() - y_unf.float()).abs().max().item() <= 4 * tol * scale
    # gradients against the f64 chain under that mask (bf16: the gradient rows are bf16-rounded dout, summed in f32)
    y_ref, dt_ref, dw_ref, db_ref = _reference(*args, keep=keep.cpu(), p=p)
    assert (y.double().cpu() - y_ref).abs().max().item() <= 2 * tol * max(1.0, y_ref.abs().max().item())
    gt = 1e-5 if dtype == torch.float32 else 6e-3
    assert _rel(dw, dw_ref) < gt and _rel(db, db_ref) < gt
    for a, b in zip(dt, dt_ref):
        assert _rel(a, b) < gt


def test_bf16_front_at_the_repo_vocabularies_many_rows_is_deterministic(cuda):
    shape = (8, 1024)
    args = _setup(shape, 3, max_len=1024)
    a = _run(*args, 0.1, 77, torch.bfloat16, cuda)
    b = _run(*args, 0.1, 77, torch.bfloat16, cuda)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    c = _run(*args, 0.1, 78, torch.bfloat16, cuda)
    assert not torch.equal(a[0], c[0])


def test_out_of_range_ids_are_clamped_and_long_sequences_refused(cuda):
    tabs, tok, w_in, b_in, pe, dout = _setup((2, 40), 9)
    bad = tok.clone()
    bad[0, 0, 1] = 10 ** 6
    bad[1, 3, 3] = -5
    good = tok.clone()
    good[0, 0, 1] = NROWS[1] - 1
    good[1, 3, 3] = 0
    y_bad = _run(tabs, bad, w_in, b_in, pe, dout, 0.0, 0, torch.float32, cuda)[0]
    y_good = _run(tabs, good, w_in, b_in, pe, dout, 0.0, 0, torch.float32, cuda)[0]
    assert torch.equal(y_bad, y_good)
    with pytest.raises(RuntimeError):
        _run(*_setup((1, 301), 1), 0.0, 0, torch.float32, cuda)
    with pytest.raises(TypeError):
        ops.embed_proj(tok.int().to(cuda), [t.to(cuda) for t in tabs], w_in.to(cuda), b_in.to(cuda), pe.to(cuda), 0.0, 0,
                       torch.float32)


def test_model_loss_and_gradients_with_and_without_the_one_pass_front(cuda, monkeypatch):
    """The whole model (small dims) through both fronts: same seeds -> same dropout draws, losses and gradients equal to
    rounding."""
    from rlmg_amd.dqn_policy import model, config
    old = dict(config.AgentConfig)
    config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    try:
        n_class = [56, 135, 18, 87, 18, 25]
        res = []
        for on in (True, False):
            monkeypatch.setattr(ops, "EMBED_PROJ", on)
            monkeypatch.setattr(ops, "EMBED_PROJ_MIN_ROWS", 1)
            torch.manual_seed(0)
            net = model.LinearTransformer(n_class).to(cuda).train()
            g = torch.Generator().manual_seed(1)
            x = torch.stack([torch.randint(0, n, (3, 64), generator=g) for n in n_class], -1).to(cuda)
            y = torch.stack([torch.randint(0, n, (3, 64), generator=g) for n in n_class], -1).to(cuda)
            mask = torch.ones(3, 64, device=cuda)
            torch.manual_seed(5)
            losses = net.train_step(x, y, mask)
            sum(losses).backward()
            res.append(([float(v.detach()) for v in losses],
                        {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
        (l_on, g_on), (l_off, g_off) = res
        assert max(abs(a - b) for a, b in zip(l_on, l_off)) < 1e-4
        assert set(g_on) == set(g_off)
        for k in g_on:
            assert _rel(g_on[k], g_off[k]) < 2e-4, k
    finally:
        config.AgentConfig.update(old)


@pytest.mark.parametrize("nrows,shape", [(NROWS, (3, 50)), (NROWS, (1, 1)), ((200, 300, 40, 150, 33, 64), (2, 100)),
                                         ((56, 135, 18, 87, 18, 25), (5, 1000))])
def test_bf16_backward_kernels_ragged_rows_and_large_vocabularies(cuda, nrows, shape):
    """bf16 gradients of the projected tables on row counts that are not multiples of the 64-row step, on a single row,
    across several 1 024-row id chunks of one split, and with vocabularies of more (attribute, 32-id tile) units than the
    shared-slab kernel's four waves take (27 > 16: the per-attribute kernel runs instead) -- against the f64 chain."""
    args = _setup(shape, 21 + shape[1], nrows=nrows, max_len=1000)
    y_ref, dt_ref, dw_ref, db_ref = _reference(*args)
    y, dt, dw, db = _run(*args, 0.0, 0, torch.bfloat16, cuda)
    assert (y.double().cpu() - y_ref).abs().max().item() <= 2.0 ** -7 * max(1.0, y_ref.abs().max().item())
    assert _rel(dw, dw_ref) < 6e-3 and _rel(db, db_ref) < 6e-3
    for a, b in zip(dt, dt_ref):
        assert _rel(a, b) < 6e-3
