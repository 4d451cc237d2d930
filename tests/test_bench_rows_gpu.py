"""GPU: the kernels that carry 2 GiB operands in the bench step, AT the bench's own row count R = 512 x 1024 = 524 288
(bf16, D = 512, F = 2048).  The g / gd / dh tensors are 2 GiB each here: every byte offset past 2^31 is exercised.
The f64 chain cannot run at this size, so each test checks
  * the LAST 4 096 rows (the ones with the largest offsets) against the f64 chain of the op on the same rounded inputs,
  * a whole-tensor identity that needs no oracle (the dropout stream of the unfused pair, column sums recomputed on the
    GPU in f64, the LayerNorm gradient sums).
Tolerances as in tests/test_ops_bf16_gpu.py: 2^-7 x scale for a tensor stored in bf16, 1e-3 x scale for f32 row sums
(here over 5 x 10^5 rows: 5e-3)."""
import pytest
import torch
import torch.nn.functional as F

import rlmg_amd  # noqa: F401
from rlmg_amd import ops

pytestmark = pytest.mark.gpu
R, D, FF, TAIL = 524288, 512, 2048, 4096
BF16_TOL = 2.0 ** -7


def _close(got, ref, tol, what):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= tol * scale, "%s: max abs err %.3e > %.3e (scale %.2f)" % (what, err, tol * scale, scale)


def _colsum64(fn, rows, step=32768):
    """sum over rows of fn(lo, hi) (a (hi - lo, C) tensor), accumulated in f64 on the GPU."""
    tot = None
    for lo in range(0, rows, step):
        part = fn(lo, min(rows, lo + step)).double().sum(0)
        tot = part if tot is None else tot + part
    return tot


def test_ffn1_one_kernel_forward_at_bench_rows(cuda):
    g0 = torch.Generator(device=cuda).manual_seed(11)
    x = torch.randn(R, D, device=cuda, generator=g0).bfloat16()
    w = (torch.randn(FF, D, device=cuda, generator=g0) * 0.08).bfloat16()
    b = torch.randn(FF, device=cuda, generator=g0) * 0.2
    p, seed = 0.1, 20261004
    assert ops.ffn1_fused_supported(x, w, b)
    g, gd = ops.ffn1_gelu_dropout(x, w, b, p, seed)
    assert g.shape == (R, FF) and g.numel() * 2 >= 2 ** 31
    # whole tensor: the dropout stream of the two-kernel path (same seed, same element index), keep rate
    h = torch.mm(x, w.t())
    g2 = ops.gelu_fwd(h, b, p, seed, gd_inplace=True)                 # h now holds gd
    keep = (g != 0) | (gd != 0)
    keep2 = (g2 != 0) | (h != 0)
    assert abs(1.0 - keep.float().mean().item() - p) < 1e-3
    # the masks can differ only where an element is kept but its value AND derivative round to zero in one path
    assert (keep != keep2).float().mean().item() < 1e-4
    eq = (g == g2) & (gd == h)
    assert eq.float().mean().item() > 0.97                            # both GEMMs rounded the pre-activation alike
    del g2, h, keep2, eq
    # last rows against the f64 chain
    xt, gt, gdt, kt = x[-TAIL:].cpu(), g[-TAIL:].cpu(), gd[-TAIL:].cpu(), keep[-TAIL:].cpu()
    pre = (xt.double() @ w.cpu().double().t()).bfloat16().double() + b.cpu().double()
    pr = pre.clone().requires_grad_(True)
    F.gelu(pr).sum().backward()
    scale = 1.0 / (1.0 - p)
    tol = 2 * BF16_TOL * max(1.0, pre.abs().max().item()) * scale
    assert (gt.double() - F.gelu(pre) * kt.double() * scale).abs().max().item() <= tol
    assert (gdt.double() - pr.grad * kt.double() * scale).abs().max().item() <= tol


def test_ffn_backward_gemm_at_bench_rows(cuda):
    g0 = torch.Generator(device=cuda).manual_seed(12)
    dy = torch.randn(R, D, device=cuda, generator=g0).bfloat16()
    w2t = (torch.randn(FF, D, device=cuda, generator=g0) * 0.05).bfloat16()      # linear2.weight transposed: (F, D)
    gd = torch.randn(R, FF, device=cuda, generator=g0).bfloat16()
    assert ops.gemm_nt_mul_supported(dy, w2t, gd)
    c, cs = ops.gemm_nt_mul(dy, w2t, gd)
    assert c.shape == (R, FF)
    # last rows against the f64 chain (product rounded to bf16, then multiplied)
    prod = dy[-TAIL:].cpu().double() @ w2t.cpu().double().t()
    ref = prod * gd[-TAIL:].cpu().double()
    scale = max(1.0, prod.abs().max().item() * gd[-TAIL:].double().abs().max().item())
    assert (c[-TAIL:].cpu().double() - ref).abs().max().item() <= 2 * BF16_TOL * scale
    # column sums over ALL rows (the linear1 bias gradient): f64 sums of bf16(product) * gd, recomputed on the GPU
    cs_ref = _colsum64(lambda lo, hi: torch.mm(dy[lo:hi], w2t.t()).float() * gd[lo:hi].float(), R)
    _close(cs, cs_ref, 5e-3, "column sums over 524 288 rows")
    # and the stored output agrees with them up to its own bf16 rounding, summed: |sum err| <= rows x 2^-9 x max|c| is
    # far too loose to say anything -- the random-sign rounding errors add up like sqrt(rows)
    cs_out = _colsum64(lambda lo, hi: c[lo:hi], R)
    bound = 8 * (R ** 0.5) * 2.0 ** -9 * c[:65536].float().abs().max().item()
    assert (cs_out - cs.double()).abs().max().item() <= bound


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_add_dropout_layernorm_at_bench_rows(cuda, p):
    g0 = torch.Generator(device=cuda).manual_seed(13)
    a = torch.randn(R, D, device=cuda, generator=g0).bfloat16()
    x = torch.randn(R, D, device=cuda, generator=g0).bfloat16()
    gamma = torch.randn(D, device=cuda, generator=g0)
    beta = torch.randn(D, device=cuda, generator=g0)
    dy = torch.randn(R, D, device=cuda, generator=g0).bfloat16()
    seed = 777
    s, y, mean, rstd = ops.ln_fwd(x, a, gamma, beta, p=p, seed=seed)
    # the keep mask of (seed, element index): every kernel shares it (test_fullsize_gpu.py)
    keep = (ops.posenc_dropout(torch.ones(R, D, device=cuda), None, 1, p=p, seed=seed) != 0) if p > 0 else None
    ks = 1.0 / (1.0 - p)
    if p > 0:
        assert abs(1.0 - keep.float().mean().item() - p) < 1e-3
        # whole tensor: s - x is either 0 (dropped) or a / (1 - p), up to the rounding of s
        d = s.float() - x.float()
        want = torch.where(keep, a.float() * ks, torch.zeros((), device=cuda))
        assert (d - want).abs().max().item() <= 2 * BF16_TOL * max(1.0, s.float().abs().max().item())
        del d, want
    # last rows against the f64 chain
    kt = keep[-TAIL:].cpu().double() * ks if p > 0 else 1.0
    s_ref = a[-TAIL:].cpu().double() * kt + x[-TAIL:].cpu().double()
    y_ref = F.layer_norm(s_ref, (D,), gamma.cpu().double(), beta.cpu().double(), 1e-5)
    _close(s[-TAIL:], s_ref, BF16_TOL, "s (tail)")
    _close(y[-TAIL:], y_ref, BF16_TOL, "y (tail)")
    _close(mean[-TAIL:], s_ref.mean(-1), 1e-3, "mean (tail)")
    # backward on the stored s: tail rows vs f64 autograd, parameter / bias gradients vs f64 sums over ALL rows
    ds, da, dg, db, dbias = ops.ln_bwd(dy, None, s, gamma, mean, rstd, p=p, seed=seed)
    sr = s[-TAIL:].cpu().double().requires_grad_(True)
    F.layer_norm(sr, (D,), gamma.cpu().double(), beta.cpu().double(), 1e-5).backward(dy[-TAIL:].cpu().double())
    _close(ds[-TAIL:], sr.grad, BF16_TOL, "ds (tail)")
    _close(da[-TAIL:], sr.grad * kt, BF16_TOL, "da (tail)")

    def xhat(lo, hi):
        return (s[lo:hi].double() - mean[lo:hi].double().unsqueeze(1)) * rstd[lo:hi].double().unsqueeze(1)

    _close(db, _colsum64(lambda lo, hi: dy[lo:hi], R), 5e-3, "dbeta over 524 288 rows")
    _close(dg, _colsum64(lambda lo, hi: dy[lo:hi].double() * xhat(lo, hi), R), 5e-3, "dgamma over 524 288 rows")
    # the bias gradient of the Linear in front = column sums of da, taken before da's rounding
    cs_da = _colsum64(lambda lo, hi: da[lo:hi], R)
    bound = 8 * (R ** 0.5) * 2.0 ** -9 * da[:65536].float().abs().max().item() + 5e-3 * cs_da.abs().max().item()
    assert (cs_da - dbias.double()).abs().max().item() <= bound
