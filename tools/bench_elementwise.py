"""Micro-benchmark of the HBM-bound elementwise entry points at the bench shape (GPU box only).
usage: python tools/bench_elementwise.py [B] [T]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import ops


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    dev = torch.device("cuda:0")
    R, D, F = B * T, 512, 2048
    h = torch.randn(R, F, device=dev).bfloat16()
    dg = torch.randn(R, F, device=dev).bfloat16()
    bias = torch.randn(F, device=dev)
    t = timeit(lambda: ops.gelu_fwd(h, bias, 0.1, 123))
    print("gelu_fwd   %8.1f us  %7.1f GB/s" % (t * 1e3, R * F * 2 * 2 / t / 1e6))
    t = timeit(lambda: ops.gelu_fwd(h, bias, 0.0, 0))
    print("gelu_fwd p=0 %6.1f us  %7.1f GB/s" % (t * 1e3, R * F * 2 * 2 / t / 1e6))
    t = timeit(lambda: ops.posenc_dropout(h, None, 1024, 0.0, 0))
    print("posenc p=0 (pure copy through VecIO) %6.1f us  %7.1f GB/s" % (t * 1e3, R * F * 2 * 2 / t / 1e6))
    t = timeit(lambda: ops.posenc_dropout(h, None, 1024, 0.1, 5))
    print("posenc p=.1 %6.1f us  %7.1f GB/s" % (t * 1e3, R * F * 2 * 2 / t / 1e6))
    t = timeit(lambda: ops.gelu_bwd(dg, h, bias, 0.1, 123))
    print("gelu_bwd   %8.1f us  %7.1f GB/s" % (t * 1e3, R * F * 2 * 3 / t / 1e6))
    x = torch.randn(R, D, device=dev).bfloat16()
    a = torch.randn(R, D, device=dev).bfloat16()
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    t = timeit(lambda: ops.ln_fwd(x, a, gam, bet, 1e-5, 0.1, 77))
    print("ln_fwd     %8.1f us  %7.1f GB/s" % (t * 1e3, R * D * 2 * 4 / t / 1e6))
    s, y, mean, rstd = ops.ln_fwd(x, a, gam, bet, 1e-5, 0.1, 77)
    t = timeit(lambda: ops.ln_bwd(a, None, s, gam, mean, rstd, 0.1, 77))
    print("ln_bwd     %8.1f us  %7.1f GB/s" % (t * 1e3, R * D * 2 * 4 / t / 1e6))
    widths, nrows = [128, 256, 64, 512, 256, 256], [56, 135, 18, 87, 18, 25]
    tables = [torch.randn(n, w, device=dev, requires_grad=True) for n, w in zip(nrows, widths)]
    tokens = torch.stack([torch.randint(0, n, (R,), device=dev) for n in nrows], -1)
    emb = ops.cw_embed(tokens, tables, torch.bfloat16)
    dout = torch.randn_like(emb)
    t = timeit(lambda: torch.autograd.grad(emb, tables, dout, retain_graph=True))
    print("embed_bwd  %8.1f us  %7.1f GB/s" % (t * 1e3, R * 1472 * 2 / t / 1e6))
    t = timeit(lambda: h.copy_(dg))
    print("torch copy %8.1f us  %7.1f GB/s  (2 streams, reference point)" % (t * 1e3, R * F * 2 * 2 / t / 1e6))


if __name__ == "__main__":
    main()
