"""Micro-benchmark of individual libcwlt entry points at the bench shape (GPU box only).
usage: python tools/bench_kernels.py [B] [T]"""
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    dev = torch.device("cuda:0")
    H, D = 8, 64
    for dt in (torch.bfloat16, torch.float32):
        s = 2 if dt == torch.bfloat16 else 4
        qkv = torch.randn(B, T, 3, H, D, device=dev).to(dt)
        dout = torch.randn(B, T, H, D, device=dev).to(dt)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        _, _, _, out, zinv, fin = ops.cla_fwd(q, k, v, final_state=True)
        R = B * T
        t = timeit(lambda: ops.cla_fwd(q, k, v))
        print("%-8s cla_fwd   %8.1f us  %7.1f GB/s (algorithmic)" % (dt, t * 1e3, R * 4 * 512 * s / t / 1e6))
        t = timeit(lambda: ops.cla_bwd(q, k, v, out, zinv, dout))
        print("%-8s cla_bwd   %8.1f us  %7.1f GB/s (algorithmic, 7 streams; dkdv + dq pair)" % (dt, t * 1e3, R * 7 * 512 * s / t / 1e6))
        if fin is not None:
            t = timeit(lambda: ops.cla_fwd(q, k, v, final_state=True))
            print("%-8s cla_fwd   %8.1f us  with the final-state hand-over" % (dt, t * 1e3))
            t = timeit(lambda: ops.cla_bwd(q, k, v, out, zinv, dout, want_colsum=True, final_state=fin))
            print("%-8s cla_bwd   %8.1f us  %7.1f GB/s (algorithmic, 7 streams; one sweep)" % (dt, t * 1e3, R * 7 * 512 * s / t / 1e6))


def bench_wgrad():
    dev = torch.device("cuda:0")
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    for N1, N2 in ((2048, 512), (512, 2048), (512, 512), (1536, 512)):
        a = torch.randn(M, N1, device=dev).bfloat16()
        b = torch.randn(M, N2, device=dev).bfloat16()
        t0 = timeit(lambda: torch.mm(a.t(), b))
        t1 = timeit(lambda: ops.wgrad(a, b))
        ref = torch.mm(a.float().t(), b.float())
        err = (ops.wgrad(a, b) - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-3, err
        fl = 2.0 * M * N1 * N2
        print("wgrad %4dx%4d  torch.mm %7.1f us (%6.0f TF)   cwlt %7.1f us (%6.0f TF)" %
              (N1, N2, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, fl / t1 / 1e9))


def bench_ffn2_dgrad():
    """FFN backward: hipBLASLt NT GEMM (dg = dy . W2) + cwlt_bias_gelu_dropout_bwd   vs   cwlt_gemm_nt_mul."""
    dev = torch.device("cuda:0")
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
    dy = torch.randn(M, 512, device=dev).bfloat16()
    w2 = (torch.randn(512, 2048, device=dev) * 0.05).bfloat16()          # linear2.weight (out 512, in 2048)
    w2t = w2.t().contiguous()                                             # (2048, 512) = [N][K]
    h = torch.randn(M, 2048, device=dev).bfloat16()
    b = torch.randn(2048, device=dev) * 0.1
    t0 = timeit(lambda: torch.mm(dy, w2t.t()))
    dg = torch.mm(dy, w2t.t())
    t1 = timeit(lambda: ops.gelu_bwd(dg, h, b, 0.1, 77))
    t2 = timeit(lambda: ops.gelu_fwd(h, b, 0.1, 77))
    hh = h.clone()
    t3 = timeit(lambda: ops.gelu_fwd(hh, b, 0.1, 77, gd_inplace=True))
    gd = h.clone()
    ops.gelu_fwd(gd, b, 0.1, 77, gd_inplace=True)
    t4 = timeit(lambda: ops.gemm_nt_mul(dy, w2t, gd))
    t5 = timeit(lambda: ops.gemm_nt_mul(dy, w2t, gd, want_colsum=False))
    fl = 2.0 * M * 512 * 2048
    by = M * (512 + 2 * 2048) * 2
    print("M=%d  unfused: mm %.1f us (%.0f TF) + gelu_bwd %.1f us = %.1f us" % (M, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3,
                                                                             (t0 + t1) * 1e3))
    print("       fused gemm_nt_mul %.1f us (%.0f TF, %.0f GB/s algorithmic); without column sums %.1f us" %
          (t4 * 1e3, fl / t4 / 1e9, by / t4 / 1e6, t5 * 1e3))
    print("       forward activation: plain %.1f us, with gd in place %.1f us (+%.1f)" % (t2 * 1e3, t3 * 1e3,
                                                                                           (t3 - t2) * 1e3))
    print("       net per layer: %.1f us saved" % ((t0 + t1 - t4 - (t3 - t2)) * 1e3))


def bench_ffn1():
    """FFN forward: hipBLASLt GEMM (h = x . W1^T) + cwlt_bias_gelu_dropout_fwd (g, gd)   vs   cwlt_gemm_nt_bias_gelu_dropout."""
    dev = torch.device("cuda:0")
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
    x = torch.randn(M, 512, device=dev).bfloat16()
    w1 = (torch.randn(2048, 512, device=dev) * 0.05).bfloat16()
    b = torch.randn(2048, device=dev) * 0.1
    t0 = timeit(lambda: torch.mm(x, w1.t()))
    h = torch.mm(x, w1.t())
    t1 = timeit(lambda: ops.gelu_fwd(h, b, 0.1, 77, gd_inplace=True))
    t2 = timeit(lambda: ops.ffn1_gelu_dropout(x, w1, b, 0.1, 77))
    var = os.environ.get("CWLT_GEMM_VARIANT")              # cwlt_gemm_bf16_tune bits (131072: g stored with the default policy)
    if var:
        from rlmg_amd import _lib
        _lib.load().cwlt_gemm_bf16_tune(int(var), None)
        t3 = [timeit(lambda: ops.ffn1_gelu_dropout(x, w1, b, 0.1, 77)) for _ in range(3)]
        _lib.load().cwlt_gemm_bf16_tune(-1, None)
        t4 = [timeit(lambda: ops.ffn1_gelu_dropout(x, w1, b, 0.1, 77)) for _ in range(3)]
        print("       variant %s: %s us; default again: %s us" % (var, ["%.1f" % (t * 1e3) for t in t3],
                                                                  ["%.1f" % (t * 1e3) for t in t4]))
    fl = 2.0 * M * 512 * 2048
    print("M=%d  unfused: mm %.1f us (%.0f TF) + activation with gd %.1f us = %.1f us" %
          (M, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, (t0 + t1) * 1e3))
    print("       fused cwlt_gemm_nt_bias_gelu_dropout %.1f us (%.0f TF, %.0f GB/s of output)" %
          (t2 * 1e3, fl / t2 / 1e9, M * (512 + 2 * 2048) * 2 / t2 / 1e6))


def bench_linear_ln():
    """Residual block: hipBLASLt GEMM (bf16 out, bias) + cwlt_add_dropout_layernorm_fwd   vs   cwlt_gemm_nt_bias_dropout_add_layernorm."""
    dev = torch.device("cuda:0")
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
    for K in (512, 2048):
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(512, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(512, device=dev) * 0.1
        bb = b.bfloat16()
        x = torch.randn(M, 512, device=dev).bfloat16()
        gm, bt = torch.randn(512, device=dev), torch.randn(512, device=dev)
        t0 = timeit(lambda: torch.addmm(bb, a, w.t()))
        o = torch.addmm(bb, a, w.t())
        t1 = timeit(lambda: ops.ln_fwd(x, o, gm, bt, p=0.1, seed=5))
        t2 = timeit(lambda: ops.linear_ln(a, w, b, x, gm, bt, p=0.1, seed=5))
        fl = 2.0 * M * 512 * K
        by = M * (K + 3 * 512) * 2
        print("M=%d K=%d  unfused: addmm %.1f us (%.0f TF) + ln_fwd %.1f us = %.1f us   fused %.1f us (%.0f TF, %.0f GB/s algorithmic)"
              % (M, K, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, (t0 + t1) * 1e3, t2 * 1e3, fl / t2 / 1e9, by / t2 / 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "linln":
        from rlmg_amd import gemm_tuning
        gemm_tuning.enable()
        bench_linear_ln()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ffn1":
        from rlmg_amd import gemm_tuning
        gemm_tuning.enable()
        bench_ffn1()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ffn2d":
        from rlmg_amd import gemm_tuning
        gemm_tuning.enable()
        bench_ffn2_dgrad()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wgrad":
        bench_wgrad()
        sys.exit(0)
    main()
