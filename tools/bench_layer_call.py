"""Host and GPU time of one 12-layer encoder pass (forward + backward) at the row counts of the reference's RL updates,
one host call per encoder pass / per layer (csrc/layer.hip) against the per-op path.  GPU box.
usage: python tools/bench_layer_call.py [N L]...        default: 30 50  1 50  4 1024

  wall   : mean time of a forward + backward when passes are issued back to back (what a training loop sees)
  host   : the same with the GPU work removed from the critical path as far as possible -- time until the LAST launch of a
           pass has been issued (the stream is left to drain afterwards)
  gpu    : HIP-event time of the pass on the stream (>= the kernels' own time; equals wall when the GPU is the bound)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import encoder, ops


def build(dev):
    enc = encoder.TransformerEncoderBuilder.from_kwargs(
        n_layers=12, n_heads=8, query_dimensions=64, value_dimensions=64, feed_forward_dimensions=2048,
        activation="gelu", dropout=0.1, attention_type="causal-linear").get().to(dev).train()
    return enc


def one_pass(enc, x, dy, mask):
    xin = x.detach().requires_grad_(True)
    y = enc(xin, attn_mask=mask)
    y.backward(dy)


def measure(enc, x, dy, mask, n=20):
    for _ in range(3):
        one_pass(enc, x, dy, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one_pass(enc, x, dy, mask)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    host, gpu = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        one_pass(enc, x, dy, mask)
        b.record()
        host.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        gpu.append(a.elapsed_time(b))
    host.sort()
    gpu.sort()
    return wall, host[len(host) // 2], gpu[len(gpu) // 2]


def profile(enc, x, dy, mask, n=30):
    """cProfile of n passes on the stack-call path: where the host time goes (the C calls show as built-in calls)."""
    import cProfile
    import pstats
    ops.LAYER_C, ops.GEMM_SMALL_PER_OP, ops.LAYER_C_STACK = True, False, True
    for _ in range(3):
        one_pass(enc, x, dy, mask)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        one_pass(enc, x, dy, mask)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


def main():
    dev = torch.device("cuda:0")
    if "--profile" in sys.argv:
        sys.argv.remove("--profile")
        enc = build(dev)
        x = torch.randn(30, 50, 512, device=dev).bfloat16()
        dy = (torch.randn(30, 50, 512, device=dev) * 0.1).bfloat16()
        return profile(enc, x, dy, encoder.TriangularCausalMask(50, device=dev))
    args = [int(a) for a in sys.argv[1:]]
    shapes = list(zip(args[0::2], args[1::2])) or [(30, 50), (1, 50), (4, 1024)]
    enc = build(dev)
    print("%-12s %-8s %9s %9s %9s" % ("N x L", "path", "wall ms", "host ms", "gpu ms"))
    for N, L in shapes:
        x = torch.randn(N, L, 512, device=dev).bfloat16()
        dy = (torch.randn(N, L, 512, device=dev) * 0.1).bfloat16()
        mask = encoder.TriangularCausalMask(L, device=dev)
        only = os.environ.get("CWLT_BENCH_MODES")               # e.g. "layer-call" under rocprofv3
        for name, c, small, stack in (("stack-call", True, False, True), ("layer-call", True, False, False),
                                      ("per-op", False, False, False), ("per-op/small", False, True, False)):
            if only and name not in only.split(","):
                continue
            ops.LAYER_C, ops.GEMM_SMALL_PER_OP, ops.LAYER_C_STACK = c, small, stack
            w, h, g = measure(enc, x, dy, mask)
            print("%-12s %-12s %9.2f %9.2f %9.2f" % ("%d x %d" % (N, L), name, w, h, g), flush=True)


if __name__ == "__main__":
    main()
