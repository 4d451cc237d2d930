"""Per-shape table of the projection GEMMs: cwlt_gemm_bf16 (csrc/gemm_bf16.hip) against hipBLASLt (torch.mm / addmm /
addmm_) on the same box, same random operands, interleaved rounds in one process (GPU box only).

usage: python tools/bench_gemm.py [M] [--variants 0,1,2,3] [--rounds 5]
Prints, per shape of the encoder layer at M token rows: correctness against an f64 product of the same bf16 operands,
then median / min microseconds and TFLOP/s of both.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import _lib, gemm_tuning, ops

# (name, N, K, bias, accumulate, hipBLASLt form)
SHAPES = [
    ("qkv fwd        (512 -> 1536) + bias", 1536, 512, True, False, "addmm"),
    ("linear2 fwd    (2048 -> 512) + bias", 512, 2048, True, False, "addmm"),
    ("out-proj fwd   (512 -> 512) + bias", 512, 512, True, False, "addmm"),
    ("linear1 fwd    (512 -> 2048)", 2048, 512, False, False, "mm"),
    ("linear1 dgrad  (2048 -> 512) C +=", 512, 2048, False, True, "addmm_"),
    ("out-proj dgrad (512 -> 512)", 512, 512, False, False, "mm_nn"),
    ("qkv dgrad      (1536 -> 512) C +=", 512, 1536, False, True, "addmm_"),
    ("heads fwd      (512 -> 384) + bias", 384, 512, True, False, "addmm"),
]


def time_rounds(fns, rounds, n=10, warm=2):
    """fns: list of callables; returns per-callable list of per-round mean milliseconds (interleaved)."""
    for f in fns:
        for _ in range(warm):
            f()
    torch.cuda.synchronize()
    out = [[] for _ in fns]
    for _ in range(rounds):
        for i, f in enumerate(fns):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                f()
            b.record()
            torch.cuda.synchronize()
            out[i].append(a.elapsed_time(b) / n)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    M = int(args[0]) if args else 524288
    variants = [0]
    rounds = 5
    for i, a in enumerate(sys.argv):
        if a == "--variants":
            variants = [int(v) for v in sys.argv[i + 1].split(",")]
        if a == "--rounds":
            rounds = int(sys.argv[i + 1])
    gemm_tuning.enable()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print("M = %d rows; variants: bit 0 = DMA 6 half-tiles ahead (default 5), bit 1 = no pre-read of the next K-tile" % M)
    for name, N, K, has_bias, acc, form in SHAPES:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev) * 0.1 if has_bias else None
        c0 = torch.randn(M, N, device=dev).bfloat16() if acc else None
        # correctness on a slab of rows (f64 product of the same bf16 operands), every variant
        rows = min(M, 4096)
        sl = slice(M - rows, M)
        ref = a[sl].double() @ w.double().t()
        if has_bias:
            ref = ref + bias.double()
        if acc:
            ref = ref + c0[sl].double()
        for v in variants:
            lib.cwlt_gemm_bf16_tune(v)
            out = c0.clone() if acc else None
            out = ops.gemm_bf16(a, w, bias, out=out, accumulate=acc)
            err = (out[sl].double() - ref).abs().max().item() / ref.abs().max().item()
            assert err < 1e-2, (name, v, err)
        # timing
        bb = bias.bfloat16() if has_bias else None
        wt = w.t().contiguous()          # (K, N): the weight as stored for the input-gradient forms
        cw = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        cacc = c0.clone() if acc else None
        cacc2 = c0.clone() if acc else None
        if form == "addmm":
            lt = lambda: torch.addmm(bb, a, w.t(), out=cw)
        elif form == "mm":
            lt = lambda: torch.mm(a, w.t(), out=cw)
        elif form == "mm_nn":
            lt = lambda: torch.mm(a, wt, out=cw)
        else:
            lt = lambda: cacc.addmm_(a, wt)
        fns = [lt]
        for v in variants:
            def run(v=v):
                lib.cwlt_gemm_bf16_tune(v)
                ops.gemm_bf16(a, w, bias, out=cacc2 if acc else cw, accumulate=acc)
            fns.append(run)
        ts = time_rounds(fns, rounds)
        fl = 2.0 * M * N * K

        def fmt(t):
            t = sorted(t)
            med, mn = t[len(t) // 2], t[0]
            return "%7.1f us med %7.1f min (%4.0f TF)" % (med * 1e3, mn * 1e3, fl / med / 1e9)
        line = "%-38s err %.1e | hipBLASLt %s" % (name, err, fmt(ts[0]))
        for i, v in enumerate(variants):
            line += " | v%d %s" % (v, fmt(ts[1 + i]))
        print(line, flush=True)
        del a, w, c0, cw, cacc, cacc2
    lib.cwlt_gemm_bf16_tune(-1)


if __name__ == "__main__":
    main()
