"""Per-shape table of the projection GEMMs: cwlt_gemm_bf16 (csrc/gemm_bf16.hip) against hipBLASLt (torch.mm / addmm /
addmm_) on the same box, same random operands, interleaved rounds in one process (GPU box only).

usage: python tools/bench_gemm.py [M] [--variants 0,1] [--rounds 5]
       python tools/bench_gemm.py [M] --trace N K [--variants 0,1]     in-kernel s_memtime stamps of one tile (diagnostic build)
       python tools/bench_gemm.py [M] --ablate N K                     main-loop ablations, start staggers, small grids
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


def trace(M, N, K, variants):
    """Workgroup 0's second tile, stamped with s_memtime (shader cycles) by the diagnostic build: tile start, operands
    landed, then every barrier of the main loop (start of each compute segment, start of each load segment), main loop
    done, next tile's prologue issued, stores issued.  Printed per wave group (wave 0: leading half, wave 4: the half
    that runs one barrier behind)."""
    import numpy as np
    lib = _lib.load()
    dev = torch.device("cuda:0")
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    nK = K // 64
    for v in variants:
        buf = torch.zeros(8 * 1024, dtype=torch.int32, device=dev)
        lib.cwlt_gemm_bf16_tune(v, _lib.dev(buf))
        for _ in range(3):
            ops.gemm_bf16(a, w)
        torch.cuda.synchronize()
        st = buf.cpu().numpy().astype(np.int64).reshape(8, 1024) & 0xffffffff
        lib.cwlt_gemm_bf16_tune(-1, None)
        n = 2 + 8 * nK + 3
        print("variant %d  N=%d K=%d (%d K-tiles): cycles (s_memtime)" % (v, N, K, nK))
        for wv in (0, 4):
            s_ = st[wv, :n]
            d = np.diff(s_) % (1 << 32)
            loop = d[1:1 + 8 * nK].reshape(nK, 8)
            # stamp order inside a phase: [start of C] ... [end of C = start of next L]; d[1] is the first load segment
            load_seg, comp_seg = loop[:, 0::2], loop[:, 1::2]
            print("  wave %d: operands landed after %d | main loop %d = %.0f per K-tile | load segments (incl. barrier wait) "
                  "median %s | compute segments median %s | prologue issue %d | epilogue %d | tile %d"
                  % (wv, d[0], loop.sum(), loop.sum() / nK, np.median(load_seg, 0).astype(int).tolist(),
                     np.median(comp_seg, 0).astype(int).tolist(), d[2 + 8 * nK], d[3 + 8 * nK], s_[-1] - s_[0]))
            print("          first K-tile %s   last K-tile %s" % (loop[0].tolist(), loop[-1].tolist()))


def ablate(M, N, K, rounds):
    """Where the tile time goes (results of the ablated kernels are wrong by construction): main loop with parts switched
    off, grids smaller than the chip (is the tile boundary bandwidth- or latency-bound?), start staggers."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    names = {0: "everything", 1: "no DMA in the loop", 2: "no fragment reads", 3: "no DMA, no reads (MFMA + barriers)",
             4: "no barriers", 5: "no DMA, no barriers", 6: "no reads, no barriers", 7: "MFMA only"}

    def run(v):
        def f():
            lib.cwlt_gemm_bf16_tune(v, None)
            ops.gemm_bf16(a, w, out=c)
        return f
    fl = 2.0 * M * N * K
    print("N=%d K=%d M=%d: main-loop ablations (us, median of %d rounds)" % (N, K, M, rounds))
    ts = time_rounds([run(ab << 4) for ab in range(8)], rounds)
    for ab in range(8):
        t = sorted(ts[ab])[len(ts[ab]) // 2]
        print("  %-36s %8.1f us (%5.0f TF-equivalent)" % (names[ab], t * 1e3, fl / t / 1e9))
    ts = time_rounds([run(0), run(8 << 4), run(1)], rounds)
    for nm, t_ in zip(("everything (again)", "no counted waits in the loop",
                       "next tile's operands requested after the main loop"), ts):
        t = sorted(t_)[len(t_) // 2]
        print("  %-36s %8.1f us (%5.0f TF-equivalent)" % (nm, t * 1e3, fl / t / 1e9))
    print("start stagger (eighths of a tile period over 16 groups of workgroups):")
    ts = time_rounds([run(sg << 1) for sg in range(8)], rounds)
    for sg in range(8):
        t = sorted(ts[sg])[len(ts[sg]) // 2]
        print("  stagger %d/8  %8.1f us (%5.0f TF)" % (sg, t * 1e3, fl / t / 1e9))
    print("grid limited to g workgroups (time x g / 256 = time the whole chip would need at that per-CU rate):")
    for g8 in (4, 8, 16, 32):
        ts = time_rounds([run(g8 << 8)], max(2, rounds // 2), n=3)
        t = sorted(ts[0])[len(ts[0]) // 2]
        print("  %3d workgroups  %9.1f us  -> x g/256 = %8.1f us" % (g8 * 8, t * 1e3, t * 1e3 * g8 * 8 / 256))
    lib.cwlt_gemm_bf16_tune(-1, None)


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
    if "--ablate" in sys.argv:
        i = sys.argv.index("--ablate")
        return ablate(M, int(sys.argv[i + 1]), int(sys.argv[i + 2]), rounds)
    if "--trace" in sys.argv:
        i = sys.argv.index("--trace")
        return trace(M, int(sys.argv[i + 1]), int(sys.argv[i + 2]), variants)
    gemm_tuning.enable()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    print("M = %d rows; variant bit 0: next tile's operands requested after the main loop (default: from inside the last K-tile); bits 1-3: start stagger in eighths of a tile (variant -1 = the shipped default)" % M)
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
            lib.cwlt_gemm_bf16_tune(v, None)
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
        # hipBLASLt gets BOTH operand forms for the input-gradient shapes -- the weight as stored (NN) and a transposed
        # copy of it (NT, what encoder.py used from 16 384 rows up) -- and is quoted at the faster of the two
        lt2 = None
        if form == "addmm":
            lt = lambda: torch.addmm(bb, a, w.t(), out=cw)
        elif form == "mm":
            lt = lambda: torch.mm(a, w.t(), out=cw)
        elif form == "mm_nn":
            lt = lambda: torch.mm(a, wt, out=cw)
            lt2 = lambda: torch.mm(a, w.t(), out=cw)
        else:
            lt = lambda: cacc.addmm_(a, wt)
            lt2 = lambda: cacc.addmm_(a, w.t())
        fns = [lt] + ([lt2] if lt2 is not None else [])
        nlib = len(fns)
        for v in variants:
            def run(v=v):
                lib.cwlt_gemm_bf16_tune(v, None)
                ops.gemm_bf16(a, w, bias, out=cacc2 if acc else cw, accumulate=acc)
            fns.append(run)
        ts = time_rounds(fns, rounds)
        fl = 2.0 * M * N * K

        def fmt(t):
            t = sorted(t)
            med, mn = t[len(t) // 2], t[0]
            return "%7.1f us med %7.1f min (%4.0f TF)" % (med * 1e3, mn * 1e3, fl / med / 1e9)
        best = min(ts[:nlib], key=lambda t: sorted(t)[len(t) // 2])
        line = "%-38s err %.1e | hipBLASLt%s %s" % (name, err, " (NT on a transposed copy)" if nlib == 2 and best is ts[1]
                                                    else " (NN)" if nlib == 2 else "", fmt(best))
        for i, v in enumerate(variants):
            line += " | v%d %s" % (v, fmt(ts[nlib + i]))
        print(line, flush=True)
        del a, w, c0, cw, cacc, cacc2
    lib.cwlt_gemm_bf16_tune(-1, None)


if __name__ == "__main__":
    main()
