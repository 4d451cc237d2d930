"""Does a zero-fill captured into a hipGraph still write zeros when the graph is replayed with other work in between?

HISTORY 4.4 recorded (decode experiments, round 1): "a hipMemsetAsync captured into a hipGraph replayed with a garbage
fill pattern (ROCm 7.2)".  A captured training step contains exactly such fills -- `GradSync.zero_grad()`
(`flat.zero_()` on 32 MiB buckets) and `torch.zeros` of the TD-loss gradient -- and round 1's unexplained fault (NaN
losses from the third replay on, once an illegal address, only with LARGE EAGER GEMMs between replays) is what a fill
node with a stale pattern / argument buffer would look like.  This probe isolates that one mechanism:

  graph A: buf.zero_()                         (torch picks hipMemsetAsync or a fill kernel, whichever it does)
  graph B: torch.zeros inside the capture      (allocation from the graph's pool + fill)
  graph C: buf.fill_(0.0)                      (always a kernel)
  graph D: hipMemsetAsync through the runtime  (explicit memset node)

each replayed 12 times with `buf` poisoned (NaN) before every replay and, between replays, large eager GEMMs plus
allocator churn.  Prints, per graph, on which replays the buffer was not all zero.  GPU box only; runs once.
"""
import ctypes
import sys

import torch


def main():
    dev = torch.device("cuda:0")
    n = 8 << 20                                           # 32 MiB of f32, the size of a gradient bucket
    buf = torch.empty(n, device=dev)
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    hip = ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL) if False else None
    try:
        hip = ctypes.CDLL(torch.__path__[0] + "/lib/libamdhip64.so")
    except OSError:
        hip = None

    def cap(fn):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        return g, out

    def f_zero():
        buf.zero_()
        return buf

    def f_zeros():
        z = torch.zeros(n, device=dev)
        return z

    def f_fill():
        buf.fill_(0.0)
        return buf

    def f_memset():
        st = torch.cuda.current_stream().cuda_stream
        rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(n * 4), ctypes.c_void_p(st))
        assert rc == 0, rc
        return buf

    cases = [("buf.zero_()", f_zero), ("torch.zeros in graph", f_zeros), ("buf.fill_(0.0)", f_fill)]
    if hip is not None:
        cases.append(("hipMemsetAsync", f_memset))
    bad_any = False
    for name, fn in cases:
        g, out = cap(fn)
        bad = []
        for it in range(12):
            out.fill_(float("nan"))
            # eager work between replays: large GEMMs (hipBLASLt workspaces, big kernarg traffic) and allocator churn
            for _ in range(3):
                c = torch.mm(a, b)
            junk = [torch.empty(1 << 22, device=dev).normal_() for _ in range(4)]
            del junk, c
            g.replay()
            torch.cuda.synchronize()
            nz = int((out != 0).sum().item())
            if nz:
                bad.append((it, nz, float(out.float().abs().nan_to_num(1e30).max().item())))
        print("%-24s %s" % (name, "all zero on 12 replays" if not bad else "NOT ZERO on replays %s" % bad), flush=True)
        bad_any = bad_any or bool(bad)
    print("verdict:", "a captured fill replayed with a wrong pattern" if bad_any else "captured fills replay correctly here")
    return 0


if __name__ == "__main__":
    sys.exit(main())
