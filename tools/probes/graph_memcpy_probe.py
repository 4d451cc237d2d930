"""Does a device-to-device copy captured into a hipGraph still copy the right bytes when the graph is replayed with other
work in between?

Round 1's captured DQN.update (NaN losses from the third replay on, once an illegal address, only with large eager GEMMs
between replays) held 1788 kernel nodes and 6 MEMCPY nodes and no memset node (profiles/r04_graph_census_r1_tree.txt), so
the memset-replay defect of ROCm 7.2 (tools/probes/graph_memset_probe.py) cannot be what broke it.  The copy nodes are the
other kind of node the runtime executes itself (a blit, not a user kernel): this probe checks that one mechanism.

  graph A: dst.copy_(src), contiguous, same dtype     (torch issues hipMemcpyAsync device-to-device: a memcpy node)
  graph B: hipMemcpyAsync through the runtime          (explicit memcpy node)
  graph C: dst.copy_(src) with a dtype change          (always a kernel)
  graph D: a small copy (one scalar), as a loss.detach().clone() / an lr tensor would be

each replayed 12 times with `src` rewritten (replay index) and `dst` poisoned (NaN) before every replay and, between
replays, large eager GEMMs plus allocator churn.  Prints, per graph, on which replays dst != src, and the node kinds of each
captured graph.  GPU box only; runs once.
"""
import ctypes
import sys

import torch


def census(graph, hip):
    kinds = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty"}
    raw = int(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n))
    nodes = (ctypes.c_void_p * max(1, n.value))()
    hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n))
    out = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        k = kinds.get(t.value, "type%d" % t.value)
        out[k] = out.get(k, 0) + 1
    return out


def main():
    dev = torch.device("cuda:0")
    hip = ctypes.CDLL(torch.__path__[0] + "/lib/libamdhip64.so")
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    bad_any = False
    for name, n, mode in (("dst.copy_(src) 32 MiB", 8 << 20, "torch"), ("hipMemcpyAsync 32 MiB", 8 << 20, "hip"),
                          ("copy_ with dtype change", 8 << 20, "cast"), ("dst.copy_(src) 1 element", 1, "torch"),
                          ("hipMemcpyAsync 4 bytes", 1, "hip")):
        src = torch.zeros(n, device=dev)
        dst = torch.empty(n, device=dev, dtype=torch.float64 if mode == "cast" else torch.float32)

        def fn():
            if mode == "hip":
                st = torch.cuda.current_stream().cuda_stream
                rc = hip.hipMemcpyAsync(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()),
                                        ctypes.c_size_t(n * 4), 3, ctypes.c_void_p(st))      # 3 = hipMemcpyDeviceToDevice
                assert rc == 0, rc
            else:
                dst.copy_(src)

        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g):
            fn()
        kinds = census(g, hip)
        bad = []
        for it in range(12):
            src.fill_(float(it + 1))
            dst.fill_(float("nan"))
            for _ in range(3):
                c = torch.mm(a, b)
            junk = [torch.empty(1 << 22, device=dev).normal_() for _ in range(4)]
            del junk, c
            g.replay()
            torch.cuda.synchronize()
            wrong = int((dst.double() != float(it + 1)).sum().item())
            if wrong:
                bad.append((it, wrong))
        print("%-28s nodes %-28s %s" % (name, kinds, "correct on 12 replays" if not bad else "WRONG on replays %s" % bad),
              flush=True)
        bad_any = bad_any or bool(bad)
    print("verdict:", "a captured device-to-device copy replayed wrongly" if bad_any
          else "captured device-to-device copies replay correctly here")
    return 0


if __name__ == "__main__":
    sys.exit(main())
