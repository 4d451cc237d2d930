"""One inspection of round 1's failing tree (GPU box, run ONCE): what kinds of node did its captured `DQN.update` hold?

Round 1 recorded NaN losses from the third replay of a captured DQN.update on (profiles/r01_diag_gemm.log) and once
hipErrorIllegalAddress (profiles/r01_dqn_trace_fault.log) whenever large eager GEMMs ran between replays; the commit that
made it go away (4d37611) does not explain it.  Round 3 found that a captured hipMemsetAsync replays a wrong fill pattern on
ROCm 7.2 once other work runs between replays (profiles/r03_graph_memset_probe.txt).  This script connects the two or
rules the connection out: it imports the tree of `4d37611^` (extracted to _r1tree/ with `git archive 4d37611^ | tar -x -C
_r1tree`, library built there with that tree's own build script), runs two eager updates at repo dims / bf16 / batch
30 x window 50 exactly as that tree's GraphedCall would (ops.GraphedCall._capture: seed base bumped inside the capture,
`_USE_SEED_BASE` on), captures the third with torch.cuda.CUDAGraph(keep_graph=True), and counts the graph's nodes by kind
with hipGraphGetNodes / hipGraphNodeGetType.  Nothing is replayed.

usage (from the repo root, after `git archive 4d37611^ | tar -x -C _r1tree` and building _r1tree's library):
    python tools/diag_r1_graph_census.py [tuned]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R1 = os.path.join(ROOT, "_r1tree")
sys.path.insert(0, R1)
os.environ.setdefault("CWLT_COMPUTE_DTYPE", "bf16")
os.environ["CWLT_TRAIN_GRAPHS"] = "1"
os.environ["CWLT_NO_PRETRAIN"] = "1"

import torch  # noqa: E402

import rlmg_amd  # noqa: E402,F401  -- the ROUND-1 package (sys.path[0] is _r1tree)
from rlmg_amd import gemm_tuning, ops  # noqa: E402
from rlmg_amd.dqn_policy import IRL_dqn_train as T  # noqa: E402

assert os.path.realpath(rlmg_amd.__file__).startswith(os.path.realpath(R1)), rlmg_amd.__file__

KINDS = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}


def census(raw, hip, counts=None, memsets=None):
    counts = {} if counts is None else counts
    memsets = [] if memsets is None else memsets
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) == 0
    if n.value == 0:
        return counts, memsets
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n)) == 0
    for node in nodes:
        t = ctypes.c_int(-1)
        assert hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t)) == 0
        k = KINDS.get(t.value, "type%d" % t.value)
        counts[k] = counts.get(k, 0) + 1
        if t.value == 2:
            # hipMemsetParams: dst, elementSize, height, pitch, value, width (hip_runtime_api.h)
            class P(ctypes.Structure):
                _fields_ = [("dst", ctypes.c_void_p), ("elementSize", ctypes.c_uint), ("height", ctypes.c_size_t),
                            ("pitch", ctypes.c_size_t), ("value", ctypes.c_uint), ("width", ctypes.c_size_t)]
            p = P()
            if hip.hipGraphMemsetNodeGetParams(ctypes.c_void_p(node), ctypes.byref(p)) == 0:
                memsets.append((p.elementSize, p.width, p.height, p.value))
        if t.value == 4:
            child = ctypes.c_void_p(0)
            if hip.hipGraphChildGraphNodeGetGraph(ctypes.c_void_p(node), ctypes.byref(child)) == 0 and child.value:
                census(child.value, hip, counts, memsets)
    return counts, memsets


def main():
    tuned = len(sys.argv) > 1 and sys.argv[1] == "tuned"
    if tuned:
        gemm_tuning.enable()
    dev = torch.device("cuda:0")
    n_class = [56, 135, 18, 87, 18, 25]
    torch.manual_seed(0)
    agent = T.DQN(n_class, Pretrain=False)
    g = torch.Generator().manual_seed(1)
    B = T.batch_size
    st = torch.stack([torch.randint(0, n, (B, 50), generator=g) for n in n_class], -1).to(dev)
    ns = torch.stack([torch.randint(0, n, (B, 50), generator=g) for n in n_class], -1).to(dev)
    ac = torch.stack([torch.randint(0, n, (B, 25), generator=g) for n in n_class], -1).to(dev)
    rw = torch.rand(B, 1, generator=g).to(dev)
    dn = torch.randint(0, 2, (B, 1), generator=g).to(dev)
    mask = (torch.rand(B, 50, generator=g) > 0.2).float().to(dev)
    args = (st, ns, ac, rw, dn, ns.clone(), mask)
    if agent.target_count % T.Target_update == 0:
        agent.target_net.load_state_dict(agent.eval_net.state_dict())
    for i in range(2):                                     # GraphedCall(grad=True): two eager calls first
        out = agent._update_device(*args)
        print("eager update %d: %s" % (i, [round(float(v), 4) for v in out]), flush=True)
    # the capture of that tree's GraphedCall._capture, with keep_graph=True so that the hipGraph_t can be inspected
    base = ops.seed_base_tensor(dev)
    static = [a.clone() for a in args]
    ops._USE_SEED_BASE = True
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    try:
        with torch.cuda.graph(graph), torch.enable_grad():
            base.add_(ops._SEED_STEP)
            agent._update_device(*static)
    finally:
        ops._USE_SEED_BASE = False
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    counts, memsets = census(int(graph.raw_cuda_graph()), hip)
    print("round-1 tree (%s), hipBLASLt tuning table %s: captured DQN.update node census: %s"
          % (os.popen("git -C %s rev-parse --short 4d37611^ 2>/dev/null" % ROOT).read().strip() or "4d37611^",
             "on" if tuned else "off", dict(sorted(counts.items()))), flush=True)
    by = {}
    for m in memsets:
        by[m] = by.get(m, 0) + 1
    for (es, w, h, v), c in sorted(by.items()):
        print("  memset node x%d: elementSize %d, width %d, height %d, value 0x%x" % (c, es, w, h, v))
    graph.reset()                                          # never instantiated, never replayed


if __name__ == "__main__":
    main()
