"""What does a captured `DQN.update` bake in?  Captures the update step (repo dims, bf16, batch 30 x window 50) with
CWLT_TRAIN_GRAPHS=1 and dumps the hipGraph's nodes (CUDAGraph.debug_dump): counts kernel / memset / memcpy nodes and
lists the memset nodes -- a captured hipMemsetAsync replays with a wrong fill pattern on ROCm 7.2
(tools/probes/graph_memset_probe.py), so every memset node is a place where a replay can read garbage.
usage: python tools/diag_graph_nodes.py [tuned|default] [bf16|f32]       (GPU box; captures once, replays nothing)"""
import collections
import glob
import os
import re
import sys

os.environ["CWLT_TRAIN_GRAPHS"] = "1"
os.environ.setdefault("CWLT_GRAPH_DEBUG_DUMP", os.path.join("gpurun_out", "graph_dump"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import rlmg_amd  # noqa: E402,F401
from rlmg_amd import gemm_tuning  # noqa: E402


def main():
    tuned = (sys.argv[1] if len(sys.argv) > 1 else "tuned") == "tuned"
    dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    os.environ["CWLT_COMPUTE_DTYPE"] = dtype
    os.environ["CWLT_NO_PRETRAIN"] = "1"
    print("tuned table:", gemm_tuning.enable() if tuned else False, "| dtype", dtype, flush=True)
    from rlmg_amd.dqn_policy import IRL_dqn_train as T
    n_class = [56, 135, 18, 87, 18, 25]
    torch.manual_seed(0)
    agent = T.DQN(n_class, Pretrain=False)
    if dtype == "bf16":
        agent.eval_net.compute_dtype = agent.target_net.compute_dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    B = 30
    st = torch.stack([torch.randint(0, n, (B, 50), generator=g) for n in n_class], -1).cuda()
    ns = torch.stack([torch.randint(0, n, (B, 50), generator=g) for n in n_class], -1).cuda()
    ac = torch.stack([torch.randint(0, n, (B, 25), generator=g) for n in n_class], -1).cuda()
    tr = {"state": st, "action": ac, "reward": torch.rand(B, 1), "nextstate": ns, "done": torch.zeros(B, 1)}
    n_replay = int(os.environ.get("DIAG_REPLAYS", "0"))
    for i in range(3 + n_replay):                        # 2 eager calls, the third is captured, then replays
        m, c, t = agent.update(tr, dict(tr), torch.ones(B, 50).cuda(), False, 0)
    torch.cuda.synchronize()
    files = sorted(glob.glob(os.path.join(os.environ["CWLT_GRAPH_DEBUG_DUMP"], "*.dot")))
    print("losses of the captured step:", m, c, t, "| dumps:", files)
    for f in files:
        txt = open(f).read()
        kinds = collections.Counter(re.findall(r'label="[^"]*?(KERNEL|MEMSET|MEMCPY|Kernel|Memset|Memcpy|EMPTY|HOST)', txt))
        print(os.path.basename(f), dict(kinds), "nodes total", txt.count("label="))
        for line in txt.splitlines():
            if re.search(r"MEMSET|Memset|memset", line):
                print("   ", line.strip()[:300])


if __name__ == "__main__":
    main()
