"""What does a captured `DQN.update` bake in?  Captures the update step (repo dims, bf16, batch 30 x window 50) with
CWLT_TRAIN_GRAPHS=1 and prints the node census of the hipGraph (`GraphedCall.census`, from hipGraphGetNodes): kernel /
memset / memcpy nodes -- a captured hipMemsetAsync replays with a wrong fill pattern on ROCm 7.2
(tools/probes/graph_memset_probe.py), so every memset node is a place where a replay can read garbage (GraphedCall
refuses to replay such a capture).  DIAG_REPLAYS=n: n further updates (replays), for a kernel trace under rocprofv3.
usage: python tools/diag_graph_nodes.py [tuned|default] [bf16|f32]       (GPU box)"""
import os
import sys

os.environ["CWLT_TRAIN_GRAPHS"] = "1"
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
    print("losses of the last step:", m, c, t)
    print("update graph census:", agent._graph_update.census)
    gc = getattr(agent, "_graph_choose", None)
    if gc is not None:
        print("choose_action graph census:", gc.census)


if __name__ == "__main__":
    main()
