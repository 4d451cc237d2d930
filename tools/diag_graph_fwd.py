"""Diagnostic: does a graphed no-grad forward stay equal to its eager twin when large eager GEMMs run between
replays?  usage: python tools/diag_graph_fwd.py {bf16|f32}"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CWLT_COMPUTE_DTYPE"] = sys.argv[1] if len(sys.argv) > 1 else "bf16"
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import ops
from rlmg_amd.dqn_policy import IRL_dqn_train as T


def main():
    n_class = [56, 135, 18, 87, 18, 25]
    with contextlib.redirect_stdout(io.StringIO()):
        agent = T.DQN(n_class, Pretrain=False)
    agent.eval_net.eval()
    g = torch.Generator().manual_seed(0)
    tok = lambda *s: torch.stack([torch.randint(0, c, s, generator=g) for c in n_class], -1).cuda()  # noqa: E731
    big_a = torch.randn(100000, 512, device="cuda").bfloat16()
    big_w = torch.randn(1536, 512, device="cuda").bfloat16()
    fused = ops.GraphedCall(lambda x: agent._fused(agent.eval_net, x))
    bad = 0
    for i in range(40):
        x = tok(1, 50)
        for _ in range(10):
            torch.mm(big_a, big_w.t())
        with torch.no_grad():
            ref = agent._fused(agent.eval_net, x).float()
        for _ in range(10):
            torch.mm(big_a, big_w.t())
        got = fused(x).float()
        err = (got - ref).abs().max().item()
        fin = bool(torch.isfinite(got).all())
        if err > 1e-3 or not fin:
            bad += 1
        print("iter %d max|graph - eager| = %.3e finite=%s" % (i, err, fin), flush=True)
    print("bad iterations: %d / 40" % bad, flush=True)


if __name__ == "__main__":
    main()
