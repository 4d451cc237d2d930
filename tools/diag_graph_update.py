"""Diagnostic: graphed DQN.update at repo dims (GPU box).  usage: python tools/diag_graph_update.py {tuned|untuned} {bf16|f32}"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CWLT_TRAIN_GRAPHS"] = "1"
os.environ["CWLT_COMPUTE_DTYPE"] = sys.argv[2] if len(sys.argv) > 2 else "bf16"
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import gemm_tuning, ops
from rlmg_amd.dqn_policy import IRL_dqn_train as T


def say(msg):
    print(msg, flush=True)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "untuned"
    if mode == "tuned":
        say("tuned table: %s" % gemm_tuning.enable())
    n_class = [56, 135, 18, 87, 18, 25]
    with contextlib.redirect_stdout(io.StringIO()):
        agent = T.DQN(n_class, Pretrain=False)
    g = torch.Generator().manual_seed(0)
    tok = lambda *s: torch.stack([torch.randint(0, c, s, generator=g) for c in n_class], -1).cuda()  # noqa: E731
    B = 30
    m = torch.ones(B, 50).cuda()
    n_upd = int(os.environ.get("DIAG_UPDATES", "8"))
    if os.environ.get("DIAG_NOGC") == "1":
        import gc
        gc.disable()
    for i in range(n_upd):
        tr = {"state": tok(B, 50), "action": tok(B, 25), "reward": torch.rand(B, 1), "nextstate": tok(B, 50),
              "done": torch.zeros(B, 1)}
        with contextlib.redirect_stdout(io.StringIO()):
            out = agent.update(tr, dict(tr), m, False, 0)
        torch.cuda.synchronize()
        say("update %d ok %s graphs=%d" % (i, ["%.4f" % v for v in out], len(getattr(agent, "_graph_update").graphs)
                                          if getattr(agent, "_graph_update", None) else 0))
        x = tok(1, 50)
        agent.choose_action(x)
        torch.cuda.synchronize()
        say("  choose_action ok")
    say("done")


if __name__ == "__main__":
    main()
