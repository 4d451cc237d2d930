"""Where an IRL_dqn_train environment step spends its time once the buffer is full (GPU box).
usage: python tools/time_dqn_parts.py [buffer_size] [--tune]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

os.environ.setdefault("CWLT_COMPUTE_DTYPE", "bf16")   # throughput mode (BASELINE configs: bf16)

import rlmg_amd  # noqa: F401
from rlmg_amd import gemm_tuning
from rlmg_amd.dqn_policy import IRL_dqn_train as T
from rlmg_amd.dqn_policy.AIRL import RewardDiscri


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    buf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs("gpurun_out/dqn_parts", exist_ok=True)
    os.chdir("gpurun_out/dqn_parts")
    if "--tune" in sys.argv:       # extend the shipped table with this loop's GEMM shapes
        gemm_tuning.tune(os.path.join(root, "gpurun_out", "gemm_gfx950.csv"))
    else:
        gemm_tuning.enable()
    n_class = [56, 135, 18, 87, 18, 25]
    with contextlib.redirect_stdout(io.StringIO()):
        agent = T.DQN(n_class, Pretrain=False)
        rew = RewardDiscri(n_class, Pretrain=False)
    g = torch.Generator().manual_seed(0)
    tok = lambda *s: torch.stack([torch.randint(0, c, s, generator=g) for c in n_class], -1).cuda()  # noqa: E731
    states, nxt = tok(buf, 50), tok(buf, 50)
    mask = torch.ones(buf, 50).cuda()
    done = torch.zeros(buf, 1).cuda()
    print("calculate_reward(%d windows)      %8.2f ms" % (buf, timed(lambda: rew.calculate_reward(states, done, nxt, mask, mask))))
    ag = (states, None, None, nxt, done)
    ex = (states, None, None, nxt, done, mask, mask)
    print("update_disc(train=False), 2 buffers %8.2f ms" % timed(lambda: rew.update_disc(ag, ex, train=False)))
    B = 30
    tr = {"state": tok(B, 50), "action": tok(B, 25), "reward": torch.rand(B, 1), "nextstate": tok(B, 50),
          "done": torch.zeros(B, 1)}
    m = torch.ones(B, 50).cuda()
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(4):                                  # eager steps + the capture of the update graph
            agent.update(tr, dict(tr), m, False, 0)
        t = timed(lambda: agent.update(tr, dict(tr), m, False, 0))
    print("DQN.update (batch 30)              %8.2f ms" % t)
    x = tok(1, 50)
    print("choose_action                      %8.2f ms" % timed(lambda: agent.choose_action(x), 20))
    with torch.no_grad():
        big = rew.disc_model
        big.train()
        print("  Longformer body, %d windows    %8.2f ms" % (buf, timed(lambda: big._encode(states, mask))))


if __name__ == "__main__":
    main()
