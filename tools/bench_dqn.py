"""Throughput of the drop-in DQN + AIRL loop (dqn_policy/IRL_dqn_train.py) at repo dims on synthetic data.
usage: python tools/bench_dqn.py [buffer_size] [update_songs]      (GPU box)"""
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


def main():
    buf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    upd_songs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    fill_songs = buf // T.EPISODES + 1
    T.BUFFER_SIZE = buf
    T.NUM_SONGS = fill_songs + upd_songs
    os.environ["CWLT_NO_PRETRAIN"] = "1"
    os.makedirs("gpurun_out/dqn_run", exist_ok=True)
    os.chdir("gpurun_out/dqn_run")
    gemm_tuning.enable()
    marks = []
    orig_update = T.DQN.update

    trace = os.environ.get("CWLT_TRACE") == "1"
    count = [0]

    def timed_update(self, *a, **k):
        torch.cuda.synchronize()
        if not marks:
            marks.append(time.perf_counter())
        if trace:
            print("update %d start" % count[0], file=sys.stderr, flush=True)
        out = orig_update(self, *a, **k)
        if trace:
            torch.cuda.synchronize()
            print("update %d ok" % count[0], file=sys.stderr, flush=True)
        count[0] += 1
        return out

    T.DQN.update = timed_update
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        T.main()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    fill_steps = fill_songs * T.EPISODES
    upd_steps = (T.NUM_SONGS * T.EPISODES) - buf - 1
    t_fill = marks[0] - t0
    t_upd = t1 - marks[0]
    print("DQN loop, repo dims, window 50, BUFFER_SIZE %d" % buf)
    print("  rollout-only phase : %d env-steps in %.1f s  -> %.1f env-steps/s (incl. model construction)" %
          (buf, t_fill, buf / t_fill))
    print("  update phase       : %d env-steps in %.1f s  -> %.2f env-steps/s "
          "(each: re-score 2 x %d windows through the 10-layer Longformer + DQN update of batch 30)" %
          (upd_steps, t_upd, upd_steps / t_upd, buf))


if __name__ == "__main__":
    main()
