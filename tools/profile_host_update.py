"""Host-side profile (cProfile) of the launch-bound DQN.update at repo dims, bf16 (GPU box)."""
import contextlib
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CWLT_COMPUTE_DTYPE", "bf16")
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import gemm_tuning
from rlmg_amd.dqn_policy import IRL_dqn_train as T

if os.environ.get("NO_TUNED") != "1":
    gemm_tuning.enable()
if os.environ.get("BLAS"):                      # "cublas" = rocBLAS, "cublaslt" = hipBLASLt (the default on this build)
    torch.backends.cuda.preferred_blas_library(os.environ["BLAS"])
    print("preferred BLAS library:", torch.backends.cuda.preferred_blas_library())
n_class = [56, 135, 18, 87, 18, 25]
with contextlib.redirect_stdout(io.StringIO()):
    agent = T.DQN(n_class, Pretrain=False)
g = torch.Generator().manual_seed(0)
tok = lambda *s: torch.stack([torch.randint(0, c, s, generator=g) for c in n_class], -1).cuda()  # noqa: E731
B = 30
tr = {"state": tok(B, 50), "action": tok(B, 25), "reward": torch.rand(B, 1), "nextstate": tok(B, 50), "done": torch.zeros(B, 1)}
m = torch.ones(B, 50).cuda()


def run(n):
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(n):
            agent.update(tr, dict(tr), m, False, 0)
    torch.cuda.synchronize()


run(3)
pr = cProfile.Profile()
pr.enable()
run(20)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(12)
print("ms per update: %.2f" % (1e3 * st.total_tt / 20))
