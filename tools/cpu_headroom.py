"""How long the host needs to ENQUEUE one pretrain step vs how long the GPU needs to run it (GPU box).
usage: python tools/cpu_headroom.py [B]"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench  # noqa: E402
import rlmg_amd  # noqa: F401
from rlmg_amd import dist as rdist, gemm_tuning
from rlmg_amd.dqn_policy import model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
gemm_tuning.enable()
with contextlib.redirect_stdout(io.StringIO()):
    net = model.LinearTransformer([56, 135, 18, 87, 18, 25]).to(dev).train()
net.compute_dtype = torch.bfloat16
sync = rdist.GradSync(net.parameters())
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
x, y, mask = bench.synth_batch(B, 1024, 1234, dev)


def step():
    sync.zero_grad()
    losses = net.train_step(x, y, mask)
    loss = (losses[0] + losses[1] + losses[2] + losses[3] + losses[4] + losses[5]) / 6
    loss.backward()
    sync.finish()
    sync.clip_grad_norm_(3.0)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
# GPU queue is empty: enqueue 1 step and see when the host returns (it runs ahead of the device)
enq = []
t0 = time.perf_counter()
for _ in range(6):
    a = time.perf_counter()
    step()
    enq.append(time.perf_counter() - a)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("host enqueue time per step (ms):", ["%.1f" % (1e3 * t) for t in enq])
print("device time per step (ms): %.1f" % (1e3 * tot / 6))
