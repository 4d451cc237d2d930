"""Which torch ops issue device-to-device copies in one pretrain step (GPU box).  usage: python tools/profile_copies.py [B]"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench  # noqa: E402
import rlmg_amd  # noqa: F401
from rlmg_amd import dist as rdist, gemm_tuning
from rlmg_amd.dqn_policy import model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
    loss = sum(losses) / 6
    loss.backward()
    sync.finish()
    sync.clip_grad_norm_(3.0)
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::add_", "aten::fill_",
                 "aten::zero_", "aten::cat"):
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
for r in sorted(rows, reverse=True)[:12]:
    print("%9.1f us  x%-4d %-18s %s" % r)
print("---- device memcpy / memset activities and their parents")
evs = [e for e in prof.events() if "emcpy" in e.name or "emset" in e.name]
agg = {}
for e in evs:
    par = e.cpu_parent.name if e.cpu_parent is not None else "?"
    gp = e.cpu_parent.cpu_parent.name if (e.cpu_parent is not None and e.cpu_parent.cpu_parent is not None) else "?"
    k = (e.name[:40], par[:30], gp[:40])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += e.device_time
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
    print("%9.1f us  x%-4d %s <- %s <- %s" % (t, n, k[0], k[1], k[2]))
