"""The bf16 attention kernels alone (one-sweep backward, forward scan) at three shapes, five rounds of ten launches each
(GPU box): python tools/bench_sweep.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import rlmg_amd  # noqa
from rlmg_amd import ops
dev = torch.device("cuda:0")
for B, T in ((512, 1024), (128, 4096), (512, 1000)):
    qkv = torch.randn(B, T, 3, 8, 64, device=dev).bfloat16()
    dout = torch.randn(B, T, 8, 64, device=dev).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    _, _, _, out, zinv, fin = ops.cla_fwd(q, k, v, final_state=True)
    def f(): ops.cla_bwd(q, k, v, out, zinv, dout, want_colsum=True, final_state=fin)
    def g(): ops.cla_fwd(q, k, v, final_state=True)
    for name, fn in (("sweep", f), ("fwd", g)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): fn()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 10)
        print("B=%d T=%d %-6s %s ms" % (B, T, name, " ".join("%.4f" % t for t in ts)), flush=True)
