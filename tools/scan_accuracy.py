"""bf16 scan kernels vs the f64 oracle on bf16-rounded inputs: max abs error relative to max |ref| (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import ops
from oracle import cla as ocla

for (N, L, H) in ((2, 1024, 2), (1, 4096, 1)):
    g0 = torch.Generator().manual_seed(11)
    q, k, v, g = (torch.randn(N, L, H, 64, generator=g0).bfloat16() for _ in range(4))
    ref = ocla.cla_grads(q.double(), k.double(), v.double(), g.double())
    qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = ops.causal_linear_attention(qd, kd, vd)
    out.backward(g.cuda())
    for name, got, r in zip(("out", "dq", "dk", "dv"), (out, qd.grad, kd.grad, vd.grad), ref):
        e = (got.detach().cpu().double() - r).abs()
        # error of rounding the exact result to bf16 once, for scale
        e0 = (r.float().bfloat16().double() - r).abs()
        print("N%d L%d %-3s max err %.3e (rms %.3e) | one bf16 rounding of the exact result: max %.3e (rms %.3e) | max|ref| %.3f"
              % (N, L, name, e.max(), e.pow(2).mean().sqrt(), e0.max(), e0.pow(2).mean().sqrt(), r.abs().max()))
