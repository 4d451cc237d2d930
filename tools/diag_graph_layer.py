"""Diagnostic for the training-graph corruption: ONE encoder layer, forward + backward captured in a hipGraph, replayed
with large eager GEMMs in between; every output is compared with the eager result.  usage: python tools/diag_graph_layer.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import encoder as enc, ops

torch.manual_seed(0)
dev = torch.device("cuda:0")
layer = enc.TransformerEncoderLayer(enc.AttentionLayer(512, 8, 64, 64), 512, 2048, 0.0).to(dev)
layer.train()
params = list(layer.parameters())
names = [n for n, _ in layer.named_parameters()]
ops.direct_grads = lambda p=None: False           # gradients come back through autograd
x = torch.randn(30, 50, 512, device=dev).bfloat16()
dout = torch.randn(30, 50, 512, device=dev).bfloat16()
for p in params:
    p.grad = torch.zeros_like(p)


def fn(xin, dy):
    for p in params:
        p.grad.zero_()
    xi = xin.detach().requires_grad_(True)
    y = layer(xi)
    y.backward(dy)
    return (y.detach(), xi.grad) + tuple(p.grad for p in params)


ref = [t.clone() for t in fn(x, dout)]
call = ops.GraphedCall(fn, grad=True, eager_calls=0)
big_a = torch.randn(100000, 512, device=dev).bfloat16()
big_w = torch.randn(1536, 512, device=dev).bfloat16()
labels = ["y", "dx"] + names
for it in range(12):
    if os.environ.get("NO_INTERLEAVE") != "1":
        for _ in range(10):
            torch.mm(big_a, big_w.t())
    out = call(x, dout)
    torch.cuda.synchronize()
    bad = []
    for lab, o, r in zip(labels, out, ref):
        d = (o.float() - r.float()).abs().max().item()
        if not (d <= 1e-3 * max(1.0, r.float().abs().max().item())):
            bad.append("%s %.3g" % (lab, d))
    print("replay %d: %s" % (it, "all outputs match" if not bad else "; ".join(bad)), flush=True)
