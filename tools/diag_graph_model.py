"""Diagnostic for the training-graph corruption, stage 2: the whole CW Linear Transformer (repo dims, bf16):
train_step forward + backward [+ capturable Adam] captured in a hipGraph, replayed with large eager GEMMs in between.
usage: python tools/diag_graph_model.py [grads|adam]"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CWLT_COMPUTE_DTYPE"] = "bf16"
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import dist as rdist, ops
from rlmg_amd.dqn_policy import model

mode = sys.argv[1] if len(sys.argv) > 1 else "grads"
torch.manual_seed(0)
dev = torch.device("cuda:0")
n_class = [56, 135, 18, 87, 18, 25]
with contextlib.redirect_stdout(io.StringIO()):
    net = model.LinearTransformer(n_class).to(dev)
net.eval()                                             # dropout off: replays must reproduce the eager numbers
sync = rdist.GradSync(net.parameters())
g = torch.Generator().manual_seed(0)
tok = lambda *s: torch.stack([torch.randint(0, c, s, generator=g) for c in n_class], -1).cuda()  # noqa: E731
x, y = tok(30, 50), tok(30, 50)
mask = torch.ones(30, 50, device=dev)
opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(1e-3, device=dev), capturable=True) if mode == "adam" else None
w0 = [p.detach().clone() for p in net.parameters()]


def fn(xi, yi, mi):
    sync.zero_grad()
    losses = net.train_step(xi, yi, mi)
    loss = (losses[0] + losses[1] + losses[2] + losses[3] + losses[4] + losses[5]) / 6
    loss.backward()
    sync.finish()
    if opt is not None:
        opt.step()
    return (loss.detach(),) + tuple(b.flat for b in sync.buckets)


def reset():
    with torch.no_grad():
        for p, w in zip(net.parameters(), w0):
            p.copy_(w)


ref = [t.clone() for t in fn(x, y, mask)]
if opt is not None:                                    # eager steps create the optimizer state
    fn(x, y, mask)
call = ops.GraphedCall(fn, grad=True, eager_calls=0)
big_a = torch.randn(100000, 512, device=dev).bfloat16()
big_w = torch.randn(1536, 512, device=dev).bfloat16()
for it in range(12):
    if os.environ.get("NO_INTERLEAVE") != "1":
        for _ in range(10):
            torch.mm(big_a, big_w.t())
    if opt is None:
        out = call(x, y, mask)
    else:
        reset()
        out = call(x, y, mask)
    torch.cuda.synchronize()
    msg = []
    for i, (o, r) in enumerate(zip(out, ref)):
        fin = bool(torch.isfinite(o).all())
        d = (o.float() - r.float()).abs().max().item()
        if not fin or (opt is None and d > 1e-3 * max(1.0, r.float().abs().max().item())):
            msg.append("out%d finite=%s maxdiff=%.3g" % (i, fin, d))
    print("replay %d: loss %.5f (ref %.5f) %s" % (it, out[0].item(), ref[0].item(), "; ".join(msg) if msg else "ok"), flush=True)
