"""Diagnostic (GPU box): re-create round 1's overwrite-on-first-delivery gradient path (torch._foreach_copy_ for a
parameter's first delivery of a step, torch._foreach_add_ for later ones) inside the captured DQN.update and find
WHICH buffer goes bad first when eager GEMMs run between replays.

    python tools/diag_fresh_copy.py [copy|mul_add|add] [gemm|none]

After every update: non-finite counts and max |x| of every gradient (by parameter), parameter, Adam state and graph
input; the first offender is printed with its delivery kind.  `copy` = round 1's path; `mul_add` = the same
overwrite semantics without _foreach_copy_ (dst *= 0 then dst += src); `add` = today's accumulate-only path."""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CWLT_TRAIN_GRAPHS", "1")
os.environ["CWLT_COMPUTE_DTYPE"] = "bf16"
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import gemm_tuning, ops
from rlmg_amd.dqn_policy import IRL_dqn_train as T


def say(msg):
    print(msg, flush=True)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "copy"
    inter = sys.argv[2] if len(sys.argv) > 2 else "gemm"
    gemm_tuning.enable()
    n_class = [56, 135, 18, 87, 18, 25]
    with contextlib.redirect_stdout(io.StringIO()):
        agent = T.DQN(n_class, Pretrain=False)
    names = {id(p): n for n, p in agent.eval_net.named_parameters()}
    fresh = {}
    kinds = {}

    real_zero = agent.sync.zero_grad

    def zero_grad():
        real_zero()
        for p in agent.eval_net.parameters():
            fresh[id(p)] = True

    agent.sync.zero_grad = zero_grad

    def deliver(pairs):
        fd, fs, ad, as_ = [], [], [], []
        for p, g in pairs:
            if not p.requires_grad:
                continue
            src = g.view(p.grad.shape)
            if mode != "add" and fresh.get(id(p), False):
                fd.append(p.grad)
                fs.append(src)
                fresh[id(p)] = False
                kinds.setdefault(names[id(p)], []).append("first")
            else:
                ad.append(p.grad)
                as_.append(src if src.dtype == p.grad.dtype else src.to(p.grad.dtype))
                kinds.setdefault(names[id(p)], []).append("later")
        if fd:
            if mode == "copy":
                torch._foreach_copy_(fd, fs)
            else:
                torch._foreach_mul_(fd, 0.0)
                torch._foreach_add_(fd, [s if s.dtype == d.dtype else s.to(d.dtype) for s, d in zip(fs, fd)])
        if ad:
            torch._foreach_add_(ad, as_)

    ops.deliver_grads = deliver
    g = torch.Generator().manual_seed(0)
    tok = lambda *s: torch.stack([torch.randint(0, c, s, generator=g) for c in n_class], -1).cuda()  # noqa: E731
    B = 30
    m = torch.ones(B, 50).cuda()
    big_a = torch.randn(100000, 512, device="cuda").bfloat16()
    big_w = torch.randn(1536, 512, device="cuda").bfloat16()
    ref_sum = None

    def stats(t):
        t = t.detach().float()
        bad = (~torch.isfinite(t)).sum().item()
        mx = t[torch.isfinite(t)].abs().max().item() if bad < t.numel() else float("nan")
        return bad, mx

    for i in range(int(os.environ.get("DIAG_UPDATES", "8"))):
        tr = {"state": tok(B, 50), "action": tok(B, 25), "reward": torch.rand(B, 1), "nextstate": tok(B, 50),
              "done": torch.zeros(B, 1)}
        with contextlib.redirect_stdout(io.StringIO()):
            out = agent.update(tr, dict(tr), m, False, 0)
        torch.cuda.synchronize()
        ng = len(agent._graph_update.graphs) if getattr(agent, "_graph_update", None) else 0
        say("update %d losses %s graphs=%d" % (i, ["%.4f" % v for v in out], ng))
        worst = []
        for n, p in agent.eval_net.named_parameters():
            for what, t in (("grad", p.grad), ("param", p)):
                if t is None:
                    continue
                bad, mx = stats(t)
                if bad or mx > 1e4:
                    worst.append((what, n, bad, mx, (kinds.get(n) or ["autograd"])[-2:]))
        for st_name in ("exp_avg", "exp_avg_sq"):
            for p, st in agent.optim.state.items():
                bad, mx = stats(st[st_name])
                if bad or mx > 1e6:
                    worst.append((st_name, names[id(p)], bad, mx, ""))
        if ng:
            for j, s in enumerate(agent._graph_update.graphs[next(iter(agent._graph_update.graphs))][1]):
                bad, mx = stats(s)
                if bad:
                    worst.append(("static_input", str(j), bad, mx, ""))
        b_bad, b_mx = stats(big_a)
        w_bad, w_mx = stats(big_w)
        if b_bad or w_bad:
            worst.append(("eager big_a/big_w", "", b_bad + w_bad, max(b_mx, w_mx), ""))
        s_now = (big_a.float().sum() + big_w.float().sum()).item()
        if ref_sum is None:
            ref_sum = s_now
        elif s_now != ref_sum:
            worst.append(("eager big_a/big_w CHANGED", "", 0, s_now - ref_sum, ""))
        for w in worst[:12]:
            say("    BAD %s %s nonfinite=%d max=%.3e kinds=%s" % w)
        if worst:
            say("    (%d offenders)" % len(worst))
        if inter == "gemm":
            for _ in range(20):
                torch.mm(big_a, big_w.t())
        torch.cuda.synchronize()
    say("done")


if __name__ == "__main__":
    main()
