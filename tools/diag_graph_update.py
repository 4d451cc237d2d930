"""Diagnostic: graphed DQN.update at repo dims (GPU box).  usage: python tools/diag_graph_update.py {tuned|untuned} {bf16|f32}"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CWLT_TRAIN_GRAPHS", "1")
os.environ["CWLT_COMPUTE_DTYPE"] = sys.argv[2] if len(sys.argv) > 2 else "bf16"
import torch

import rlmg_amd  # noqa: F401
from rlmg_amd import gemm_tuning, ops
from rlmg_amd.dqn_policy import IRL_dqn_train as T


def say(msg):
    print(msg, flush=True)


def patch_parts(agent):
    """DIAG_PART=ce|td: keep only one of the two loss terms of DQN._update_device (bisecting the graph corruption)."""
    part = os.environ.get("DIAG_PART")
    if not part:
        return
    from rlmg_amd import rl_ops

    def upd(agent_state, agent_next_state, agent_action, agent_reward, agent_done, expert_next_state, mask):
        zero = torch.zeros((), device=agent_state.device)
        if part == "td":
            y = agent._fused(agent.eval_net, agent_state)
            with torch.no_grad():
                yt = agent._fused(agent.target_net, agent_next_state)
            mse = rl_ops.dqn_td_mse(y, yt, agent_action, agent_reward, agent_done, agent.n_class, T.GAMMA).sum() / 6
            ce, total = zero, mse
        else:
            c = agent.eval_net.train_step(agent_state, expert_next_state, mask)
            ce = (c[0] + c[1] + c[2] + c[3] + c[4] + c[5]) / 6
            mse, total = zero, ce
        agent.sync.zero_grad()
        total.backward()
        agent.sync.finish()
        agent.optim.step()
        return mse.detach(), ce.detach(), total.detach()

    agent._update_device = upd


def main():
    if os.environ.get("DIAG_NO_DIRECT") == "1":
        ops.direct_grads = lambda p=None: False
    if os.environ.get("DIAG_DELIVER") == "add":        # always accumulate (gradients are zeroed first)
        def deliver(pairs):
            d, s = [], []
            for p, g in pairs:
                if p.requires_grad:
                    d.append(p.grad)
                    s.append(g.view(p.grad.shape).to(p.grad.dtype))
            torch._foreach_add_(d, s)
        ops.deliver_grads = deliver
    if os.environ.get("DIAG_DELIVER") == "single":     # the fresh / accumulate logic, one tensor op per parameter
        def deliver1(pairs):
            for p, g in pairs:
                ops.deliver_grad(p, g)
        ops.deliver_grads = deliver1
    if os.environ.get("DIAG_BLAS"):
        torch.backends.cuda.preferred_blas_library(os.environ["DIAG_BLAS"])
        say("preferred blas: %s" % torch.backends.cuda.preferred_blas_library())
    if os.environ.get("DIAG_NO_WGRAD") == "1":
        ops.wgrad_supported = lambda a, b: False
    mode = sys.argv[1] if len(sys.argv) > 1 else "untuned"
    if mode == "tuned":
        say("tuned table: %s" % gemm_tuning.enable())
    n_class = [56, 135, 18, 87, 18, 25]
    with contextlib.redirect_stdout(io.StringIO()):
        agent = T.DQN(n_class, Pretrain=False)
    patch_parts(agent)
    g = torch.Generator().manual_seed(0)
    tok = lambda *s: torch.stack([torch.randint(0, c, s, generator=g) for c in n_class], -1).cuda()  # noqa: E731
    B = 30
    m = torch.ones(B, 50).cuda()
    inter = os.environ.get("DIAG_INTERLEAVE", "")
    rew = None
    if inter == "score":
        from rlmg_amd.dqn_policy.AIRL import RewardDiscri
        os.makedirs("gpurun_out/diag_run", exist_ok=True)
        os.chdir("gpurun_out/diag_run")
        with contextlib.redirect_stdout(io.StringIO()):
            rew = RewardDiscri(n_class, Pretrain=False)
        if os.environ.get("DIAG_REW_DTYPE") == "f32":
            rew.disc_model.compute_dtype = torch.float32
        elif os.environ.get("DIAG_REW_DTYPE") == "bf16":
            rew.disc_model.compute_dtype = torch.bfloat16
        buf_s, buf_m, buf_d = tok(2000, 50), torch.ones(2000, 50).cuda(), torch.zeros(2000, 1).cuda()
    big_a = torch.randn(100000, 512, device="cuda").bfloat16() if inter == "gemm" else None
    big_w = torch.randn(1536, 512, device="cuda").bfloat16() if inter == "gemm" else None
    n_upd = int(os.environ.get("DIAG_UPDATES", "8"))
    if os.environ.get("DIAG_NOGC") == "1":
        import gc
        gc.disable()
    for i in range(n_upd):
        tr = {"state": tok(B, 50), "action": tok(B, 25), "reward": torch.rand(B, 1), "nextstate": tok(B, 50),
              "done": torch.zeros(B, 1)}
        with contextlib.redirect_stdout(io.StringIO()):
            out = agent.update(tr, dict(tr), m, False, 0)
        torch.cuda.synchronize()
        say("update %d ok %s graphs=%d" % (i, ["%.4f" % v for v in out], len(getattr(agent, "_graph_update").graphs)
                                          if getattr(agent, "_graph_update", None) else 0))
        if inter == "score":
            rew.calculate_reward(buf_s, buf_d, buf_s, buf_m, buf_m)
            rew.calculate_reward(buf_s, buf_d, buf_s, buf_m, buf_m)
        elif inter == "gemm":
            for _ in range(20):
                torch.mm(big_a, big_w.t())
        elif inter == "alloc":
            tmp = [torch.full((n,), float("nan"), dtype=torch.bfloat16, device="cuda")
                   for n in (1 << 8, 1 << 12, 1 << 16, 1 << 19, 1 << 20, 1 << 22, 1 << 24, 1 << 26, 1 << 28) for _ in range(4)]
            del tmp
        x = tok(1, 50)
        agent.choose_action(x)
        torch.cuda.synchronize()
        say("  choose_action ok")
    say("done")


if __name__ == "__main__":
    main()
