"""Python bindings of the RL arithmetic kernels (csrc/rl.hip) + autograd wrappers."""
import torch

from . import _lib, ops
from .ops import _call


def rollout_gather(ids, probs, n_class, n_actions, mode):
    """ids (R, T, A) int64, probs (R, T, W) f32 or None -> action (R, NA, A) [, logp (R, NA, A)]."""
    R, T, A = ids.shape
    ids = ids.contiguous()
    action = torch.empty((R, n_actions, A), dtype=torch.int64, device=ids.device)
    logp = None
    ldp = 0
    if mode != 0:
        probs = probs.contiguous()
        ldp = probs.shape[-1]
        logp = torch.empty((R, n_actions, A), dtype=torch.float32, device=ids.device)
    _call("cwlt_rollout_gather", _lib.dev(ids, "ids"), _lib.opt(probs), _lib.int_array(n_class), A, _lib.dev(action),
          _lib.opt(logp), R, T, n_actions, ldp, mode, _lib.stream_ptr())
    return action, logp


def ppo_returns_adv(rewards, values, gamma, normalize=True):
    """rewards, values: (E,) or (E, 1) f32 on GPU -> returns (E, 1), advantages (E, 1)."""
    E = rewards.numel()
    r = rewards.reshape(E).float().contiguous()
    v = values.reshape(E).float().contiguous()
    ret = torch.empty(E, dtype=torch.float32, device=r.device)
    adv = torch.empty(E, dtype=torch.float32, device=r.device)
    _call("cwlt_ppo_returns_adv", _lib.dev(r, "rewards"), _lib.dev(v), _lib.dev(ret), _lib.dev(adv), E, float(gamma),
          1 if normalize else 0, _lib.stream_ptr())
    return ret.unsqueeze(1), adv.unsqueeze(1)


class PPOPolicyLossFn(torch.autograd.Function):
    """-mean(min(0.2*A, clamp(exp(new - old_int), 1-clip, 1+clip) * A))  (ppo_train.py:388-396)."""

    @staticmethod
    def forward(ctx, new_logp, old_logp_int, adv, clip):
        NA, A = new_logp.shape
        E = old_logp_int.shape[0]
        nl = new_logp.float().contiguous()
        old = old_logp_int.reshape(E, NA, A).to(torch.int64).contiguous()
        a = adv.reshape(E).float().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=nl.device)
        grad = torch.empty((NA, A), dtype=torch.float32, device=nl.device)
        _call("cwlt_ppo_policy_loss", _lib.dev(nl, "new_logp"), _lib.dev(old), _lib.dev(a), _lib.dev(loss),
              _lib.dev(grad), E, NA, A, float(clip), _lib.stream_ptr())
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


def ppo_policy_loss(new_logp, old_logp_int, adv, clip):
    return PPOPolicyLossFn.apply(new_logp, old_logp_int, adv, clip)


class LogPArgmaxFn(torch.autograd.Function):
    """rows of fused logits -> log softmax value at the per-attribute argmax, differentiable."""

    @staticmethod
    def forward(ctx, logits, n_class):
        logits = logits.contiguous()
        res = ops.heads_forward(logits, n_class, want_argmax=True, want_pmax=True)
        ctx.save_for_backward(logits, res["argmax"])
        ctx.n_class = n_class
        ctx.mark_non_differentiable(res["argmax"])
        return torch.log(res["pmax"]), res["argmax"]

    @staticmethod
    def backward(ctx, g, _):
        logits, ids = ctx.saved_tensors
        n_class = ctx.n_class
        w = (-g).float().contiguous()
        dl = torch.empty_like(logits)
        _call("cwlt_heads_logp_bwd", _lib.dev(logits), _lib.int_array(n_class), len(n_class), _lib.dev(ids),
              _lib.dev(w), _lib.dev(dl), logits.shape[0], logits.shape[1], _lib.dtype_code(logits.dtype),
              _lib.stream_ptr())
        return dl, None


def logp_argmax(logits, n_class):
    return LogPArgmaxFn.apply(logits, tuple(int(n) for n in n_class))


class DQNTDLossFn(torch.autograd.Function):
    """MSE part of DQN.update (IRL_dqn_train.py:285-330) -> (A,) per-attribute MSEs; gradient flows to the
    eval-net logits only (rows of batch element 0 -- the reference's gather quirk)."""

    @staticmethod
    def forward(ctx, y, yt, action, reward, done, n_class, gamma):
        B, T, W = y.shape
        NA, A = action.shape[1], action.shape[2]
        y32, yt32 = y.float().contiguous(), yt.float().contiguous()
        action = action.to(torch.int64).contiguous()
        r = reward.reshape(B).float().contiguous()
        d = done.reshape(B).float().contiguous()
        part = torch.empty((B, A), dtype=torch.float32, device=y.device)
        dq = torch.empty((B, NA, A), dtype=torch.float32, device=y.device)
        _call("cwlt_dqn_td_fwd", _lib.dev(y32, "y"), _lib.dev(yt32), _lib.int_array(n_class), A, _lib.dev(action),
              _lib.dev(r), _lib.dev(d), _lib.dev(part), _lib.dev(dq), B, T, NA, W, float(gamma), _lib.stream_ptr())
        ctx.save_for_backward(dq, action)
        ctx.meta = (B, T, W, NA, A, n_class, y.dtype)
        return part.sum(0) / float(B * NA)

    @staticmethod
    def backward(ctx, g):
        dq, action = ctx.saved_tensors
        B, T, W, NA, A, n_class, dt = ctx.meta
        # MSEloss = sum_f mse_f / A was folded into dq; per-attribute upstream gradients g[f] * A rescale it
        dqg = (dq * (g.float() * A).view(1, 1, A)).contiguous()
        dy = torch.zeros((B, T, W), dtype=torch.float32, device=dq.device)
        one = torch.ones(1, dtype=torch.float32, device=dq.device)
        _call("cwlt_dqn_td_bwd", _lib.dev(dqg), _lib.dev(action), _lib.int_array(n_class), A, _lib.dev(dy),
              _lib.dev(one), B, NA, W, _lib.stream_ptr())
        return dy.to(dt), None, None, None, None, None, None


def dqn_td_mse(y, yt, action, reward, done, n_class, gamma):
    return DQNTDLossFn.apply(y, yt, action, reward, done, tuple(int(n) for n in n_class), gamma)
