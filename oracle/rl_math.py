"""RL arithmetic of the reference, restated with plain torch on CPU (TEST INFRASTRUCTURE; see
oracle/__init__.py).  Each function follows the cited reference lines literally, quirks included
(SURVEY §8a): the code shapes below are the reference's own expressions with the I/O stripped.
"""
import torch
import torch.nn.functional as F


def dqn_choose_action(logits, n_actions):
    """dqn_policy/IRL_dqn_train.py:240-264.  logits: 6 tensors (1, T, n_f).  -> (n_actions, 6) int64.
    `-idx` with idx = 0 is position 0, so rows are positions [0, T-1, T-2, ...]."""
    m = torch.nn.Softmax(dim=-1)
    ids = [torch.argmax(m(y), dim=-1) for y in logits]
    action = None
    for idx in range(n_actions):
        tmp = torch.cat([i[:, -idx] for i in ids], dim=0).unsqueeze(0)
        action = tmp if action is None else torch.cat((action, tmp), dim=0)
    return action


def ppo_choose_action(logits, n_actions):
    """ppo_policy/ppo_train.py:259-290.  -> action (n_actions, 6), log_prob (n_actions, 6).
    tempo / chord log-probs use the class chosen at position +idx (reference quirk, :273-274)."""
    m = torch.nn.Softmax(dim=-1)
    ys = [m(y) for y in logits]
    ids = [torch.argmax(y, dim=-1) for y in ys]
    action, logp = None, None
    for idx in range(1, n_actions + 1):
        tmp = torch.cat([i[:, -idx] for i in ids], dim=0).unsqueeze(0)
        prob = torch.cat([ys[0][0, -idx, ids[0][:, idx]], ys[1][0, -idx, ids[1][:, idx]]] +
                         [ys[f][0, -idx, ids[f][:, -idx]] for f in range(2, 6)], dim=0)
        lp = torch.log(prob).unsqueeze(0)
        action = tmp if action is None else torch.cat((action, tmp), dim=0)
        logp = lp if logp is None else torch.cat((logp, lp), dim=0)
    return action, logp


def ppo_select_update(logits, n_actions):
    """ppo_policy/ppo_train.py:300-346: greedy rows / log-probs for every batch element, but only the LAST
    batch element's (n_actions, 6) pair is returned (:346)."""
    m = torch.nn.Softmax(dim=-1)
    ys = [m(y) for y in logits]
    ids = [torch.argmax(y, dim=-1) for y in ys]
    b = logits[0].shape[0] - 1
    action = torch.stack([torch.stack([ids[f][b, -idx] for f in range(6)]) for idx in range(1, n_actions + 1)])
    logp = torch.stack([torch.stack([torch.log(ys[f][b, -idx, ids[f][b, -idx]]) for f in range(6)])
                        for idx in range(1, n_actions + 1)])
    return action, logp


def ppo_returns(rewards, discount, normalize=True):
    """ppo_policy/ppo_train.py:348-357 (iterates rewards in FORWARD order, inserting at the front)."""
    returns, R = [], 0
    for r in rewards:
        R = r + R * discount
        returns.insert(0, R)
    returns = torch.tensor([float(x) for x in returns]).unsqueeze(1)
    if normalize:
        returns = (returns - returns.mean()) / returns.std()
    return returns


def ppo_advantages(returns, values, normalize=True):
    """ppo_policy/ppo_train.py:359-363."""
    adv = returns - values
    if normalize:
        adv = (adv - adv.mean()) / adv.std()
    return adv


def ppo_policy_loss(new_logp, log_actions_int, advantages, clip):
    """ppo_policy/ppo_train.py:388-396.  new_logp (NA, 6) f32, log_actions_int (E, NA, 6) int64 (the buffer
    returns stored log-probs through `.long()`, :122,135), advantages (E, 1)."""
    ratio = (new_logp - log_actions_int).exp()
    l1 = (0.2 * advantages).unsqueeze(2)
    l2 = torch.clamp(ratio, min=1.0 - clip, max=1.0 + clip) * advantages.unsqueeze(2)
    return -torch.min(l1, l2).mean()


def dqn_td_loss(y, yt, action, reward, done, gamma, n_actions):
    """dqn_policy/IRL_dqn_train.py:285-330.  y, yt: 6 tensors (B, T, n_f) from eval / target nets; action
    (B, NA, 6) int64; reward (B, 1) float; done (B, 1) int64.  -> (MSEloss, [6 per-attribute MSEs])."""
    losses = []
    for f in range(6):
        qval = y[f].gather(2, action[:, :, f].unsqueeze(0)).squeeze(0)        # index (1, B, NA): batch 0 only
        nxt = yt[f].max(2)[0]
        top, _ = nxt.topk(n_actions, dim=1)
        tgt = reward + gamma * (1 - done) * top
        losses.append(F.mse_loss(qval, tgt))
    return sum(losses) / 6, losses
