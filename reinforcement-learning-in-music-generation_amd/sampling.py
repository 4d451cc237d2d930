"""Host-side (numpy) token samplers used at generation time -- dqn_policy/model.py:19-55,
ppo_policy/model.py:28-64.  Not on the training hot path; kept for the class surface."""
import numpy as np


def softmax_with_temperature(logits, temperature):
    z = np.exp(logits / temperature)
    return z / np.sum(z)


def weighted_sampling(probs):
    probs = probs / sum(probs)
    order = np.argsort(probs)[::-1]
    return np.random.choice(order, size=1, p=probs[order])[0]


def nucleus(probs, p):
    probs = probs / (sum(probs) + 1e-5)
    order = np.argsort(probs)[::-1]
    cusum = np.cumsum(probs[order])
    after = cusum > p
    if after.sum() > 0:
        last = np.where(after)[0][0] + 1
        cand = order[:last]
    else:
        cand = order[:]
    cp = np.array([probs[i] for i in cand])
    cp = cp / cp.sum()
    return np.random.choice(cand, size=1, p=cp)[0]


def sampling(logit, p=None, t=1.0):
    logit = logit.squeeze().detach().cpu().numpy()
    probs = softmax_with_temperature(logits=logit, temperature=t)
    if p is not None:
        return nucleus(probs, p=p)
    return weighted_sampling(probs)
