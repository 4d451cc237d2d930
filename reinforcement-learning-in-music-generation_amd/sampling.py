"""Host-side (numpy) token samplers used at generation time -- dqn_policy/model.py:19-55,
ppo_policy/model.py:28-64.  They consume `np.random` exactly as the reference does (same number and order of
draws, same float32 arithmetic incl. Python's sequential `sum`), so a seeded run reproduces its token stream.
"""
import numpy as np


def softmax_with_temperature(logits, temperature):
    return np.exp(logits / temperature) / np.sum(np.exp(logits / temperature))


def weighted_sampling(probs):
    probs = probs / sum(probs)
    order = np.argsort(probs)[::-1]
    return np.random.choice(order, size=1, p=np.sort(probs)[::-1])[0]


def nucleus(probs, p):
    probs = probs / (sum(probs) + 1e-5)
    order = np.argsort(probs)[::-1]
    cusum = np.cumsum(np.sort(probs)[::-1])
    after = cusum > p
    if sum(after) > 0:
        last = np.where(after)[0][0] + 1
        cand = order[:last]
    else:
        cand = order[:]
    cp = [probs[i] for i in cand]
    cp = cp / sum(cp)
    return np.random.choice(cand, size=1, p=cp)[0]


def sampling(logit, p=None, t=1.0):
    if not isinstance(logit, np.ndarray):
        logit = logit.squeeze().detach().cpu().numpy()
    probs = softmax_with_temperature(logits=logit, temperature=t)
    if p is not None:
        return nucleus(probs, p=p)
    return weighted_sampling(probs)


def sample_cw(y):
    """Next CW token from the six logit vectors (tempo, chord, barbeat, pitch, duration, velocity) with the
    reference's per-attribute temperature / nucleus settings.  The DRAW order is tempo, barbeat, chord, pitch,
    duration, velocity (dqn_policy/model.py:281-286); the returned array is in attribute order (:289-296)."""
    tempo = sampling(y[0], t=1.2, p=0.9)
    barbeat = sampling(y[2], t=1.2)
    chord = sampling(y[1], p=0.99)
    pitch = sampling(y[3], p=0.9)
    duration = sampling(y[4], t=2, p=0.9)
    velocity = sampling(y[5], t=5)
    return np.array([tempo, chord, barbeat, pitch, duration, velocity])
