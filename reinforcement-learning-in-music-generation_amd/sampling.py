"""Host-side (numpy) token samplers used at generation time -- dqn_policy/model.py:19-55,
ppo_policy/model.py:28-64.  They consume `np.random` exactly as the reference does (one uniform double per draw,
in the reference's order of draws) and do its float32 arithmetic step for step, so a seeded run reproduces its
token stream bit for bit (tests/golden/dqn_generation_small.npz; tests/test_generation_cpu.py also checks them
against a line-by-line restatement on random logits).

They are on the critical path of host-sampled generation (one call per attribute per token), so the Python-level
loops of the reference are replaced by numpy calls with the SAME rounding:
  * `sum(x)` over a float32 array (sequential float32 adds)  ==  `np.cumsum(x)[-1]` (cumsum is sequential too);
  * `np.random.choice(a, size=1, p=p)`  ==  `a[searchsorted(cumsum(float64(p)) / total, random_sample(), 'right')]`,
    which is what RandomState.choice does after validating p; non-finite p goes to np.random.choice itself so
    that it raises what the reference raises.
"""
import numpy as np


def _seq_sum(x):
    """Python's sum() over a float32 vector: left-to-right float32 additions."""
    return x.cumsum()[-1]


def _choice(values, p):
    """np.random.choice(values, size=1, p=p)[0] without its argument validation (same draw, same result)."""
    cdf = p.astype(np.float64).cumsum()
    total = cdf[-1]
    if not abs(total - 1.0) <= 1e-4:                              # NaN / inf / not normalised (p >= 0: it is exp / sum)
        return np.random.choice(values, size=1, p=p)[0]          # let numpy raise / decide, as in the reference
    cdf /= total
    return values[cdf.searchsorted(np.random.random_sample(), side="right")]


def softmax_with_temperature(logits, temperature):
    e = np.exp(logits / temperature)
    return e / np.sum(e)


def weighted_sampling(probs):
    probs = probs / _seq_sum(probs)
    order = probs.argsort()[::-1]
    return _choice(order, probs[order])


def nucleus(probs, p):
    probs = probs / (_seq_sum(probs) + 1e-5)
    order = probs.argsort()[::-1]
    sorted_probs = probs[order]
    hit = (sorted_probs.cumsum() > p).nonzero()[0]
    cand = order[:hit[0] + 1] if hit.size else order
    cp = sorted_probs[:cand.size]
    cp = cp / _seq_sum(cp)
    return _choice(cand, cp)


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
