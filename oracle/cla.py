"""Causal linear attention -- CPU oracle (TEST INFRASTRUCTURE; see oracle/__init__.py).

Restates pytorch-fast-transformers==0.4.0 (requirements.txt:54; NOT under /root/reference, parity
unpinned) `attention/causal_linear_attention.py::CausalLinearAttention.forward` as the reference
reaches it through attention_type="causal-linear" (dqn_policy/model.py:128-137,231-232):

    Q = elu(q)+1 ; K = elu(k)+1                      (feature_maps: elu_feature_map)
    Z = 1 / (einsum("nlhi,nlhi->nlh", Q, K.cumsum(1)) + eps)          eps = 1e-6
    V = causal_dot_product(Q, K, v)   ;  out = V * Z[..., None]

Tensors use the package's own layout: q, k (N, L, H, E), v (N, L, H, M).
Three formulations that must agree: quadratic-masked, cumulative-sum, recurrent
(`RecurrentLinearAttention`, the form dqn_policy/model.py:141-150,236-238 uses for generation).
"""
import torch
import torch.nn.functional as F

EPS = 1e-6


def feature_map(x):
    """elu(x) + 1 (fast_transformers.feature_maps.elu_feature_map)."""
    return F.elu(x) + 1


def cla_quadratic(q, k, v, eps=EPS):
    """O(L^2): scores (Q K^T) masked lower-triangular, row-normalised by their own sum + eps."""
    Q, K = feature_map(q), feature_map(k)
    L = q.shape[1]
    s = torch.einsum("nlhe,nshe->nhls", Q, K)
    s = s * torch.tril(torch.ones(L, L, dtype=q.dtype, device=q.device))
    den = s.sum(-1) + eps                                   # (N, H, L)
    out = torch.einsum("nhls,nshm->nlhm", s, v)
    return out / den.permute(0, 2, 1)[..., None]


def cla_cumsum(q, k, v, eps=EPS):
    """The package's own formulation: K.cumsum normaliser + prefix sum of K (x) V outer products."""
    Q, K = feature_map(q), feature_map(k)
    Z = 1.0 / (torch.einsum("nlhi,nlhi->nlh", Q, K.cumsum(1)) + eps)
    kv = torch.einsum("nlhe,nlhm->nlhem", K, v).cumsum(1)   # causal_dot_product's running E x M state
    V = torch.einsum("nlhe,nlhem->nlhm", Q, kv)
    return V * Z[..., None]


def cla_recurrent_step(q_t, k_t, v_t, state=None, eps=EPS):
    """One token of RecurrentLinearAttention: q_t, k_t (N, H, E), v_t (N, H, M); state = [S, Zs]."""
    Q, K = feature_map(q_t), feature_map(k_t)
    if state is None:
        N, H, E = Q.shape
        S = q_t.new_zeros((N, H, E, v_t.shape[-1]))
        Zs = q_t.new_zeros((N, H, E))
    else:
        S, Zs = state
    Zs = Zs + K
    S = S + torch.einsum("nhd,nhm->nhdm", K, v_t)
    Z = 1.0 / (torch.einsum("nhd,nhd->nh", Q, Zs) + eps)
    V = torch.einsum("nhd,nhdm,nh->nhm", Q, S, Z)
    return V, [S, Zs]


def cla_recurrent(q, k, v, eps=EPS):
    state = None
    outs = []
    for t in range(q.shape[1]):
        o, state = cla_recurrent_step(q[:, t], k[:, t], v[:, t], state, eps)
        outs.append(o)
    return torch.stack(outs, 1)


def cla_chunked(q, k, v, eps=EPS, chunk=64):
    """Linear-cost form: quadratic inside chunks of `chunk` tokens, running (E x M) state between
    chunks -- what a CPU implementation of causal_dot_product amounts to, BLAS-friendly."""
    Q, K = feature_map(q), feature_map(k)
    N, L, H, E = Q.shape
    M = v.shape[-1]
    S = q.new_zeros((N, H, E, M))
    z = q.new_zeros((N, H, E))
    outs = []
    for c0 in range(0, L, chunk):
        Qc, Kc, Vc = Q[:, c0:c0 + chunk], K[:, c0:c0 + chunk], v[:, c0:c0 + chunk]
        C = Qc.shape[1]
        A = torch.einsum("nlhe,nshe->nhls", Qc, Kc) * torch.tril(torch.ones(C, C, dtype=q.dtype, device=q.device))
        num = torch.einsum("nhls,nshm->nlhm", A, Vc) + torch.einsum("nlhe,nhem->nlhm", Qc, S)
        den = A.sum(-1).permute(0, 2, 1) + torch.einsum("nlhe,nhe->nlh", Qc, z) + eps
        outs.append(num / den[..., None])
        S = S + torch.einsum("nshe,nshm->nhem", Kc, Vc)
        z = z + Kc.sum(1)
    return torch.cat(outs, 1)


def cla_reference(q, k, v, eps=EPS):
    """Default oracle form: quadratic for short sequences, chunked (same arithmetic, linear cost) beyond."""
    if q.shape[1] <= 256:
        return cla_quadratic(q, k, v, eps)
    return cla_chunked(q, k, v, eps)


def cla_grads(q, k, v, dout, fn=cla_quadratic, eps=EPS):
    """(out, dq, dk, dv) by autograd through `fn`."""
    q, k, v = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    out = fn(q, k, v, eps)
    out.backward(dout)
    return out.detach(), q.grad, k.grad, v.grad
