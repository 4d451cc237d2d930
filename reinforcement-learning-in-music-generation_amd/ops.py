"""torch.autograd bindings of the libcwlt kernels (one Function per C-ABI fwd/bwd pair)."""
import torch

from . import _lib

CLA_EPS = 1e-6  # fast_transformers CausalLinearAttention default eps


def _row_stride(t):
    """(N, L, H, D) view whose last two dims are dense and whose batch stride is L*row_stride."""
    N, L, H, D = t.shape
    if t.stride(3) != 1 or t.stride(2) != D:
        return None
    ld = t.stride(1)
    if N > 1 and t.stride(0) != L * ld:
        return None
    if ld % 4 != 0 or t.data_ptr() % 16 != 0:
        return None
    return ld


def _as_rows(t):
    ld = _row_stride(t)
    if ld is None:
        t = t.contiguous()
        ld = _row_stride(t)
    return t, ld


class CausalLinearAttentionFn(torch.autograd.Function):
    """out = CLA(q, k, v): q, k, v (N, L, H, 64) raw projections, elu+1 applied inside the kernel.

    Replaces fast_transformers CausalLinearAttention.forward + causal_dot_product
    (reference call sites: dqn_policy/model.py:128-137,231-232).
    """

    @staticmethod
    def forward(ctx, q, k, v, eps=CLA_EPS):
        lib = _lib.load()
        N, L, H, D = q.shape
        if k.shape != q.shape or v.shape != q.shape:
            raise ValueError("causal linear attention needs q, k, v of one shape (N, L, H, D)")
        if q.dtype != k.dtype or q.dtype != v.dtype:
            raise TypeError("q, k, v dtypes differ")
        q, ldq = _as_rows(q)
        k, ldk = _as_rows(k)
        v, ldv = _as_rows(v)
        out = torch.empty((N, L, H, D), dtype=q.dtype, device=q.device)
        zinv = torch.empty((N, L, H), dtype=torch.float32, device=q.device)
        _lib.check(lib.cwlt_causal_linear_fwd(
            _lib.dev(q, "q"), _lib.dev(k, "k"), _lib.dev(v, "v"), _lib.dev(out), _lib.dev(zinv),
            N, H, L, D, ldq, ldk, ldv, H * D, float(eps), _lib.dtype_code(q.dtype), _lib.stream_ptr()),
            "cwlt_causal_linear_fwd")
        ctx.save_for_backward(q, k, v, out, zinv)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        q, k, v, out, zinv = ctx.saved_tensors
        N, L, H, D = q.shape
        dout, lddo = _as_rows(dout)
        _, ldq = _as_rows(q)
        _, ldk = _as_rows(k)
        _, ldv = _as_rows(v)
        # one (N, L, 3, H, D) buffer: dq|dk|dv side by side = the gradient of a fused QKV projection
        dqkv = torch.empty((N, L, 3, H, D), dtype=q.dtype, device=q.device)
        dq, dk, dv = dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]
        ld = 3 * H * D
        _lib.check(lib.cwlt_causal_linear_bwd(
            _lib.dev(q), _lib.dev(k), _lib.dev(v), _lib.dev(out), _lib.dev(zinv), _lib.dev(dout, "dout"),
            _lib.dev(dq), _lib.dev(dk), _lib.dev(dv), N, H, L, D,
            ldq, ldk, ldv, H * D, lddo, ld, ld, ld, _lib.dtype_code(q.dtype), _lib.stream_ptr()),
            "cwlt_causal_linear_bwd")
        return dq, dk, dv, None


def causal_linear_attention(q, k, v, eps=CLA_EPS):
    return CausalLinearAttentionFn.apply(q, k, v, eps)
