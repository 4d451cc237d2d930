/* cwlt.h -- C-ABI of libcwlt.so: the MI355X (gfx950) hot path of the compound-word (CW)
 * Linear-Transformer + AIRL / PPO / DQN training path.
 *
 * Every entry point
 *   - is `extern "C"`, takes plain device pointers + sizes (no torch types),
 *   - enqueues on the hipStream_t passed as `stream` (void*; 0 = default stream) and returns
 *     without synchronising, never allocates, never throws, keeps no global mutable state,
 *   - returns 0 on success, a hipError_t value (1..999) if the launch failed, or
 *     CWLT_ERR_ARG (1001) / CWLT_ERR_DTYPE (1002) for arguments it refuses.
 * Workspaces are allocated by the caller and passed in.
 *
 * File:line citations are relative to the reference checkout (/root/reference).
 * `dtype` arguments: CWLT_F32 = 0, CWLT_BF16 = 1 (storage type of activations; all arithmetic
 * and every running state is f32).
 */
#ifndef CWLT_H
#define CWLT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CWLT_OK 0
#define CWLT_ERR_ARG 1001
#define CWLT_ERR_DTYPE 1002
#define CWLT_F32 0
#define CWLT_BF16 1

/* ABI version of this header; bumped on any signature change. */
int cwlt_abi_version(void);

/* ---- causal linear attention -------------------------------------------------------------------
 * Replaces fast_transformers (pytorch-fast-transformers==0.4.0, requirements.txt:54)
 * `CausalLinearAttention.forward` incl. its `causal_dot_product` extension, reached from
 * dqn_policy/model.py:128-137,231-232; dqn_policy/agent_pretrain.py:244-253,345-346;
 * ppo_policy/model.py:129-138,233-234,313-321,371-372 (attention_type="causal-linear").
 *
 * q, k, v, out: (N, L, H, head_dim) with token-row stride ld* elements (head h at column
 * h*head_dim, batch stride L*ld*), RAW projections -- the elu(x)+1 feature map is applied inside.
 * zinv: (N, L, H) f32, 1/(phi(q_l).sum_{j<=l} phi(k_j) + eps), written by fwd, read by bwd.
 * head_dim must be 64 (d_model 512 / 8 heads, dqn_policy/config.py:11-15). */
int cwlt_causal_linear_fwd(const void* q, const void* k, const void* v, void* out, float* zinv,
                           int N, int H, int L, int head_dim,
                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                           float eps, int dtype, void* stream);

/* dq, dk, dv are gradients w.r.t. the RAW q, k, v (feature-map derivative applied inside);
 * out / zinv are the forward's outputs, dout the gradient w.r.t. out (row stride lddo). */
int cwlt_causal_linear_bwd(const void* q, const void* k, const void* v, const void* out,
                           const float* zinv, const void* dout, void* dq, void* dk, void* dv,
                           int N, int H, int L, int head_dim,
                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                           int64_t lddq, int64_t lddk, int64_t lddv, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CWLT_H */
