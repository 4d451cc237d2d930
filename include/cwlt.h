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
 * head_dim must be 64 (d_model 512 / 8 heads, dqn_policy/config.py:11-15).
 *
 * Few streams: with N * H far below the 256 CUs one workgroup per (sequence, head) leaves the chip idle (the
 * reference's own pretrain batch is 4 sequences: dqn_policy/agent_pretrain.py:48).  `segments` > 1 cuts every
 * sequence into that many runs of whole 64-token chunks, one workgroup each: a first pass reduces each run to its
 * state increment, a prefix pass turns increments into starting states, the scan proper starts every run from its
 * state.  cwlt_scan_segments() is the library's choice (1 once the streams fill the chip; bf16 with row strides % 8
 * == 0 only); seg_ws: cwlt_scan_seg_floats(N, H, segments, backward) floats, NULL when segments == 1.  The backward
 * calls share ONE workspace: dkdv (first) fills it, dq reads it.  A run owns ceil(chunks / segments) chunks, and every
 * run must own at least one: (segments - 1) * ceil(chunks / segments) < chunks, chunks = ceil(L / 64) -- e.g. 9 chunks
 * take 3 segments, not 4 (3 + 3 + 3 + 0); a count that leaves a run empty is CWLT_ERR_ARG.
 *
 * final_state (optional; bf16, row strides % 8 == 0, segments == 1): cwlt_scan_final_state_floats(N, H) floats that
 * receive, per (sequence, head), the scan's state after the last token -- sum_j phi(k_j) v_j^T transposed (64 x 64 f32)
 * followed by sum_j phi(k_j) (64 f32).  It is what cwlt_causal_linear_bwd_sweep needs to run the whole backward in one
 * pass. */
int cwlt_scan_segments(int N, int H, int L, int dtype);
int64_t cwlt_scan_seg_floats(int N, int H, int segments, int backward);
int64_t cwlt_scan_final_state_floats(int N, int H);
int cwlt_causal_linear_fwd(const void* q, const void* k, const void* v, void* out, float* zinv,
                           int N, int H, int L, int head_dim,
                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                           float eps, int segments, float* seg_ws, float* final_state, int dtype, void* stream);

/* The whole backward (dq, dk, dv and, optionally, their per-(sequence, head) column sums: colsum_* (N, H*64) f32, all
 * three or none) in ONE reverse sweep: q, k, v, out, dout are each read once -- 8 streams of HBM traffic against the 12
 * of the dkdv + dq pair below.  The dQ scan needs the state of EARLIER tokens while the sweep runs from the last token
 * to the first: it starts from the forward's final_state and removes each chunk's contribution as it passes (f32
 * accumulators).  bf16 with row strides % 8 == 0 only (CWLT_ERR_DTYPE / CWLT_ERR_ARG otherwise); one 8-wave workgroup
 * per (sequence, head), so meant for launches whose N * H fills the chip -- few-stream launches use the segmented pair.
 * Same reference call sites as above (the backward of causal_dot_product). */
int cwlt_causal_linear_bwd_sweep(const void* q, const void* k, const void* v, const void* out, const float* zinv,
                                 const void* dout, const float* final_state, void* dq, void* dk, void* dv,
                                 float* colsum_q, float* colsum_k, float* colsum_v,
                                 int N, int H, int L, int head_dim,
                                 int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                                 int64_t lddq, int64_t lddk, int64_t lddv, int dtype, void* stream);

/* dq, dk, dv are gradients w.r.t. the RAW q, k, v (feature-map derivative applied inside);
 * out / zinv are the forward's outputs, dout the gradient w.r.t. out (row stride lddo). */
int cwlt_causal_linear_bwd(const void* q, const void* k, const void* v, const void* out,
                           const float* zinv, const void* dout, void* dq, void* dk, void* dv,
                           int N, int H, int L, int head_dim,
                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                           int64_t lddq, int64_t lddk, int64_t lddv, int dtype, void* stream);

/* The two kernels of the backward, separately launchable (the call above = dkdv then dq):
 * reverse scan producing dk, dv; forward scan producing dq.  colsum_* (each (N, H*head_dim) f32, may be
 * NULL; bf16 tensors with row strides % 8 == 0 only; (N * segments, H*head_dim) when segments > 1): per-sequence
 * (per-segment) column sums of the written gradient,
 * i.e. the partial bias gradients of the key / value / query projections (summed over N by the caller),
 * which saves a separate pass over dQ|dK|dV.
 * dden (N, L, H) f32, may be NULL: dden_l = -(dout_l . out_l) * zinv_l, the gradient through the normaliser.  Both
 * scans need it; when the caller passes one buffer to both calls (dkdv FIRST), dkdv writes it and dq reads it in place
 * of the whole `out` stream (bf16 fast path; `out` may then be NULL for dq).  The f32 kernels ignore it. */
int cwlt_causal_linear_bwd_dkdv(const void* q, const void* k, const void* v, const void* out,
                                const float* zinv, const void* dout, void* dk, void* dv,
                                float* colsum_k, float* colsum_v, float* dden,
                                int N, int H, int L, int head_dim,
                                int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                                int64_t lddk, int64_t lddv, int segments, float* seg_ws, int dtype, void* stream);
int cwlt_causal_linear_bwd_dq(const void* q, const void* k, const void* v, const void* out,
                              const float* zinv, const void* dout, void* dq, float* colsum_q, const float* dden,
                              int N, int H, int L, int head_dim,
                              int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                              int64_t lddq, int segments, const float* seg_ws, int dtype, void* stream);

/* ---- fused residual + dropout + LayerNorm ------------------------------------------------------
 * s = x + dropout_p(a) ; y = LayerNorm(s) * gamma + beta.  Replaces, inside fast_transformers'
 * TransformerEncoderLayer.forward (built at dqn_policy/model.py:128-137, called :232):
 *   x = x + self.dropout(attn(...)); x = self.norm1(x); ... ; self.norm2(x + self.dropout(ffn));
 * and TransformerEncoder's final self.norm(x) (x == NULL, p == 0).
 * x may be NULL (plain LayerNorm of a); s_out may be NULL; mean/rstd (rows) f32 are saved stats.
 * D % 8 == 0, D <= 1024.  Dropout masks are regenerated from (seed, element index) in backward.
 * seed_base (every entry point that takes a seed): optional device pointer to one uint64 that the kernel adds
 * to `seed` when it runs (NULL = 0).  `seed` is baked into a captured hipGraph; bumping *seed_base between
 * replays (itself a captured device op) gives each replay fresh masks.  Forward and backward of one op must
 * see the same seed + *seed_base. */
int cwlt_ln_blocks(int64_t rows);
int cwlt_add_dropout_layernorm_fwd(const void* x, const void* a, const float* gamma, const float* beta,
                                   void* s_out, void* y, float* mean, float* rstd,
                                   int64_t rows, int D, float eps, float p, uint64_t seed,
                                   const uint64_t* seed_base, int dtype, void* stream);
/* dy2 (optional second upstream gradient, summed with dy), ds = d/ds (also the residual gradient),
 * da = dropout-masked ds (NULL when p == 0: then da == ds).  part: cwlt_ln_blocks(rows)*3*D f32
 * workspace; stats: (3, D) f32 output = dgamma | dbeta | dbias, where dbias = column sum of da,
 * i.e. the bias gradient of the Linear that produced `a`. */
int cwlt_add_dropout_layernorm_bwd(const void* dy, const void* dy2, const void* s, const float* gamma,
                                   const float* mean, const float* rstd, void* ds, void* da,
                                   float* part, float* stats,
                                   int64_t rows, int D, float p, uint64_t seed, const uint64_t* seed_base,
                                   int dtype, void* stream);

/* ---- deterministic column sums (bias gradients) -------------------------------------------------
 * out[c] = sum_r x[r*ld + c]; part: cwlt_colsum_blocks(rows)*ncols f32.  Replaces the reductions
 * autograd runs for nn.Linear bias gradients (dqn_policy/model.py:123,156-161). */
int cwlt_colsum_blocks(int64_t rows);
int cwlt_colsum(const void* x, float* part, float* out, int64_t rows, int ncols, int64_t ld,
                int dtype, void* stream);

/* ---- FFN activation: g = dropout_p(gelu(h + bias)) ----------------------------------------------
 * Replaces `self.dropout(self.activation(self.linear1(y)))` (activation='gelu' = exact erf,
 * dqn_policy/model.py:134) with the Linear run bias-free.  bias may be NULL.  F % 8 == 0.
 * f32 tensors: libm erff / expf (the parity path).  bf16 tensors: Phi(-|x|) = exp(-x^2 / 2) * Q(|x|) with a degree-5
 * minimax Q -- value and derivative within 1.5e-4 (absolute) of the exact-erf forms, under the 2^-9 relative rounding of
 * a bf16 result; the GEMM-epilogue forms below use the same function, bit for bit.
 * gd (rows, F), may be NULL: also write the backward's factor gd = mask * 1/(1-p) * gelu'(h + bias), so that
 * dh = dg * gd is one multiply in the epilogue of the GEMM that produces dg (cwlt_gemm_nt_mul).  gd may be h itself
 * (in place: h is then consumed). */
int cwlt_rowslab_blocks(int64_t rows);
int cwlt_bias_gelu_dropout_fwd(const void* h, const float* bias, void* g, void* gd, int64_t rows, int F,
                               float p, uint64_t seed, const uint64_t* seed_base, int dtype, void* stream);
/* dh = dropout_bwd(dg) * gelu'(h + bias); dbias (F) f32 = column sums of dh (NULL to skip);
 * part: cwlt_rowslab_blocks(rows)*F f32 (needed iff dbias). */
int cwlt_bias_gelu_dropout_bwd(const void* dg, const void* h, const float* bias, void* dh,
                               float* part, float* dbias, int64_t rows, int F, float p,
                               uint64_t seed, const uint64_t* seed_base, int dtype, void* stream);

/* ---- FFN backward: dh = (dy . W2) * gd with the activation gradient in the GEMM epilogue --------
 * Replaces, in the backward of fast_transformers' TransformerEncoderLayer (dqn_policy/model.py:128-137), the input
 * gradient of `linear2` (a GEMM writing dg, rows x d_ff) followed by the backward of dropout(gelu(.)) above: the
 * product never reaches memory.  a = dy (M, K), w = linear2.weight TRANSPOSED, (N, K) row-major, g = gd (M, N) from
 * cwlt_bias_gelu_dropout_fwd, c = dh (M, N); all bf16, f32 accumulation, the product rounded to bf16 before the
 * multiply (the arithmetic of the two-kernel path).  colsum (N) f32 = column sums of c = linear1's bias gradient,
 * through part (cwlt_gemm_nt_tiles(M) * N floats), both or neither NULL.  N % 256 == 0, K % 64 == 0, row strides
 * multiples of 8, 16-byte aligned pointers. */
int64_t cwlt_gemm_nt_tiles(int64_t M);
int cwlt_gemm_nt_mul(const void* a, const void* w, const void* g, void* c, float* part, float* colsum,
                     int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldg, int64_t ldc, void* stream);

/* ---- FFN forward: linear1 + bias + GELU + dropout in the GEMM's epilogue ------------------------
 * x = bf16(a (M, K) . w (N, K)^T) + bias (N, f32);  g = dropout_p(gelu(x)),  gd = mask * 1/(1-p) * gelu'(x):
 * `self.dropout(self.activation(self.linear1(y)))` of fast_transformers' TransformerEncoderLayer
 * (dqn_policy/model.py:128-137, activation='gelu' = exact erf; evaluated as cwlt_bias_gelu_dropout_fwd evaluates it for
 * bf16 tensors) together with the factor cwlt_gemm_nt_mul multiplies the
 * upstream gradient with -- what a plain GEMM followed by cwlt_bias_gelu_dropout_fwd(gd) produces (same arithmetic,
 * the pre-activation rounded to bf16 as the GEMM's output would be, same dropout stream keyed by (seed, element
 * index)), without the (M, N) pre-activation's write and read.  w = linear1.weight as stored, (N, K) row-major.
 * g, gd: dense (M, N) bf16, distinct.  N % 256 == 0, K % 64 == 0, lda / ldw multiples of 8, 16-byte aligned. */
int cwlt_gemm_nt_bias_gelu_dropout(const void* a, const void* w, const float* bias, void* g, void* gd,
                                   int64_t M, int N, int K, int64_t lda, int64_t ldw, float p,
                                   uint64_t seed, const uint64_t* seed_base, void* stream);

/* The post-LN residual block in a GEMM's epilogue:  s = x + dropout(bf16(a (M, K) . w (N, K)^T) + bias);
 * y = LayerNorm(s) * gamma + beta;  mean, rstd per row -- `x = norm1(x + dropout(attention.out_projection(.)))` and
 * `norm2(x + dropout(linear2(.)))` of fast_transformers' post-LN TransformerEncoderLayer reached from
 * /root/reference/dqn_policy/model.py:128-137,231-232; replaces a GEMM with bf16 output followed by
 * cwlt_add_dropout_layernorm_fwd (same dropout stream (seed, row * N + column), same statistics, same rounding points;
 * the bias is added in f32).  The GEMM's output never reaches HBM.  a, w bf16 with row strides lda, ldw (multiples
 * of 8); x, s, y (M, N) bf16 dense; bias, gamma, beta (N) f32; mean, rstd (M) f32.  N == 512 (a workgroup owns whole
 * rows), K % 64 == 0, 16-byte aligned pointers, 0 <= p < 1. */
int cwlt_gemm_nt_bias_dropout_add_layernorm(const void* a, const void* w, const float* bias, const void* x,
                                            const float* gamma, const float* beta, void* s, void* y, float* mean,
                                            float* rstd, int64_t M, int N, int K, int64_t lda, int64_t ldw, float eps,
                                            float p, uint64_t seed, const uint64_t* seed_base, void* stream);

/* ---- the dense projections: C (M, N) [+]= A (M, K) . W (N, K)^T [+ bias] -------------------------
 * The `nn.Linear`s of fast_transformers' AttentionLayer / TransformerEncoderLayer (query / key / value / out projection,
 * linear1, linear2: /root/reference/dqn_policy/model.py:128-137) and the six output heads as one projection
 * (dqn_policy/model.py:156-161,241-249), forward and input-gradient forms -- what `torch.addmm / mm / addmm_` sent to
 * hipBLASLt.  a (M, K), w (N, K), c (M, N) bf16 row-major with row strides lda, ldw, ldc; f32 accumulation; bias (N)
 * f32 or NULL.  Forward: w = the layer's weight as stored.  Input gradient: w = the weight TRANSPOSED (dx = dy . W needs
 * W's columns K-contiguous: pass w = W^T, (in, out) -> rows of length `out`); accumulate != 0 adds the product onto
 * the bf16 values already in c (the residual gradient: `ds.addmm_(dh, W1)`).  256 x 256 tiles, LDS-DMA half-tile
 * ring, v_mfma_f32_16x16x32_bf16 (csrc/gemm_bf16.hip).  N % 8 == 0, K % 64 == 0, K >= 128, strides multiples of 8,
 * 16-byte aligned pointers (N <= 8192 with a bias: the strip is kept in LDS).  Persistent: one workgroup per CU walks the
 * tiles, the next tile's first operand pieces are requested before this tile's stores.
 * cwlt_gemm_bf16_tune(variant, trace): A/B and diagnostic switches (tools/bench_gemm.py); variant < 0: default.  Bit 0:
 * the next tile's first operands are requested after the main loop instead of from inside its last K-tile; bits 1-3:
 * start stagger of the workgroups in eighths of a tile period (default: 4 for K <= 1024, else 0); bits 4-7: timing
 * experiments with WRONG results on bias-free, non-accumulating launches (1 no DMA, 2 no fragment reads, 4 no barriers,
 * 8 no counted waits in the main loop); bits 8-15: at most that many x 8 workgroups.  trace != NULL (8192 uint32 of
 * device memory): bias-free, non-accumulating launches run the diagnostic build and leave workgroup 0's s_memtime stamps
 * of its second tile there. */
int cwlt_gemm_bf16(const void* a, const void* w, const float* bias, void* c, int64_t M, int N, int K, int64_t lda,
                   int64_t ldw, int64_t ldc, int accumulate, void* stream);
int cwlt_gemm_bf16_tune(int variant, void* trace);

/* ---- positional encoding + dropout --------------------------------------------------------------
 * y = dropout_p(x + pe[r % T]) -- PositionalEncoding.forward, dqn_policy/model.py:90-92.  pe is the
 * registered (max_len, D) f32 buffer; pe == NULL gives plain dropout, which is also this op's
 * backward (dx = dropout with the same seed applied to dy). */
int cwlt_posenc_dropout(const void* x, const float* pe, void* y, int64_t rows, int T, int D,
                        float p, uint64_t seed, const uint64_t* seed_base, int dtype, void* stream);

/* ---- compound-word multi-embedding --------------------------------------------------------------
 * out[r, off_f : off_f+width_f] = table_f[tokens[r, f]] * sqrt(width_f), f = 0..n_attr-1.
 * Replaces Embeddings.forward x6 + torch.cat (dqn_policy/model.py:67-74,206-221;
 * ppo_policy/model.py:69-76,208-223; dqn_policy/AIRL_model.py:34-41,108-115).
 * tables / widths / nrows are HOST arrays (device pointers to f32 tables; widths % 64 == 0).  out: 16-byte aligned,
 * row stride ldo a multiple of 8 elements (bf16) / 4 (f32). */
int cwlt_embed_splits(int64_t rows);
int cwlt_cw_embed_fwd(const int64_t* tokens, const void* const* tables, const int* widths,
                      const int* nrows, int n_attr, void* out, int64_t rows, int64_t ldo,
                      int dtype, void* stream);
/* dtables: one flat f32 buffer, table f's gradient at element offset sum_{g<f} nrows[g]*widths[g];
 * part: cwlt_embed_splits(rows) * (total table elements) f32.  Deterministic (no atomics). */
int cwlt_cw_embed_bwd(const int64_t* tokens, const int* widths, const int* nrows, int n_attr,
                      const void* dout, float* part, float* dtables, int64_t rows, int64_t ldd,
                      int dtype, void* stream);
/* The whole input front in one pass -- embedding x6 + cat + in_linear + PositionalEncoding
 * (dqn_policy/model.py:206-223 and 90-92; ppo_policy/model.py:208-225) -- from PROJECTED tables:
 *   out[r, :] = dropout(sum_f tproj[rowofs_f + tokens[r, f], :] + bias + pe[r % T, :]),  rowofs_f = sum_{g<f} nrows[g],
 *   tproj = cat_f(sqrt(width_f) * table_f . W_in[:, cols_f]^T)   (sum nrows, D), built by the caller per step.
 * The (rows, sum widths) concatenated embeddings and the in_linear GEMM over the token rows do not exist on this path.
 * tproj has out's dtype (bf16 / f32), dense; bias (D), pe (max_len >= T, D) f32 (pe may be NULL); out (rows, D) dense;
 * 16-byte aligned pointers; D % 64 == 0, D <= 2048; ids outside [0, nrows) are clamped; 0 <= p < 1.  The dropout
 * stream is cwlt_posenc_dropout's for the same seed. */
int cwlt_cw_embed_proj_fwd(const int64_t* tokens, const void* tproj, const int* nrows, int n_attr,
                           const float* bias, const float* pe, void* out, int64_t rows, int T, int D,
                           float p, uint64_t seed, const uint64_t* seed_base, int dtype, void* stream);
/* dtproj[rowofs_f + id, :] = sum of dpre[r, :] over the token rows with tokens[r, f] == id ((sum nrows, D) f32), dpre =
 * the gradient in front of the dropout (cwlt_posenc_dropout(dout, pe = NULL) with the forward's seed), row stride ldd.
 * The bias gradient is the column sum of any ONE attribute's rows of dtproj.
 * part: cwlt_embed_splits(rows) * (sum nrows) * D f32.  Deterministic (no atomics). */
int cwlt_cw_embed_proj_bwd(const int64_t* tokens, const int* nrows, int n_attr, int D, const void* dpre,
                           float* part, float* dtproj, int64_t rows, int64_t ldd, int dtype, void* stream);

/* ---- per-attribute softmax heads ----------------------------------------------------------------
 * logits (rows, ld), attribute f in columns [sum_{g<f} n_class[g], + n_class[f]) (n_class: HOST
 * array, each <= 256).  Replaces compute_loss x6 (dqn_policy/model.py:163-197: CE(reduction='none')
 * * loss_mask, summed) and Softmax + argmax x6 (dqn_policy/IRL_dqn_train.py:244-250,
 * ppo_policy/ppo_train.py:259-267).  Outputs (each may be NULL): loss_sum (n_attr) =
 * sum_r mask_r * nll_{r,f} [caller divides by sum(mask)]; argmax (rows, n_attr) int64 = first index
 * of the largest softmax value; pmax (rows, n_attr) its probability; probs (rows, ldp) the softmax.
 * loss_part: cwlt_heads_blocks(rows) * n_attr f32. */
int cwlt_heads_blocks(int64_t rows);
int cwlt_heads_fwd(const void* logits, const int* n_class, int n_attr, const int64_t* target,
                   const float* mask, float* loss_part, float* loss_sum, int64_t* argmax,
                   float* pmax, float* probs, int64_t rows, int64_t ld, int64_t ldp,
                   int dtype, void* stream);
/* dlogits = (softmax - onehot(target)) * mask_r * coef[f]; coef (n_attr) f32 on device. */
int cwlt_heads_ce_bwd(const void* logits, const int* n_class, int n_attr, const int64_t* target,
                      const float* mask, const float* coef, void* dlogits, int64_t rows, int64_t ld,
                      int dtype, void* stream);

/* ---- sliding-window attention of the AIRL discriminator --------------------------------
 * Replaces HF LongformerSelfAttention.forward (banded scores -> fp32 softmax -> probs @ value) as
 * reached from dqn_policy/AIRL_model.py:78-90,117 (attention_window 50) and ppo_policy/model.py:440-451,470
 * (attention_window 512).  q, k, v, out: (B, L, H, 64), row strides ld*; mask (B, L) f32, nonzero =
 * attend (NULL = all); window = one-sided width (attention_window / 2); scale = 1/sqrt(64) applied to q;
 * masked queries produce zero rows; p = dropout on the probabilities. */
int cwlt_band_attn_fwd(const void* q, const void* k, const void* v, const float* mask, void* out, float* lse,
                       int B, int H, int L, int head_dim, int window,
                       int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                       float scale, float p, uint64_t seed, const uint64_t* seed_base, int dtype, void* stream);

/* Backward of cwlt_band_attn_fwd -- what torch autograd derives through HF LongformerSelfAttention when the
 * reference trains the discriminator (dqn_policy/AIRL.py:135-170, global_loss.backward()).  lse (B, H, L) f32 =
 * the forward's optional log-sum-exp output (+inf on zeroed query rows; pass NULL to the forward when no
 * backward follows); out = the forward's output; dout its gradient (row stride lddo).  dq, dk, dv: gradients
 * w.r.t. the raw q, k, v, same (B, L, H, 64) layout with row strides lddq / lddk / lddv.  The dropout mask is
 * regenerated from (seed, b, h, i, j). */
int cwlt_band_attn_bwd(const void* q, const void* k, const void* v, const float* mask, const void* out,
                       const float* lse, const void* dout, void* dq, void* dk, void* dv,
                       int B, int H, int L, int head_dim, int window,
                       int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo,
                       int64_t lddq, int64_t lddk, int64_t lddv,
                       float scale, float p, uint64_t seed, const uint64_t* seed_base, int dtype, void* stream);

/* Gradient of sum_{r,f} w[r][f] * log softmax_f(logits_r)[target[r][f]] w.r.t. the logits, given as the kernel
 * form dlogits = (softmax - onehot(target)) * w: pass w = -(upstream gradient).  Used for the PPO log pi(a)
 * of the greedy action (ppo_policy/ppo_train.py:320-336).  w (rows, n_attr) f32 on device. */
int cwlt_heads_logp_bwd(const void* logits, const int* n_class, int n_attr, const int64_t* target,
                        const float* w, void* dlogits, int64_t rows, int64_t ld, int dtype, void* stream);

/* ---- RL arithmetic (one launch each; the reference's indexing quirks reproduced) -----------------
 * cwlt_rollout_gather: greedy action rows (+ log-probs) from per-position argmax ids (R, T, A) int64 and
 *   softmax probs (R, T, ldp) f32.  mode 0 = DQN.choose_action (dqn_policy/IRL_dqn_train.py:256-264:
 *   positions [0, T-1, T-2, ...] because -0 == 0); mode 1 = PPO.choose_action (ppo_policy/ppo_train.py:
 *   269-290: rows at -1..-NA, tempo/chord log-prob class taken at position +idx); mode 2 = select_udpate
 *   (ppo_train.py:312-336, no quirk).  action (R, NA, A) int64, logp (R, NA, A) f32 (modes 1, 2). */
int cwlt_rollout_gather(const int64_t* ids, const float* probs, const int* n_class, int n_attr,
                        int64_t* action, float* logp, int R, int T, int NA, int64_t ldp, int mode,
                        void* stream);
/* PPO.calculate_returns + calculate_advantages (ppo_train.py:348-363): forward-order discounted sums,
 * (x - mean) / unbiased std, adv = returns - values, normalised.  E in [2, 8192]. */
int cwlt_ppo_returns_adv(const float* rewards, const float* values, float* returns, float* adv, int E,
                         float gamma, int normalize, void* stream);
/* PPO surrogate (ppo_train.py:388-396): L = -mean(min(0.2*A_e, clamp(exp(new - old), 1-clip, 1+clip)*A_e));
 * new_logp (NA, A) f32, old_logp (E, NA, A) int64 (stored log-probs truncated to integers), adv (E) f32.
 * Outputs loss (1) and grad (NA, A) = dL/dnew_logp. */
int cwlt_ppo_policy_loss(const float* new_logp, const int64_t* old_logp, const float* adv, float* loss,
                         float* grad, int E, int NA, int n_attr, float clip, void* stream);
/* DQN TD loss (dqn_policy/IRL_dqn_train.py:285-330): q gather from batch element 0 (index shape (1, B, NA)),
 * target = reward + gamma (1 - done) topk_NA(max_c target_logits), per-attribute MSE.  y, yt (B, T, ld) f32.
 * mse_part (B, A): sum_k squared error; dq (B, NA, A): d MSEloss / d qval. */
int cwlt_dqn_td_fwd(const float* y, const float* yt, const int* n_class, int n_attr, const int64_t* action,
                    const float* reward, const float* done, float* mse_part, float* dq, int B, int T,
                    int NA, int64_t ld, float gamma, void* stream);
/* scatter dq into the gradient rows of batch element 0 of y (dy zero-filled by the caller), times *gout */
int cwlt_dqn_td_bwd(const float* dq, const int64_t* action, const int* n_class, int n_attr, float* dy,
                    const float* gout, int B, int NA, int64_t ld, void* stream);

/* ---- weight-gradient GEMM of the dense projections ------------------------------------------------
 * out (N1, N2) f32 (+)= A^T B with A = upstream gradient (M, N1), B = layer input (M, N2), bf16 row-major,
 * M = B*T token rows (the reduction).  Replaces the weight-gradient GEMMs autograd runs for the nn.Linear
 * layers of the encoder (dqn_policy/model.py:128-137; FT AttentionLayer / TransformerEncoderLayer).
 * N1, N2 multiples of 256; part: cwlt_wgrad_splits(M, N1, N2) * N1 * N2 f32.  Deterministic. */
int cwlt_wgrad_splits(int64_t M, int N1, int N2);
int cwlt_wgrad_bf16(const void* a, const void* b, float* part, float* out, int64_t M, int N1, int N2,
                    int64_t lda, int64_t ldb, int accumulate, void* stream);

/* Up to four such products over the SAME M token rows in one launch + one reduce launch: the four weight gradients of an
 * encoder layer at few token rows run 20-80 workgroups each, 240 together.  All arrays are HOST arrays of `count` (1..4)
 * entries; widths multiples of 256; part[p]: cwlt_wgrad_splits(M, n1[p], n2[p]) * n1[p] * n2[p] floats, distinct buffers.
 * Results bit-identical to count calls of cwlt_wgrad_bf16 (every product cuts the rows as its own launch would). */
int cwlt_wgrad_bf16_group(const void* const* a, const void* const* b, float* const* part, float* const* out, const int* n1,
                          const int* n2, const int64_t* lda, const int64_t* ldb, int count, int64_t M, int accumulate,
                          void* stream);

/* ---- recurrent (generation) form of the causal linear attention ------------------------------------
 * One token: Zi += phi(k); Si += phi(k) (x) v; out = (phi(q) . Si) / (phi(q) . Zi + eps), state updated in
 * place.  Replaces fast_transformers RecurrentLinearAttention.forward as built by RecurrentEncoderBuilder at
 * dqn_policy/model.py:141-150 and driven by dqn_policy/testing-no-type-cp.py:126-179.
 * q, k, v, out: (N, H*64) rows (row strides ld*); S (N, H, 64, 64) f32; Z (N, H, 64) f32. */
int cwlt_recurrent_cla_step(const void* q, const void* k, const void* v, float* S, float* Z, void* out,
                            int N, int H, int head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                            float eps, int dtype, void* stream);

/* ---- generation: one CW token through the whole recurrent-form model (f32) --------------------------
 * Replaces the per-token body of the reference's generation loop: `forward_hidden(input_, memory,
 * is_training=False)` (dqn_policy/testing-no-type-cp.py:150,166 -> dqn_policy/model.py:200-238 -> the
 * RecurrentEncoderBuilder product of dqn_policy/model.py:141-150) followed by the six head projections of
 * `forward_output_sampling` (dqn_policy/model.py:273-278); likewise ppo_policy/inference.py:78-160 over
 * ppo_policy/model.py:200-262.  All pointers are device pointers unless marked HOST; the structs themselves
 * are HOST memory, read during the call only.  Per-song state S / Z is updated in place. */
typedef struct cwlt_decode_layer {
    const float *wqkv, *bqkv;     /* (3D, D), (3D): query | key | value projection rows stacked */
    const float *wo, *bo;         /* (D, D), (D): out_projection */
    const float *ln1_w, *ln1_b;   /* norm1 */
    const float *w1, *b1;         /* (F, D), (F): linear1 */
    const float *w2, *b2;         /* (D, F), (D): linear2 */
    const float *ln2_w, *ln2_b;   /* norm2 */
    float *S, *Z;                 /* (n_songs, H, 64, 64), (n_songs, H, 64) f32 recurrent state */
} cwlt_decode_layer;

typedef struct cwlt_decode_model {
    int n_layer, n_head, d_model, d_ff, n_attr, emb_width, n_logits;
    float eps_ln, eps_attn;            /* 1e-5, 1e-6 in the reference */
    const void* const* tables;         /* HOST array of n_attr device pointers (f32 embedding tables) */
    const int* widths;                 /* HOST (n_attr): embedding widths, sum = emb_width */
    const int* nrows;                  /* HOST (n_attr): vocabulary sizes */
    const float *w_in, *b_in;          /* (D, emb_width), (D): in_linear */
    const float* pe0;                  /* (D): row 0 of the positional-encoding buffer (pos_emb.pe) */
    const cwlt_decode_layer* layers;   /* HOST array of n_layer */
    const float *lnf_w, *lnf_b;        /* final encoder norm (NULL: none) */
    const float *w_heads, *b_heads;    /* (n_logits, D), (n_logits): the proj_* heads stacked */
} cwlt_decode_model;

/* floats of workspace PER SONG (-1: unsupported model: needs d_model = 64*n_head <= 2048, d_ff <= 2048,
 * emb_width <= 2048, all multiples of 4) */
int64_t cwlt_decode_workspace_floats(const cwlt_decode_model* m);
/* tokens: (n_songs, n_attr) int64; work: n_songs * cwlt_decode_workspace_floats(m) f32 (16-byte aligned);
 * hidden: (n_songs, D) f32 or NULL -- what forward_hidden returns; logits: (n_songs, n_logits) f32. */
int cwlt_decode_step(const cwlt_decode_model* m, const int64_t* tokens, float* work, float* hidden,
                     float* logits, int n_songs, void* stream);
/* The step's building block: out[n, r] = epi(W[r, :] . pro(xin[n, :]) + bias[r]); pro = optional LayerNorm
 * (ln_w, ln_b) and an optional second one (ln2_w, ln2_b), the normalised vector also stored to x_out when given;
 * epi = exact-erf GELU when act == 1, then + res[n, r] when res != NULL.  K % 4 == 0, K <= 2048. */
int cwlt_decode_gemv(const float* W, const float* bias, const float* xin, const float* ln_w,
                     const float* ln_b, const float* ln2_w, const float* ln2_b, float eps,
                     const float* res, float* out, float* x_out, int n_out, int K, int act, int n_songs,
                     int64_t ld_x, int64_t ld_res, int64_t ld_out, int64_t ld_xo, void* stream);

/* Device-side sampling of the next CW token: for each of `rows` songs and each attribute a, draw from
 * Categorical(softmax(logits[row, off_a : off_a + n_class[a]] / temperature[a])) and store the class id in
 * tokens[row, a] (and in song[counter, row, a] when song != NULL and *counter < song_rows).  Replaces the six
 * `Categorical(softmax(y)).sample()` draws + torch.cat of ppo_policy/inference.py:115-141 (same distribution; the
 * generator is counter-based, keyed by (seed, *counter, row, a), so a captured graph draws fresh numbers per replay
 * as long as the caller advances *counter).  top_p[a] < 1 restricts attribute a to its nucleus first -- the
 * smallest set of most-probable classes whose mass exceeds top_p[a], renormalised: `nucleus` of
 * dqn_policy/model.py:33-47, with temperature[a] its `softmax_with_temperature` (:19-21).
 * n_class / temperature / top_p: HOST arrays (NULL = 1.0); n_class[a] <= 256, n_attr <= 8; counter: device int64
 * (NULL = 0). */
int cwlt_sample_categorical(const float* logits, const int* n_class, const float* temperature,
                            const float* top_p, int n_attr, int64_t rows, int64_t ld, uint64_t seed,
                            const int64_t* counter, int64_t* tokens, int64_t* song, int64_t song_rows,
                            void* stream);

/* ---- the dense projections at few token rows, and a whole encoder layer per host call ---------------------------------
 * The reference's own RL setting is 30 windows x 50 tokens = 1 500 token rows per network pass
 * (dqn_policy/IRL_dqn_train.py:267-345 `DQN.update`, ppo_policy/ppo_train.py:365-417 `update_policy`): ~970 launches of
 * ~10 us per update, bound by the HOST when every launch is its own Python-level call.
 *
 * cwlt_gemm_bf16_small: cwlt_gemm_bf16's product on 64 x 64 output tiles (192-768 workgroups at 1 500 rows where 256-row
 * tiles leave 12), operands straight from L2 into MFMA fragments.  N % 8 == 0, K % 32 == 0, row strides multiples of 8,
 * 16-byte aligned pointers; same arithmetic (f32 accumulation, bias in f32, one rounding).
 * cwlt_transpose_bf16_many: dst_i (cols, rows) = src_i (rows, cols)^T for the n dense bf16 matrices of `table` (DEVICE
 * memory, n x 4 int64: element offset from src, element offset from dst, rows, cols) in one launch -- the transposed
 * weight copies the input-gradient products read. */
int cwlt_gemm_bf16_small(const void* a, const void* w, const float* bias, void* c, int64_t M, int N, int K, int64_t lda,
                         int64_t ldw, int64_t ldc, int accumulate, void* stream);
int cwlt_transpose_bf16_many(const void* src, void* dst, const int64_t* table, int n, void* stream);
/* dst_i = bf16(src_i) for the n arrays of `table` (DEVICE memory, n x 3 int64: element offset from src (f32), element
 * offset from dst (bf16), element count) in one launch: the bf16 copies of an encoder's master weights, refreshed once per
 * forward (torch's multi-tensor copy takes four launches and half the bandwidth for the same 84 tensors). */
int cwlt_cast_bf16_many(const float* src, void* dst, const int64_t* table, int n, void* stream);
/* cwlt_gemm_nt_bias_gelu_dropout on the split-K small tiles, for passes of a few hundred rows at most (a 50-token rollout
 * step): g = dropout_p(gelu(bf16(a . w^T) + bias)), and gd = mask / (1 - p) * gelu'(.) when gd != NULL -- what
 * cwlt_gemm_bf16_small followed by cwlt_bias_gelu_dropout_fwd produce, bit for bit (same rounding of the product, same
 * dropout stream keyed by (seed, row * N + column)).  g, gd dense (M, N); K % 128 == 0, N % 8 == 0. */
int cwlt_gemm_bf16_small_gelu(const void* a, const void* w, const float* bias, void* g, void* gd, int64_t M, int N, int K,
                              int64_t lda, int64_t ldw, float p, uint64_t seed, const uint64_t* seed_base, void* stream);

/* One post-LN encoder layer -- fast_transformers' TransformerEncoderLayer(AttentionLayer(CausalLinearAttention)) as
 * built at dqn_policy/model.py:128-137 (ppo_policy/model.py:129-138) and called at :232:
 *     x1 = norm1(x + dropout(out_projection(causal_linear_attention(q, k, v))));
 *     y  = norm2(x1 + dropout(linear2(dropout(gelu(linear1(x1))))))
 * -- enqueued by ONE host call forward (8 launches) and one backward (13-21 launches: 13 with the four weight gradients as one grouped launch), through the entry points above
 * (same kernels, same dropout streams as the per-op path).  bf16 activations; d_model 512, 8 heads of 64, d_ff % 256 == 0.
 * The struct is HOST memory, read during the call only; every pointer in it is a device pointer.
 * forward : reads x and the parameters; writes y and `saved`; uses `scratch` (fwd_scratch_bytes).
 * backward: reads dy, x, `saved` (of the matching forward, same n_seq/len/p_drop/seeds/want_backward = 1), the
 *           TRANSPOSED weight copies and gamma1/gamma2; writes dx and `grads`; uses `scratch` (bwd_scratch_bytes).
 * All buffers 256-byte aligned.  grads (f32, grad_floats): the 12 parameter gradients at grad_off[] (float offsets),
 * in the order CWLT_LAYER_GRAD_* below; the Q / K / V weight (bias) gradients are the three row blocks of wqkv (bqkv). */
#define CWLT_LAYER_NGRADS 12
#define CWLT_LAYER_NSAVED 13
enum { CWLT_LAYER_GRAD_WQKV = 0, CWLT_LAYER_GRAD_BQKV, CWLT_LAYER_GRAD_WO, CWLT_LAYER_GRAD_BO, CWLT_LAYER_GRAD_W1,
       CWLT_LAYER_GRAD_B1, CWLT_LAYER_GRAD_W2, CWLT_LAYER_GRAD_B2, CWLT_LAYER_GRAD_GAMMA1, CWLT_LAYER_GRAD_BETA1,
       CWLT_LAYER_GRAD_GAMMA2, CWLT_LAYER_GRAD_BETA2 };
typedef struct cwlt_encoder_layer {
    int64_t n_seq, len;                /* N sequences x L tokens: R = N L token rows */
    int32_t d_model, d_ff, n_heads;
    int32_t want_backward;             /* forward: also leave the scan's final state for a one-sweep backward */
    float p_drop, ln_eps, attn_eps;    /* dropout probability (0 in eval mode); 1e-5, 1e-6 in the reference */
    int32_t reserved;
    uint64_t seed[3];                  /* dropout streams: residual block 1, FFN, residual block 2 */
    const uint64_t* seed_base;         /* optional device uint64 added to every seed (hipGraph replays), or NULL */
    const void *wqkv, *wo, *w1, *w2;   /* bf16 (3D, D) query|key|value rows stacked, (D, D), (F, D), (D, F) as stored */
    const float *bqkv, *bo, *b1, *b2, *gamma1, *beta1, *gamma2, *beta2;     /* f32 */
    const void *wqkv_t, *wo_t, *w1_t, *w2_t;   /* backward: bf16 transposed copies (D, 3D), (D, D), (D, F), (F, D) */
    const void* x;                     /* (R, D) bf16 layer input */
    void* y;                           /* (R, D) bf16 layer output (forward) */
    void* saved;                       /* saved_bytes: activations kept for the backward */
    void* scratch;
    const void* dy;                    /* (R, D) bf16 gradient of y (backward) */
    void* dx;                          /* (R, D) bf16 gradient of x (backward) */
    float* grads;                      /* grad_floats f32 (backward) */
} cwlt_encoder_layer;
typedef struct cwlt_encoder_layer_plan_t {
    int64_t saved_bytes, fwd_scratch_bytes, bwd_scratch_bytes, grad_floats;
    int64_t grad_off[CWLT_LAYER_NGRADS];
    int64_t saved_off[CWLT_LAYER_NSAVED];  /* byte offsets inside `saved` (tests): qkv, attention out, zinv, final state,
                                            * s1, x1, mean1, rstd1, gd, g, s2, mean2, rstd2 */
} cwlt_encoder_layer_plan_t;
int cwlt_encoder_layer_plan(int64_t n_seq, int64_t len, int d_model, int d_ff, int n_heads, float p_drop,
                            int want_backward, cwlt_encoder_layer_plan_t* plan);
int cwlt_encoder_layer_fwd(const cwlt_encoder_layer* layer, void* stream);
int cwlt_encoder_layer_bwd(const cwlt_encoder_layer* layer, void* stream);
/* The whole stack -- fast_transformers' TransformerEncoder.forward loop over its layers (dqn_policy/model.py:232) -- in one
 * host call: `layers` is a HOST array of n structs wired by the caller (layer i's y is layer i + 1's x; in the backward
 * layer i's dx is layer i - 1's dy); forward runs them in order, backward in reverse. */
int cwlt_encoder_fwd(const cwlt_encoder_layer* layers, int n, void* stream);
int cwlt_encoder_bwd(const cwlt_encoder_layer* layers, int n, void* stream);

/* ---- hipGraph hygiene (no counterpart in the reference, which has no graphs; used by the drop-in RL loops' replayed
 * steps) -------------------------------------------------------------------------------------------
 * Replace every MEMSET node of a captured, not yet instantiated hipGraph_t by a kernel node writing the same bytes (same
 * dependencies, same dependents): on ROCm 7.2 a captured hipMemsetAsync replays a wrong fill pattern from the second
 * replay on, and library code inside a captured step (PyTorch's reduction semaphores, hipBLASLt's split-K workspaces)
 * issues such memsets.  *replaced (may be NULL): nodes replaced.  Child graphs are not entered.  0 or a hipError_t. */
int cwlt_graph_replace_memset_nodes(void* graph, int* replaced);

#ifdef __cplusplus
}
#endif
#endif /* CWLT_H */
