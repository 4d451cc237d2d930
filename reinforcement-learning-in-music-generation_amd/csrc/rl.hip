// RL arithmetic around the CW transformer: greedy action / log-prob gather, PPO returns + advantages,
// PPO ratio-clip surrogate (fwd + grad), DQN target-Q / TD loss (fwd + grad) -- each ONE launch instead
// of the reference's Python loops of tiny tensor ops (25 torch.cat per action, 750 hstack/cat per PPO
// inner step, ~40 ops per DQN update).  The reference's indexing quirks are reproduced exactly
// (SURVEY §8a "quirks" column); every kernel cites the lines it follows.  Sizes are tiny (windows of
// 50..1024 tokens, 25..512 actions): these kernels exist to remove launch overhead, not to move bytes.
#include "cwlt_common.h"

#define CWLT_MAX_ATTR 8

namespace cwlt {

struct RLHeads {
    int n[CWLT_MAX_ATTR];
    int off[CWLT_MAX_ATTR];
    int n_attr;
};

// ---------------------------------------------------------------------------------------------
// Greedy action rows (+ log-probs) from per-position argmax ids and softmax probabilities.
//   mode 0 (DQN, IRL_dqn_train.py:256-264): action[k] = ids[pos(-k)], k = 0..NA-1, where -0 == 0
//          -> positions [0, T-1, T-2, ...]
//   mode 1 (PPO, ppo_train.py:269-290):     action[k] = ids[T-(k+1)], k = 0..NA-1;
//          logp[k][f] = log probs[T-(k+1)][f][c], c = ids[T-(k+1)][f] for f >= 2, but for tempo and
//          chord (f = 0, 1) c = ids[+(k+1)][f] -- the class chosen at position +idx (reference quirk)
//   mode 2 (select_udpate, ppo_train.py:312-336): as mode 1 without the quirk (c = own argmax for all f)
// ids (R, T, A) int64; probs (R, T, ldp) f32; action (R, NA, A) int64; logp (R, NA, A) f32 (modes 1, 2)
// ---------------------------------------------------------------------------------------------
__global__ void rollout_gather_kernel(const int64_t* __restrict__ ids, const float* __restrict__ probs, RLHeads hd,
                                      int64_t* __restrict__ action, float* __restrict__ logp, int R, int T, int NA,
                                      long ldp, int mode) {
    const long total = (long)R * NA * hd.n_attr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i % hd.n_attr);
        const int k = (int)((i / hd.n_attr) % NA);
        const int r = (int)(i / ((long)hd.n_attr * NA));
        int pos;
        if (mode == 0) pos = k == 0 ? 0 : T - k;
        else pos = T - (k + 1);
        const int64_t a = ids[((long)r * T + pos) * hd.n_attr + f];
        action[i] = a;
        if (mode != 0 && logp) {
            int64_t c = a;
            if (mode == 1 && f < 2) c = ids[((long)r * T + (k + 1)) * hd.n_attr + f];
            logp[i] = logf(probs[((long)r * T + pos) * ldp + hd.off[f] + c]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// PPO returns + advantages (ppo_train.py:348-363).  One wave.
//   R = 0; for r in rewards (FORWARD order): R = r + R*gamma; returns.insert(0, R)
//   => returns[i] = sum_{t <= E-1-i} gamma^(E-1-i-t) r_t     (not the textbook recursion)
//   returns = (returns - mean) / std (unbiased);  adv = returns - values;  adv = (adv - mean) / std
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ppo_returns_adv_kernel(const float* __restrict__ rewards,
                                                             const float* __restrict__ values,
                                                             float* __restrict__ returns, float* __restrict__ adv,
                                                             int E, float gamma, int normalize) {
    extern __shared__ float buf[];   // E floats
    const int lane = threadIdx.x;
    if (lane == 0) {
        float Rr = 0.f;
        for (int t = 0; t < E; ++t) {
            Rr = rewards[t] + Rr * gamma;
            buf[E - 1 - t] = Rr;
        }
    }
    __syncthreads();
    // mean / unbiased std over E elements (E may exceed 64: strided)
    float s = 0.f;
    for (int i = lane; i < E; i += 64) s += buf[i];
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
    for (int i = lane; i < E; i += 64) q += (buf[i] - mean) * (buf[i] - mean);
    const float sd = sqrtf(wave_sum(q) / (float)(E - 1));
    float s2 = 0.f;
    for (int i = lane; i < E; i += 64) {
        const float rn = normalize ? (buf[i] - mean) / sd : buf[i];
        returns[i] = rn;
        const float a = rn - values[i];
        buf[i] = a;
        s2 += a;
    }
    __syncthreads();
    const float am = wave_sum(s2) / (float)E;
    float q2 = 0.f;
    for (int i = lane; i < E; i += 64) q2 += (buf[i] - am) * (buf[i] - am);
    const float asd = sqrtf(wave_sum(q2) / (float)(E - 1));
    for (int i = lane; i < E; i += 64) adv[i] = normalize ? (buf[i] - am) / asd : buf[i];
}

// ---------------------------------------------------------------------------------------------
// PPO surrogate (ppo_train.py:388-396):
//   ratio[e][k][f] = exp(new_logp[k][f] - old[e][k][f])        old = stored log-probs TRUNCATED to int64
//   L = -mean_{e,k,f} min(0.2 * A_e, clamp(ratio, 1-clip, 1+clip) * A_e)      (surrogate 1 is 0.2*A, not ratio*A)
// Emits the loss and dL/dnew_logp[k][f] (sum over e), fixed-order reduction (one workgroup).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ppo_policy_loss_kernel(const float* __restrict__ new_logp,
                                                              const int64_t* __restrict__ old_logp,
                                                              const float* __restrict__ adv, float* __restrict__ loss,
                                                              float* __restrict__ grad, int E, int KF, float clip) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const float invn = 1.0f / ((float)E * (float)KF);
    float lsum = 0.f;
    for (int kf = tid; kf < KF; kf += 256) {
        const float nl = new_logp[kf];
        float g = 0.f;
        for (int e = 0; e < E; ++e) {
            const float a = adv[e];
            const float ratio = expf(nl - (float)old_logp[(long)e * KF + kf]);
            const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
            const float l1 = 0.2f * a, l2 = rc * a;
            lsum += fminf(l1, l2);
            // d min / d new_logp: only through l2, only when it is the smaller term and the ratio is unclamped
            if (l2 < l1 && ratio > 1.0f - clip && ratio < 1.0f + clip) g += a * ratio;
        }
        grad[kf] = -g * invn;
    }
    red[tid] = lsum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) loss[0] = -red[0] * invn;
}

// ---------------------------------------------------------------------------------------------
// DQN TD loss (IRL_dqn_train.py:285-330).  y = eval-net logits (B, T, ld), yt = target-net logits.
//   qval_f[j][k]  = y_f[0, j, action[j][k][f]]          gather index has shape (1, B, NA): batch element 0,
//                                                       sequence position j = 0..B-1 (quirk; needs B <= T)
//   next_f[j][t]  = max_c yt_f[j, t, c];  top_f[j][:] = topk(next_f[j], NA) (descending)
//   target_f[j][k] = reward[j] + gamma * (1 - done[j]) * top_f[j][k]
//   mse_f = mean_{j,k} (qval - target)^2 ;  MSEloss = sum_f mse_f / A
// One wave per (j, f).  Emits mse_part[j][f] (sum over k of squared error) and, for the backward,
// dq[j][k][f] = 2 (qval - target) / (B * NA * A); the gradient w.r.t. y is scattered by a second kernel
// into rows of batch element 0 only.  T <= 64 * 16.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dqn_td_kernel(const float* __restrict__ y, const float* __restrict__ yt,
                                                    RLHeads hd, const int64_t* __restrict__ action,
                                                    const float* __restrict__ reward, const float* __restrict__ done,
                                                    float* __restrict__ mse_part, float* __restrict__ dq, int B, int T,
                                                    int NA, long ld, float gamma) {
    extern __shared__ float nx[];   // T floats: next-state max-Q per position, then sorted descending
    const int lane = threadIdx.x;
    const int j = blockIdx.x, f = blockIdx.y;
    const int n = hd.n[f], off = hd.off[f];
    int npad = 64;
    while (npad < T) npad <<= 1;
    for (int t = lane; t < npad; t += 64) {
        float m = -INFINITY;
        if (t < T) {
            const float* row = yt + ((long)j * T + t) * ld + off;
            for (int c = 0; c < n; ++c) m = fmaxf(m, row[c]);
        }
        nx[t] = m;
    }
    __syncthreads();
    // bitonic sort, descending, npad elements in LDS
    for (int size = 2; size <= npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = lane; i < npad / 2; i += 64) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const float a = nx[lo], b2 = nx[hi];
                if ((a < b2) == desc) { nx[lo] = b2; nx[hi] = a; }
            }
            __syncthreads();
        }
    }
    const float coef = gamma * (1.0f - done[j]);
    const float gscale = 2.0f / ((float)B * (float)NA * (float)hd.n_attr);
    float se = 0.f;
    for (int k = lane; k < NA; k += 64) {
        const int64_t a = action[((long)j * NA + k) * hd.n_attr + f];
        const float qv = y[((long)0 * T + j) * ld + off + a];          // batch element 0, position j
        const float tg = reward[j] + coef * nx[k];
        const float d = qv - tg;
        se += d * d;
        dq[((long)j * NA + k) * hd.n_attr + f] = d * gscale;
    }
    se = wave_sum(se);
    if (lane == 0) mse_part[j * hd.n_attr + f] = se;
}

// dy[0, j, off_f + action[j][k][f]] += dq[j][k][f] * gout   (one thread per (j, f): serial over k, no atomics)
__global__ void dqn_td_scatter_kernel(const float* __restrict__ dq, const int64_t* __restrict__ action, RLHeads hd,
                                      float* __restrict__ dy, int B, int NA, long ld, const float* __restrict__ gout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * hd.n_attr) return;
    const int j = i / hd.n_attr, f = i % hd.n_attr;
    const float g = gout[0];
    float* row = dy + (long)j * ld + hd.off[f];
    for (int k = 0; k < NA; ++k) {
        const long idx = ((long)j * NA + k) * hd.n_attr + f;
        row[action[idx]] += dq[idx] * g;
    }
}

static int fill_rl(RLHeads& a, const int* n_class, int n_attr) {
    if (!n_class || n_attr <= 0 || n_attr > CWLT_MAX_ATTR) return CWLT_ERR_ARG;
    int off = 0;
    for (int f = 0; f < n_attr; ++f) {
        if (n_class[f] <= 0) return CWLT_ERR_ARG;
        a.n[f] = n_class[f];
        a.off[f] = off;
        off += n_class[f];
    }
    for (int f = n_attr; f < CWLT_MAX_ATTR; ++f) a.n[f] = a.off[f] = 0;
    a.n_attr = n_attr;
    return CWLT_OK;
}

}  // namespace cwlt

extern "C" {

int cwlt_rollout_gather(const int64_t* ids, const float* probs, const int* n_class, int n_attr, int64_t* action,
                        float* logp, int R, int T, int NA, int64_t ldp, int mode, void* stream) {
    using namespace cwlt;
    RLHeads hd;
    int e = fill_rl(hd, n_class, n_attr);
    if (e) return e;
    if (!ids || !action || R < 0 || T <= 0 || NA <= 0 || mode < 0 || mode > 2) return CWLT_ERR_ARG;
    if (mode == 0 ? NA > T : NA + 1 > T) return CWLT_ERR_ARG;            // positions must exist
    if (mode != 0 && (!probs || !logp)) return CWLT_ERR_ARG;
    if (R == 0) return CWLT_OK;
    const long total = (long)R * NA * n_attr;
    long nb = (total + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(rollout_gather_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, ids, probs, hd, action, logp,
                       R, T, NA, (long)ldp, mode);
    return (int)hipGetLastError();
}

int cwlt_ppo_returns_adv(const float* rewards, const float* values, float* returns, float* adv, int E, float gamma,
                         int normalize, void* stream) {
    using namespace cwlt;
    if (!rewards || !values || !returns || !adv || E < 2 || E > 8192) return CWLT_ERR_ARG;
    hipLaunchKernelGGL(ppo_returns_adv_kernel, dim3(1), dim3(64), sizeof(float) * E, (hipStream_t)stream, rewards,
                       values, returns, adv, E, gamma, normalize);
    return (int)hipGetLastError();
}

int cwlt_ppo_policy_loss(const float* new_logp, const int64_t* old_logp, const float* adv, float* loss, float* grad,
                         int E, int NA, int n_attr, float clip, void* stream) {
    using namespace cwlt;
    if (!new_logp || !old_logp || !adv || !loss || !grad || E <= 0 || NA <= 0 || n_attr <= 0) return CWLT_ERR_ARG;
    hipLaunchKernelGGL(ppo_policy_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, new_logp, old_logp, adv,
                       loss, grad, E, NA * n_attr, clip);
    return (int)hipGetLastError();
}

/* y, yt: (B, T, ld) f32 fused logits of the eval / target nets; action (B, NA, A) int64; reward, done (B) f32.
 * mse_part (B, A) f32: sum_k (q - target)^2 per (j, f) [caller: mse_f = sum_j part / (B*NA)];
 * dq (B, NA, A) f32: d MSEloss / d qval. */
int cwlt_dqn_td_fwd(const float* y, const float* yt, const int* n_class, int n_attr, const int64_t* action,
                    const float* reward, const float* done, float* mse_part, float* dq, int B, int T, int NA,
                    int64_t ld, float gamma, void* stream) {
    using namespace cwlt;
    RLHeads hd;
    int e = fill_rl(hd, n_class, n_attr);
    if (e) return e;
    if (!y || !yt || !action || !reward || !done || !mse_part || !dq) return CWLT_ERR_ARG;
    if (B <= 0 || T <= 0 || NA <= 0 || NA > T || B > T || T > 8192) return CWLT_ERR_ARG;   // B <= T: the gather quirk
    int npad = 64;
    while (npad < T) npad <<= 1;
    hipLaunchKernelGGL(dqn_td_kernel, dim3(B, n_attr), dim3(64), sizeof(float) * npad, (hipStream_t)stream, y, yt, hd,
                       action, reward, done, mse_part, dq, B, T, NA, (long)ld, gamma);
    return (int)hipGetLastError();
}

/* dy: (T, ld) f32 gradient rows of batch element 0 of y (caller zero-fills the whole (B, T, ld) gradient);
 * gout: device scalar = upstream gradient of MSEloss. */
int cwlt_dqn_td_bwd(const float* dq, const int64_t* action, const int* n_class, int n_attr, float* dy,
                    const float* gout, int B, int NA, int64_t ld, void* stream) {
    using namespace cwlt;
    RLHeads hd;
    int e = fill_rl(hd, n_class, n_attr);
    if (e) return e;
    if (!dq || !action || !dy || !gout || B <= 0 || NA <= 0) return CWLT_ERR_ARG;
    const int total = B * n_attr;
    hipLaunchKernelGGL(dqn_td_scatter_kernel, dim3((total + 63) / 64), dim3(64), 0, (hipStream_t)stream, dq, action, hd,
                       dy, B, NA, (long)ld, gout);
    return (int)hipGetLastError();
}

}  // extern "C"
