// Per-attribute softmax heads over the fused logits row: cross-entropy loss (fwd + bwd), greedy
// argmax and softmax probabilities, all A attributes of a CW token in one pass.
//
// Replaces, from /root/reference/dqn_policy/model.py:163-197 (ppo_policy/model.py:167-199):
//     6 x [ permute(0,2,1) -> CrossEntropyLoss(reduction='none') -> * loss_mask -> sum / sum(mask) ]
// and from dqn_policy/IRL_dqn_train.py:244-250, ppo_policy/ppo_train.py:259-267,300-307:
//     6 x [ Softmax(dim=-1) -> argmax(dim=-1) ]
// The six skinny logits tensors are one (rows, ld) matrix (one 512 x sum(n_f) GEMM); attribute f
// owns columns [off_f, off_f + n_f).  One wave per token row; segment max / sum by wave shuffles;
// per-block loss partials are summed in fixed order (deterministic).  HBM-bound.
#include "cwlt_common.h"

#define CWLT_MAX_ATTR 8
#define CWLT_HEAD_MAXV 256  // largest per-attribute vocabulary (4 values per lane)

namespace cwlt {

struct HeadArgs {
    int n[CWLT_MAX_ATTR];
    int off[CWLT_MAX_ATTR];
    int n_attr;
};

// exp for the softmax: libm-accurate for f32 logits (parity path), hardware v_exp for bf16 logits
template <typename T>
__device__ __forceinline__ float sm_exp(float x) { return sizeof(T) == 2 ? __expf(x) : expf(x); }

// softmax pieces of one attribute segment held 4-per-lane
struct Seg {
    float x[4];
    float mx, sum;
};

template <typename T>
__device__ __forceinline__ Seg load_seg(const T* row, int off, int n, int lane) {
    Seg s;
    float m = -INFINITY;
    const int ns = (n + 63) >> 6;            // 64-wide slots in use (wave-uniform): most attributes need one
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s.x[j] = -INFINITY;
        if (j < ns) {
            const int c = lane + 64 * j;
            s.x[j] = c < n ? load1(row + off + c) : -INFINITY;
            m = fmaxf(m, s.x[j]);
        }
    }
    s.mx = wave_max(m);
    float e = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < ns) {
            const int c = lane + 64 * j;
            if (c < n) e += sm_exp<T>(s.x[j] - s.mx);
        }
    }
    s.sum = wave_sum(e);
    return s;
}

// loss_part[blockIdx.x][A]; optional argmax (rows, A) int64, pmax (rows, A) f32, probs (rows, ldp) f32
template <typename T>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const T* __restrict__ logits, HeadArgs a,
                                                        const int64_t* __restrict__ target,
                                                        const float* __restrict__ mask, float* __restrict__ loss_part,
                                                        int64_t* __restrict__ argmax, float* __restrict__ pmax,
                                                        float* __restrict__ probs, long rows, long ld, long ldp) {
    __shared__ float red[4][CWLT_MAX_ATTR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[CWLT_MAX_ATTR];
#pragma unroll
    for (int f = 0; f < CWLT_MAX_ATTR; ++f) acc[f] = 0.f;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        const T* row = logits + r * ld;
        const float mk = mask ? mask[r] : 1.f;
#pragma unroll
        for (int f = 0; f < CWLT_MAX_ATTR; ++f) {
            if (f >= a.n_attr) break;
            const int n = a.n[f], off = a.off[f];
            const Seg s = load_seg(row, off, n, lane);
            if (target) {
                long t = target[r * a.n_attr + f];
                t = t < 0 ? 0 : (t >= n ? n - 1 : t);
                // x_t lives in lane t & 63, slot t >> 6
                const int slot = (int)(t >> 6);
                float xt = slot == 0 ? s.x[0] : (slot == 1 ? s.x[1] : (slot == 2 ? s.x[2] : s.x[3]));
                xt = __shfl(xt, (int)(t & 63), 64);
                acc[f] += mk * ((logf(s.sum) + s.mx) - xt);
            }
            if (argmax || pmax || probs) {
                // first index of the largest softmax VALUE (softmax then argmax, as the reference does)
                float best = -1.f;
                int bi = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = lane + 64 * j;
                    if (64 * j < n && c < n) {
                        const float p = sm_exp<T>(s.x[j] - s.mx) / s.sum;
                        if (probs) probs[r * ldp + off + c] = p;
                        if (p > best) { best = p; bi = c; }
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ob = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
                if (lane == 0) {
                    if (argmax) argmax[r * a.n_attr + f] = bi;
                    if (pmax) pmax[r * a.n_attr + f] = best;
                }
            }
        }
    }
    if (loss_part) {
        if (lane == 0)
#pragma unroll
            for (int f = 0; f < CWLT_MAX_ATTR; ++f) red[wave][f] = acc[f];
        __syncthreads();
        if (threadIdx.x < a.n_attr)
            loss_part[(long)blockIdx.x * a.n_attr + threadIdx.x] =
                (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// dlogits[r, seg f] = (softmax - onehot(target)) * mask_r * coef[f]; columns outside every segment <- 0
template <typename T>
__global__ __launch_bounds__(256) void heads_ce_bwd_kernel(const T* __restrict__ logits, HeadArgs a,
                                                           const int64_t* __restrict__ target,
                                                           const float* __restrict__ mask,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ wrf, T* __restrict__ dlogits,
                                                           long rows, long ld, int ncols_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int used = 0;
    for (int f = 0; f < a.n_attr; ++f) used = max(used, a.off[f] + a.n[f]);
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        const T* row = logits + r * ld;
        T* drow = dlogits + r * ld;
        const float mk = mask ? mask[r] : 1.f;
#pragma unroll
        for (int f = 0; f < CWLT_MAX_ATTR; ++f) {
            if (f >= a.n_attr) break;
            const int n = a.n[f], off = a.off[f];
            const Seg s = load_seg(row, off, n, lane);
            long t = target[r * a.n_attr + f];
            t = t < 0 ? 0 : (t >= n ? n - 1 : t);
            const float w = wrf ? wrf[r * a.n_attr + f] : mk * coef[f];   // per-(row, attribute) weight or mask*coef
            const float inv = 1.0f / s.sum;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = lane + 64 * j;
                if (64 * j < n && c < n) {
                    const float p = sm_exp<T>(s.x[j] - s.mx) * inv;
                    store1(drow + off + c, (p - (c == (int)t ? 1.f : 0.f)) * w);
                }
            }
        }
        for (int c = used + lane; c < ncols_total; c += 64) store1(drow + c, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// Tiled variants (the ones the launchers use whenever the row layout allows 16-byte accesses).
// The wave-per-row kernels above spend a token row on ~12 dependent cross-lane reductions per attribute and
// read the logits 2 bytes per lane.  Here a workgroup stages 32 rows x ld columns in LDS with 16-byte coalesced
// loads (row stride odd in words -> the column walks below are bank-conflict free), then ONE THREAD owns one
// (row, attribute) softmax and walks its n_f columns serially in LDS: no cross-lane traffic at all.  Attributes
// are assigned to the 8 half-wave slots in order of decreasing vocabulary so both halves of a wave loop alike.
// Results that are matrices (probs, dlogits) go back through the LDS tile and leave with coalesced stores.
// ------------------------------------------------------------------------------------------------
// The tile kernels walk a row's n_f logits serially in LDS.  Written one element per iteration the loop is a chain of
// dependent LDS round trips (~100 cycles each: 40 k cycles per 32-row tile at 135 classes); these
// helpers keep 8 independent reads in flight per step.
__device__ __forceinline__ float row_max(const float* x, int n) {
    float mx = -INFINITY;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[j + u];
        mx = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])), mx));
    }
    for (; j < n; ++j) mx = fmaxf(mx, x[j]);
    return mx;
}
// sum_j exp(x_j - mx); STORE: x_j <- exp(x_j - mx)
template <typename T, bool STORE>
__device__ __forceinline__ float row_sumexp(float* x, int n, float mx) {
    float sum = 0.f;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sm_exp<T>(v[u] - mx);
        if (STORE) {
#pragma unroll
            for (int u = 0; u < 8; ++u) x[j + u] = v[u];
        }
        // the summation order of the one-by-one loop: bit-identical results
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; j < n; ++j) {
        const float e = sm_exp<T>(x[j] - mx);
        if (STORE) x[j] = e;
        sum += e;
    }
    return sum;
}

constexpr int HT_ROWS = 32;

struct TileOrder {
    int attr[CWLT_MAX_ATTR];   // slot -> attribute index (-1: idle slot)
};

template <typename T>
__device__ __forceinline__ void heads_tile_load(const T* __restrict__ logits, float* xs, long r0, long rows, long ld,
                                                int W1) {
    constexpr int V = VecIO<T>::N;
    const int nv = (int)(ld / V);
    for (int i = threadIdx.x; i < HT_ROWS * nv; i += 256) {
        const int r = i / nv, cv = i - r * nv;
        float x[V];
        if (r0 + r < rows) {
            VecIO<T>::load(logits + (r0 + r) * ld + cv * V, x);
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) x[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) xs[r * W1 + cv * V + j] = x[j];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void heads_fwd_tile_kernel(const T* __restrict__ logits, HeadArgs a, TileOrder ord,
                                                             const int64_t* __restrict__ target,
                                                             const float* __restrict__ mask,
                                                             float* __restrict__ loss_part,
                                                             int64_t* __restrict__ argmax, float* __restrict__ pmax,
                                                             float* __restrict__ probs, long rows, long ld, long ldp,
                                                             int W1, int used) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [HT_ROWS][W1]
    __shared__ float red[CWLT_MAX_ATTR][HT_ROWS];
    const int row = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const int f = ord.attr[slot];
    const int n = f >= 0 ? a.n[f] : 0, off = f >= 0 ? a.off[f] : 0;
    float acc = 0.f;
    const long ntile = (rows + HT_ROWS - 1) / HT_ROWS;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long r0 = t * HT_ROWS, r = r0 + row;
        __syncthreads();
        heads_tile_load<T>(logits, xs, r0, rows, ld, W1);
        __syncthreads();
        if (f >= 0 && r < rows) {
            float* x = xs + row * W1 + off;
            const float mx = row_max(x, n);
            const float sum = row_sumexp<T, false>(x, n, mx);
            if (target) {
                long tg = target[r * a.n_attr + f];
                tg = tg < 0 ? 0 : (tg >= n ? n - 1 : tg);
                acc += (mask ? mask[r] : 1.f) * ((logf(sum) + mx) - x[tg]);
            }
            if (argmax || pmax || probs) {
                // first index of the largest softmax VALUE (softmax then argmax, as the reference does)
                float best = -1.f;
                int bi = 0;
                int j = 0;
                for (; j + 8 <= n; j += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = x[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = sm_exp<T>(v[u] - mx) / sum;
                    if (probs) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[j + u] = v[u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (v[u] > best) { best = v[u]; bi = j + u; }
                }
                for (; j < n; ++j) {
                    const float p = sm_exp<T>(x[j] - mx) / sum;
                    if (probs) x[j] = p;
                    if (p > best) { best = p; bi = j; }
                }
                if (argmax) argmax[r * a.n_attr + f] = bi;
                if (pmax) pmax[r * a.n_attr + f] = best;
            }
        }
        if (probs) {
            __syncthreads();
            for (int i = threadIdx.x; i < HT_ROWS * used; i += 256) {
                const int rr = i / used, c = i - rr * used;
                if (r0 + rr < rows) probs[(r0 + rr) * ldp + c] = xs[rr * W1 + c];
            }
        }
    }
    if (loss_part) {
        if (f >= 0) red[slot][row] = acc;
        __syncthreads();
        if (threadIdx.x < CWLT_MAX_ATTR && ord.attr[threadIdx.x] >= 0) {
            float sum = 0.f;
            for (int i = 0; i < HT_ROWS; ++i) sum += red[threadIdx.x][i];
            loss_part[(long)blockIdx.x * a.n_attr + ord.attr[threadIdx.x]] = sum;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void heads_ce_bwd_tile_kernel(const T* __restrict__ logits, HeadArgs a,
                                                                TileOrder ord, const int64_t* __restrict__ target,
                                                                const float* __restrict__ mask,
                                                                const float* __restrict__ coef,
                                                                const float* __restrict__ wrf, T* __restrict__ dlogits,
                                                                long rows, long ld, int W1, int used) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    constexpr int V = VecIO<T>::N;
    const int row = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const int f = ord.attr[slot];
    const int n = f >= 0 ? a.n[f] : 0, off = f >= 0 ? a.off[f] : 0;
    const int nv = (int)(ld / V);
    const long ntile = (rows + HT_ROWS - 1) / HT_ROWS;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long r0 = t * HT_ROWS, r = r0 + row;
        __syncthreads();
        heads_tile_load<T>(logits, xs, r0, rows, ld, W1);
        __syncthreads();
        if (f >= 0 && r < rows) {
            float* x = xs + row * W1 + off;
            const float mx = row_max(x, n);
            const float sum = row_sumexp<T, true>(x, n, mx);
            long tg = target[r * a.n_attr + f];
            tg = tg < 0 ? 0 : (tg >= n ? n - 1 : tg);
            const float w = wrf ? wrf[r * a.n_attr + f] : (mask ? mask[r] : 1.f) * coef[f];
            const float inv = 1.0f / sum;
            int j = 0;
            for (; j + 8 <= n; j += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = x[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[j + u] = (v[u] * inv - (j + u == (int)tg ? 1.f : 0.f)) * w;
            }
            for (; j < n; ++j) x[j] = (x[j] * inv - (j == (int)tg ? 1.f : 0.f)) * w;
        } else if (f < 0 && slot == CWLT_MAX_ATTR - 1) {
            // the last (always idle when n_attr < 8) slot clears the padding columns of its row
            for (int c = used; c < (int)ld; ++c) xs[row * W1 + c] = 0.f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < HT_ROWS * nv; i += 256) {
            const int rr = i / nv, cv = i - rr * nv;
            if (r0 + rr < rows) {
                float x[V];
#pragma unroll
                for (int j = 0; j < V; ++j) x[j] = xs[rr * W1 + cv * V + j];
                VecIO<T>::store(dlogits + (r0 + rr) * ld + cv * V, x);
            }
        }
    }
}

// slot order: attributes by decreasing vocabulary; returns false when the tiled kernels cannot be used
static bool tile_plan(const HeadArgs& a, int64_t ld, const void* p0, const void* p1, int vec, TileOrder& ord, int& W1,
                      size_t& lds) {
    if (a.n_attr >= CWLT_MAX_ATTR) return false;               // the bwd kernel needs one idle slot for the padding
    if (ld % vec || ((uintptr_t)p0 & 15) || ((uintptr_t)p1 & 15)) return false;
    W1 = (int)ld | 1;
    lds = (size_t)HT_ROWS * W1 * sizeof(float);
    if (lds > 60 * 1024) return false;
    int idx[CWLT_MAX_ATTR];
    for (int f = 0; f < a.n_attr; ++f) idx[f] = f;
    for (int i = 1; i < a.n_attr; ++i)
        for (int j = i; j > 0 && a.n[idx[j]] > a.n[idx[j - 1]]; --j) { int t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t; }
    for (int s = 0; s < CWLT_MAX_ATTR; ++s) ord.attr[s] = s < a.n_attr ? idx[s] : -1;
    return true;
}

static int fill_heads(HeadArgs& a, const int* n_class, int n_attr) {
    if (!n_class || n_attr <= 0 || n_attr > CWLT_MAX_ATTR) return CWLT_ERR_ARG;
    int off = 0;
    for (int f = 0; f < n_attr; ++f) {
        if (n_class[f] <= 0 || n_class[f] > CWLT_HEAD_MAXV) return CWLT_ERR_ARG;
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

int cwlt_heads_blocks(int64_t rows) {
    int64_t b = (rows + 3) / 4;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

/* logits (rows, ld): attribute f in columns [sum_{g<f} n_class[g], +n_class[f]); n_class: HOST array.
 * target (rows, n_attr) int64 and mask (rows) f32 may be NULL (then no loss).  Outputs, each may be
 * NULL: loss_sum (n_attr) f32 = sum_r mask_r * nll_{r,f}  [needs loss_part: cwlt_heads_blocks(rows)
 * * n_attr floats]; argmax (rows, n_attr) int64; pmax (rows, n_attr) f32 = softmax value at argmax;
 * probs (rows, ldp) f32 full softmax. */
int cwlt_heads_fwd(const void* logits, const int* n_class, int n_attr, const int64_t* target, const float* mask,
                   float* loss_part, float* loss_sum, int64_t* argmax, float* pmax, float* probs, int64_t rows,
                   int64_t ld, int64_t ldp, int dtype, void* stream) {
    using namespace cwlt;
    HeadArgs a;
    int e = fill_heads(a, n_class, n_attr);
    if (e) return e;
    const int used = a.off[n_attr - 1] + a.n[n_attr - 1];
    if (!logits || rows < 0 || ld < used) return CWLT_ERR_ARG;
    if (loss_sum && (!loss_part || !target)) return CWLT_ERR_ARG;
    if (probs && ldp < used) return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return loss_sum ? (int)hipMemsetAsync(loss_sum, 0, sizeof(float) * n_attr, st) : CWLT_OK;
    int nb = cwlt_heads_blocks(rows);
    const int64_t* tg = loss_sum ? target : nullptr;
    float* lp = loss_sum ? loss_part : nullptr;
    TileOrder ord;
    int W1 = 0;
    size_t lds = 0;
    if ((dtype == CWLT_F32 || dtype == CWLT_BF16) &&
        tile_plan(a, ld, logits, logits, dtype == CWLT_BF16 ? 8 : 4, ord, W1, lds)) {
        const int64_t ntile = (rows + HT_ROWS - 1) / HT_ROWS;
        if (ntile < nb) nb = (int)ntile;
        if (dtype == CWLT_F32)
            hipLaunchKernelGGL((heads_fwd_tile_kernel<float>), dim3(nb), dim3(256), lds, st, (const float*)logits, a,
                               ord, tg, mask, lp, argmax, pmax, probs, (long)rows, (long)ld, (long)ldp, W1, used);
        else
            hipLaunchKernelGGL((heads_fwd_tile_kernel<bf16_t>), dim3(nb), dim3(256), lds, st, (const bf16_t*)logits, a,
                               ord, tg, mask, lp, argmax, pmax, probs, (long)rows, (long)ld, (long)ldp, W1, used);
    } else if (dtype == CWLT_F32)
        hipLaunchKernelGGL((heads_fwd_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)logits, a, tg, mask, lp,
                           argmax, pmax, probs, (long)rows, (long)ld, (long)ldp);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((heads_fwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)logits, a, tg, mask,
                           lp, argmax, pmax, probs, (long)rows, (long)ld, (long)ldp);
    else
        return CWLT_ERR_DTYPE;
    e = (int)hipGetLastError();
    if (e || !loss_sum) return e;
    return launch_colsum_finalize(loss_part, loss_sum, nb, (long)n_attr, n_attr, 1.0f, 0, st);
}

/* coef (n_attr) f32 on device: upstream gradient of loss_f divided by sum(mask). */
int cwlt_heads_ce_bwd(const void* logits, const int* n_class, int n_attr, const int64_t* target, const float* mask,
                      const float* coef, void* dlogits, int64_t rows, int64_t ld, int dtype, void* stream) {
    using namespace cwlt;
    HeadArgs a;
    int e = fill_heads(a, n_class, n_attr);
    if (e) return e;
    const int used = a.off[n_attr - 1] + a.n[n_attr - 1];
    if (!logits || !target || !coef || !dlogits || rows < 0 || ld < used) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    hipStream_t st = (hipStream_t)stream;
    int nb = cwlt_heads_blocks(rows);
    TileOrder ord;
    int W1 = 0;
    size_t lds = 0;
    if ((dtype == CWLT_F32 || dtype == CWLT_BF16) &&
        tile_plan(a, ld, logits, dlogits, dtype == CWLT_BF16 ? 8 : 4, ord, W1, lds)) {
        const int64_t ntile = (rows + HT_ROWS - 1) / HT_ROWS;
        if (ntile < nb) nb = (int)ntile;
        if (dtype == CWLT_F32)
            hipLaunchKernelGGL((heads_ce_bwd_tile_kernel<float>), dim3(nb), dim3(256), lds, st, (const float*)logits, a,
                               ord, target, mask, coef, (const float*)nullptr, (float*)dlogits, (long)rows, (long)ld,
                               W1, used);
        else
            hipLaunchKernelGGL((heads_ce_bwd_tile_kernel<bf16_t>), dim3(nb), dim3(256), lds, st, (const bf16_t*)logits,
                               a, ord, target, mask, coef, (const float*)nullptr, (bf16_t*)dlogits, (long)rows,
                               (long)ld, W1, used);
        return (int)hipGetLastError();
    }
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((heads_ce_bwd_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)logits, a, target,
                           mask, coef, (const float*)nullptr, (float*)dlogits, (long)rows, (long)ld, (int)ld);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((heads_ce_bwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)logits, a, target,
                           mask, coef, (const float*)nullptr, (bf16_t*)dlogits, (long)rows, (long)ld, (int)ld);
    else
        return CWLT_ERR_DTYPE;
    return (int)hipGetLastError();
}

/* Gradient of sum_{r,f} g[r][f] * log softmax_f(logits_r)[target[r][f]]:
 * dlogits[r, seg f] = (onehot(target) - softmax) * g[r][f];  g (rows, n_attr) f32 on device. */
int cwlt_heads_logp_bwd(const void* logits, const int* n_class, int n_attr, const int64_t* target, const float* g,
                        void* dlogits, int64_t rows, int64_t ld, int dtype, void* stream) {
    using namespace cwlt;
    HeadArgs a;
    int e = fill_heads(a, n_class, n_attr);
    if (e) return e;
    const int used = a.off[n_attr - 1] + a.n[n_attr - 1];
    if (!logits || !target || !g || !dlogits || rows < 0 || ld < used) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    hipStream_t st = (hipStream_t)stream;
    int nb = cwlt_heads_blocks(rows);
    // the kernel computes (softmax - onehot) * w; the caller passes w = -g
    TileOrder ord;
    int W1 = 0;
    size_t lds = 0;
    if ((dtype == CWLT_F32 || dtype == CWLT_BF16) &&
        tile_plan(a, ld, logits, dlogits, dtype == CWLT_BF16 ? 8 : 4, ord, W1, lds)) {
        const int64_t ntile = (rows + HT_ROWS - 1) / HT_ROWS;
        if (ntile < nb) nb = (int)ntile;
        if (dtype == CWLT_F32)
            hipLaunchKernelGGL((heads_ce_bwd_tile_kernel<float>), dim3(nb), dim3(256), lds, st, (const float*)logits, a,
                               ord, target, (const float*)nullptr, (const float*)nullptr, g, (float*)dlogits,
                               (long)rows, (long)ld, W1, used);
        else
            hipLaunchKernelGGL((heads_ce_bwd_tile_kernel<bf16_t>), dim3(nb), dim3(256), lds, st, (const bf16_t*)logits,
                               a, ord, target, (const float*)nullptr, (const float*)nullptr, g, (bf16_t*)dlogits,
                               (long)rows, (long)ld, W1, used);
        return (int)hipGetLastError();
    }
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((heads_ce_bwd_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)logits, a, target,
                           (const float*)nullptr, (const float*)nullptr, g, (float*)dlogits, (long)rows, (long)ld,
                           (int)ld);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((heads_ce_bwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)logits, a, target,
                           (const float*)nullptr, (const float*)nullptr, g, (bf16_t*)dlogits, (long)rows, (long)ld,
                           (int)ld);
    else
        return CWLT_ERR_DTYPE;
    return (int)hipGetLastError();
}

}  // extern "C"
