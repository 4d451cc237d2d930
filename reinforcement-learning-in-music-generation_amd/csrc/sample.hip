// Device-side categorical sampling of the next CW token -- the sampling half of the reference's PPO-side
// generation loop (/root/reference/ppo_policy/inference.py:115-141: softmax of each attribute's logits,
// torch.distributions.Categorical(...).sample(), actions concatenated and fed back as the next input).
//
// ... and of the DQN-side loop's samplers (dqn_policy/model.py:19-55,281-286: temperature, nucleus) when the caller
// asks for device-side sampling.
//
// One workgroup per song, one wave per attribute: softmax (max / sum by DPP reductions), inclusive prefix sum of
// the probabilities (lane-blocked layout: lane l owns classes 4l .. 4l+3, so class order == lane order), one
// uniform draw from the counter-based generator of cwlt_common.h keyed by (seed, draw counter, song, attribute),
// first class whose cumulative mass exceeds u * total.  The chosen ids go straight into the decode step's token
// buffer (and, optionally, row `counter` of the song), so a generation loop needs no host round trip per token.
// The draw counter is a device int64 read by the kernel: a captured hipGraph draws fresh numbers on every replay.
// Same distribution as the reference's sampler, not the same stream (torch's Philox stream is device- and
// version-specific anyway).
#include "cwlt_common.h"

#include <climits>

#define CWLT_MAX_ATTR 8

namespace cwlt {

struct SampleArgs {
    int n[CWLT_MAX_ATTR];
    int off[CWLT_MAX_ATTR];
    float inv_t[CWLT_MAX_ATTR];                      // 1 / temperature per attribute
    float top_p[CWLT_MAX_ATTR];                      // nucleus mass per attribute; >= 1: plain categorical
};

__global__ __launch_bounds__(64 * CWLT_MAX_ATTR) void sample_categorical_kernel(
    const float* __restrict__ logits, long ld, SampleArgs A, int n_attr, uint64_t seed,
    const int64_t* __restrict__ counter, int64_t* __restrict__ tokens, int64_t* __restrict__ song, long song_rows) {
    __shared__ float e_s[CWLT_MAX_ATTR][256];
    const int lane = threadIdx.x & 63, a = threadIdx.x >> 6, n = blockIdx.x;
    if (a >= n_attr) return;                         // wave-uniform; no workgroup barriers in this kernel
    const int nc = A.n[a];
    const float* x = logits + (long)n * ld + A.off[a];
    const long step = counter ? *counter : 0;
    float v[4];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 4 + j;
        v[j] = c < nc ? x[c] * A.inv_t[a] : -INFINITY;
        m = fmaxf(m, v[j]);
    }
    m = wave_max(m);
    float e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = lane * 4 + j < nc ? expf(v[j] - m) : 0.f;
    if (A.top_p[a] < 1.0f) {
        // nucleus (dqn_policy/model.py:33-47): in descending-probability order keep every class whose PRECEDING
        // mass is <= p (the class that crosses p is kept); probabilities there are exp/(sum + 1e-5).  The mass
        // ahead of class i needs no sort: G_i = sum of e_j over classes ranked before i (larger e, ties: larger
        // index first, as argsort()[::-1] orders them).  One broadcast LDS read per class, four running sums per lane.
        float* ew = e_s[a];
#pragma unroll
        for (int j = 0; j < 4; ++j) ew[lane * 4 + j] = e[j];
        float tot = (e[0] + e[1]) + (e[2] + e[3]);
        tot = wave_sum(tot);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        float ahead[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nc; ++c) {
            const float ec = ew[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = lane * 4 + j;
                ahead[j] += (ec > e[j] || (ec == e[j] && c > i)) ? ec : 0.f;
            }
        }
        const float limit = A.top_p[a] * (tot * (1.0f + 1e-5f));
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = ahead[j] <= limit ? e[j] : 0.f;
    }
    float run = 0.f, cum[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        run += e[j];
        cum[j] = run;                                // inclusive within the lane
    }
    // exclusive prefix over lanes (Hillis-Steele on the lane totals)
    float inc = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    const float before = inc - run;
    const float total = lane_value(inc, 63);
    const uint32_t r = rng_pair(seed, ((uint64_t)step * gridDim.x + n) * CWLT_MAX_ATTR + a);
    const float u = (float)(r >> 8) * (1.0f / 16777216.0f);      // [0, 1)
    const float target = u * total;
    int pick = INT_MAX;
#pragma unroll
    for (int j = 3; j >= 0; --j) {
        const int c = lane * 4 + j;
        if (c < nc && e[j] > 0.f && before + cum[j] > target) pick = c;
    }
    // the first lane (lowest classes) whose cumulative mass passes the target wins
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pick = min(pick, __shfl_xor(pick, d, 64));
    if (pick == INT_MAX) {                           // u * total rounded past the last kept class: take that class
        int last = -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) last = e[j] > 0.f ? lane * 4 + j : last;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) last = max(last, __shfl_xor(last, d, 64));
        pick = last < 0 ? 0 : last;
    }
    if (lane == 0) {
        tokens[(long)n * n_attr + a] = pick;
        if (song && step < song_rows) song[((long)step * gridDim.x + n) * n_attr + a] = pick;
    }
}

}  // namespace cwlt

extern "C" int cwlt_sample_categorical(const float* logits, const int* n_class, const float* temperature,
                                       const float* top_p, int n_attr, int64_t rows, int64_t ld, uint64_t seed, const int64_t* counter,
                                       int64_t* tokens, int64_t* song, int64_t song_rows, void* stream) {
    using namespace cwlt;
    if (!logits || !n_class || !tokens || n_attr <= 0 || n_attr > CWLT_MAX_ATTR || rows <= 0) return CWLT_ERR_ARG;
    SampleArgs A;
    int off = 0;
    for (int a = 0; a < n_attr; ++a) {
        if (n_class[a] <= 0 || n_class[a] > 256) return CWLT_ERR_ARG;
        if (temperature && !(temperature[a] > 0.f)) return CWLT_ERR_ARG;
        A.n[a] = n_class[a];
        A.off[a] = off;
        A.inv_t[a] = temperature ? 1.0f / temperature[a] : 1.0f;
        A.top_p[a] = top_p ? top_p[a] : 1.0f;
        if (!(A.top_p[a] > 0.f)) return CWLT_ERR_ARG;
        off += n_class[a];
    }
    if (ld < off) return CWLT_ERR_ARG;
    hipLaunchKernelGGL(sample_categorical_kernel, dim3((unsigned)rows), dim3(64 * n_attr), 0, (hipStream_t)stream, logits,
                       (long)ld, A, n_attr, seed, counter, tokens, song, (long)song_rows);
    return (int)hipGetLastError();
}
