// Compound-word (CW) multi-embedding front end: A per-attribute table gathers * sqrt(d_emb),
// concatenated along the feature dim, in one pass.  fwd + bwd.
//
// Replaces, from /root/reference/dqn_policy/model.py:67-74,206-221 (same code in
// ppo_policy/model.py:69-76,208-223 and dqn_policy/AIRL_model.py:34-41,108-115):
//     emb_f = Embeddings_f(x[..., f]) = lut_f(x[..., f]) * sqrt(d_emb_f)        f = 0..5
//     embs  = torch.cat([emb_tempo, emb_chord, emb_barbeat, emb_pitch, emb_duration, emb_velocity], -1)
// i.e. six gather kernels, six multiplies and a concat (three HBM passes over (B*T, 1216)).
//
// tokens: (rows, A) int64, coalesced (48 B per CW token); tables stay L2-resident (372 KB total).
// Backward is deterministic: a single-wave workgroup owns a 64-column slab of one table and a
// slab of token rows, accumulates into an LDS copy of the table slab (plain read-modify-write, no
// atomics), and writes a partial that a fixed-order tree sums.  HBM-bound.
#include "cwlt_common.h"

#define CWLT_MAX_ATTR 8

namespace cwlt {

struct EmbedArgs {
    const float* tab[CWLT_MAX_ATTR];  // (nrows[f], width[f]) f32 row-major
    int width[CWLT_MAX_ATTR];
    int nrows[CWLT_MAX_ATTR];
    int off[CWLT_MAX_ATTR];     // column offset in the concatenated output
    int tabofs[CWLT_MAX_ATTR];  // element offset of table f in the flat gradient buffer
    float scale[CWLT_MAX_ATTR];
    int n_attr;
    int dcat;
    int total;  // sum nrows*width
};

__device__ __forceinline__ int clamp_id(long id, int n) { return id < 0 ? 0 : (id >= n ? n - 1 : (int)id); }

template <typename T>
__global__ __launch_bounds__(256) void cw_embed_fwd_kernel(const int64_t* __restrict__ tokens, EmbedArgs a,
                                                           T* __restrict__ out, long rows, int rows_per_block,
                                                           long ldo) {
    int f[2], lc[2];
    bool act[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = (threadIdx.x + 256 * j) * 4;
        act[j] = col < a.dcat;
        f[j] = 0;
        lc[j] = 0;
        for (int t = 0; t < a.n_attr; ++t)
            if (col >= a.off[t] && col < a.off[t] + a.width[t]) {
                f[j] = t;
                lc[j] = col - a.off[t];
            }
    }
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (long r = r0; r < r1; ++r) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!act[j]) continue;
            const int id = clamp_id(tokens[r * a.n_attr + f[j]], a.nrows[f[j]]);
            float4 v = load4(a.tab[f[j]] + (long)id * a.width[f[j]] + lc[j]);
            const float s = a.scale[f[j]];
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
            store4(out + r * ldo + (threadIdx.x + 256 * j) * 4, v);
        }
    }
}

// grid: (dcat/64 column slabs, row splits); block = one wave
template <typename T>
__global__ __launch_bounds__(64) void cw_embed_bwd_kernel(const int64_t* __restrict__ tokens, EmbedArgs a,
                                                          const T* __restrict__ dout, float* __restrict__ part,
                                                          long rows, long ldd) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    const int lane = threadIdx.x;
    int f = 0, cb = 0;
    {
        int slab = blockIdx.x;
        for (int t = 0; t < a.n_attr; ++t) {
            const int ns = a.width[t] >> 6;
            if (slab < ns) { f = t; cb = slab; break; }
            slab -= ns;
        }
    }
    const int nr = a.nrows[f], wd = a.width[f];
    for (int i = lane; i < nr * 64; i += 64) acc[i] = 0.f;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
    const float s = a.scale[f];
    const T* dp = dout + a.off[f] + cb * 64 + lane;
    const int64_t* tp = tokens + f;
    long r = r0;
    for (; r + 3 < r1; r += 4) {
        const int i0 = clamp_id(tp[(r + 0) * a.n_attr], nr), i1 = clamp_id(tp[(r + 1) * a.n_attr], nr);
        const int i2 = clamp_id(tp[(r + 2) * a.n_attr], nr), i3 = clamp_id(tp[(r + 3) * a.n_attr], nr);
        const float v0 = load1(dp + (r + 0) * ldd), v1 = load1(dp + (r + 1) * ldd);
        const float v2 = load1(dp + (r + 2) * ldd), v3 = load1(dp + (r + 3) * ldd);
        acc[i0 * 64 + lane] += v0 * s;
        acc[i1 * 64 + lane] += v1 * s;
        acc[i2 * 64 + lane] += v2 * s;
        acc[i3 * 64 + lane] += v3 * s;
    }
    for (; r < r1; ++r) acc[clamp_id(tp[r * a.n_attr], nr) * 64 + lane] += load1(dp + r * ldd) * s;
    float* pp = part + (long)blockIdx.y * a.total + a.tabofs[f] + cb * 64 + lane;
    for (int id = 0; id < nr; ++id) pp[(long)id * wd] = acc[id * 64 + lane];
}

static int fill_args(EmbedArgs& a, const void* const* tables, const int* widths, const int* nrows, int n_attr) {
    if (!tables || !widths || !nrows || n_attr <= 0 || n_attr > CWLT_MAX_ATTR) return CWLT_ERR_ARG;
    int off = 0, tot = 0;
    for (int f = 0; f < n_attr; ++f) {
        if (!tables[f] || widths[f] <= 0 || (widths[f] & 63) || nrows[f] <= 0) return CWLT_ERR_ARG;
        a.tab[f] = (const float*)tables[f];
        a.width[f] = widths[f];
        a.nrows[f] = nrows[f];
        a.off[f] = off;
        a.tabofs[f] = tot;
        a.scale[f] = sqrtf((float)widths[f]);
        off += widths[f];
        tot += widths[f] * nrows[f];
    }
    a.n_attr = n_attr;
    a.dcat = off;
    a.total = tot;
    return off > 2048 ? CWLT_ERR_ARG : CWLT_OK;
}

}  // namespace cwlt

extern "C" {

int cwlt_embed_splits(int64_t rows) {
    int64_t b = (rows + 255) / 256;
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return (int)b;
}

/* tokens (rows, n_attr) int64; tables/widths/nrows: HOST arrays of n_attr entries (device pointers
 * to f32 tables, embedding widths (multiples of 64), vocabulary sizes).  out (rows, sum widths),
 * row stride ldo.  Ids outside [0, nrows) are clamped (the reference would raise IndexError). */
int cwlt_cw_embed_fwd(const int64_t* tokens, const void* const* tables, const int* widths, const int* nrows,
                      int n_attr, void* out, int64_t rows, int64_t ldo, int dtype, void* stream) {
    using namespace cwlt;
    EmbedArgs a;
    int e = fill_args(a, tables, widths, nrows, n_attr);
    if (e) return e;
    if (rows < 0 || ldo < a.dcat || (ldo & 3)) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    if (!tokens || !out) return CWLT_ERR_ARG;
    const int rpb = 16;
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((cw_embed_fwd_kernel<float>), grid, block, 0, st, tokens, a, (float*)out, (long)rows, rpb,
                           (long)ldo);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((cw_embed_fwd_kernel<bf16_t>), grid, block, 0, st, tokens, a, (bf16_t*)out, (long)rows, rpb,
                           (long)ldo);
    else
        return CWLT_ERR_DTYPE;
    return (int)hipGetLastError();
}

/* dtables: ONE flat f32 buffer holding the gradients of all tables back to back (table f at element
 * offset sum_{g<f} nrows[g]*widths[g]); part: cwlt_embed_splits(rows) * total floats. */
int cwlt_cw_embed_bwd(const int64_t* tokens, const int* widths, const int* nrows, int n_attr, const void* dout,
                      float* part, float* dtables, int64_t rows, int64_t ldd, int dtype, void* stream) {
    using namespace cwlt;
    EmbedArgs a;
    const void* dummy[CWLT_MAX_ATTR];
    for (int f = 0; f < CWLT_MAX_ATTR; ++f) dummy[f] = (const void*)dtables;
    int e = fill_args(a, dummy, widths, nrows, n_attr);
    if (e) return e;
    if (!dtables || rows < 0 || ldd < a.dcat) return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return (int)hipMemsetAsync(dtables, 0, sizeof(float) * a.total, st);
    if (!tokens || !dout || !part) return CWLT_ERR_ARG;
    int maxr = 0;
    for (int f = 0; f < n_attr; ++f) maxr = nrows[f] > maxr ? nrows[f] : maxr;
    const size_t lds = (size_t)maxr * 64 * sizeof(float);
    if (lds > 160 * 1024) return CWLT_ERR_ARG;
    const int ns = cwlt_embed_splits(rows);
    const dim3 grid(a.dcat / 64, ns), block(64);
    if (dtype == CWLT_F32) {
        if (lds > 64 * 1024)
            hipFuncSetAttribute((const void*)cw_embed_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        hipLaunchKernelGGL((cw_embed_bwd_kernel<float>), grid, block, lds, st, tokens, a, (const float*)dout, part,
                           (long)rows, (long)ldd);
    } else if (dtype == CWLT_BF16) {
        if (lds > 64 * 1024)
            hipFuncSetAttribute((const void*)cw_embed_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        hipLaunchKernelGGL((cw_embed_bwd_kernel<bf16_t>), grid, block, lds, st, tokens, a, (const bf16_t*)dout, part,
                           (long)rows, (long)ldd);
    } else {
        return CWLT_ERR_DTYPE;
    }
    e = (int)hipGetLastError();
    if (e) return e;
    return launch_colsum_finalize(part, dtables, ns, (long)a.total, a.total, 1.0f, 0, st);
}

}  // extern "C"
