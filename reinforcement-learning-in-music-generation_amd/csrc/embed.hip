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
#include "cwlt_mfma_bf16.h"

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

// A thread owns 8 adjacent columns of the concatenated row (every table width is a multiple of 8: 16-byte bf16 stores,
// two 16-byte f32 loads) and the threads of a block cover `rows_at_once` rows at a time (dcat / 8 = 152 chunks per row at
// the repo's widths: 320 threads = 2 rows, 16 idle lanes).  Round 2's form -- 4 columns per thread in two passes of 256
// threads -- left three quarters of the second pass idle and stored 8 bytes per lane: 0.48 ms for the bench's 1.27 GB.
template <typename T>
__global__ __launch_bounds__(320) void cw_embed_fwd_kernel(const int64_t* __restrict__ tokens, EmbedArgs a,
                                                           T* __restrict__ out, long rows, int rows_per_block,
                                                           long ldo) {
    const int nchunk = a.dcat >> 3;                         // 8-column chunks per row
    const int rows_at_once = max(1, (int)blockDim.x / nchunk);
    const int sub = threadIdx.x / nchunk, chunk = threadIdx.x - sub * nchunk;
    if (sub >= rows_at_once) return;                        // idle tail lanes
    const int col = chunk * 8;
    int f = 0, lc = 0;
    for (int t = 0; t < a.n_attr; ++t)
        if (col >= a.off[t] && col < a.off[t] + a.width[t]) {
            f = t;
            lc = col - a.off[t];
        }
    const float s = a.scale[f];
    const float* tab = a.tab[f];
    const int width = a.width[f], nr = a.nrows[f];
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (long r = r0 + sub; r < r1; r += rows_at_once) {
        const int id = clamp_id(tokens[r * a.n_attr + f], nr);
        float v[8];
        load8(tab + (long)id * width + lc, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= s;
        store8(out + r * ldo + col, v);
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

// bf16 storage: the same scatter-add as a GEMM on the MFMA pipe.  dTable_f (ids x 64 cols) = onehot(tokens_f)^T .
// dout_f, contracted over the token rows of this workgroup's row split:
//   A operand: the one-hot rows are GENERATED in registers (lane of id-row m compares 8 token ids with m);
//   B operand: a 16-row x 64-col tile of dout staged row-major in LDS and fetched transposed
//              (ds_read_b64_tr_b16), as in the scan / wgrad kernels;
//   accumulators: up to EB_MT x 2 tiles of 32 x 32 f32 (ids x cols), summed over the split in fixed order.
// The LDS read-modify-write version above (kept for f32) serialises one dependent LDS round trip per token
// row per wave; this one is bound by the coalesced read of dout.  One wave per workgroup.
constexpr int EB_MT = 5;          // id tiles of 32 -> vocabularies up to 160
constexpr int EB_MAXROWS = 4096;  // token rows per split whose ids fit the LDS id cache

__global__ __launch_bounds__(64, 2) void cw_embed_bwd_mfma_kernel(const int64_t* __restrict__ tokens, EmbedArgs a,
                                                               const bf16_t* __restrict__ dout,
                                                               float* __restrict__ part, long rows, long ldd) {
    using namespace b16;
    __shared__ __attribute__((aligned(16))) bf16_t ts[16 * LD];     // dout tile [16 token rows][64 cols]
    __shared__ __attribute__((aligned(16))) int ids[EB_MAXROWS + 16];
    const int lane = threadIdx.x, l31 = lane & 31, hf = lane >> 5;
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
    const int nmt = (nr + 31) >> 5;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
    const int n = (int)(r1 - r0);                       // <= EB_MAXROWS (checked by the launcher)
    const int npad = (n + 15) & ~15;
    for (int i = lane; i < npad; i += 64)
        ids[i] = i < n ? clamp_id(tokens[(r0 + i) * a.n_attr + f], nr) : -1;   // -1 matches no id row
    const bf16_t* dp = dout + a.off[f] + cb * 64;
    // this lane's slot of the 16 x 64 tile: row lane / 4 (16 rows), 16 columns starting at (lane % 4) * 16
    const int trow = lane >> 2, tcol = (lane & 3) * 16;
    uint4 p0, p1;
    // this split's rows of the 64-column slab as a buffer resource: rows >= n read back as zeros
    const __amdgpu_buffer_rsrc_t dr = make_rsrc(dp + r0 * ldd, n > 0 ? (uint32_t)(((long)(n - 1) * ldd + 64) * 2) : 0u);
#define EB_LOAD(k0)                                                                              \
    {                                                                                            \
        const uint32_t off = ((uint32_t)((k0) + trow) * (uint32_t)ldd + tcol) * 2;               \
        p0 = buf_load16(dr, off);                                                                \
        p1 = buf_load16(dr, off + 16);                                                           \
    }
    f32x16 acc[EB_MT][2];
#pragma unroll
    for (int mt = 0; mt < EB_MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = zero16();
    EB_LOAD(0);
    for (int k0 = 0; k0 < npad; k0 += 16) {
        *reinterpret_cast<uint4*>(ts + trow * LD + tcol) = p0;
        *reinterpret_cast<uint4*>(ts + trow * LD + tcol + 8) = p1;
        if (k0 + 16 < npad) { EB_LOAD(k0 + 16); }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the tile and the ids are visible to this (only) wave
        const int4 ia = *reinterpret_cast<const int4*>(ids + k0 + 8 * hf);
        const int4 ib = *reinterpret_cast<const int4*>(ids + k0 + 8 * hf + 4);
        const bf16x8 b0 = tfrag8(ts, 0, 0, lane), b1 = tfrag8(ts, 0, 32, lane);
        const int id8[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
#pragma unroll
        for (int mt = 0; mt < EB_MT; ++mt) {
            if (mt < nmt) {
                const int m = 32 * mt + l31;
                bf16x8 oh;
#pragma unroll
                for (int j = 0; j < 8; ++j) oh[j] = (__bf16)(id8[j] == m ? 1.0f : 0.0f);
                acc[mt][0] = mfma(oh, b0, acc[mt][0]);
                acc[mt][1] = mfma(oh, b1, acc[mt][1]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // transposed reads done before the next tile overwrites ts
    }
#undef EB_LOAD
    const float s = a.scale[f];
    float* pp = part + (long)blockIdx.y * a.total + a.tabofs[f] + cb * 64;
#pragma unroll
    for (int mt = 0; mt < EB_MT; ++mt) {
        if (mt < nmt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int id = 32 * mt + acc_row(r, hf);
                if (id < nr) {
                    pp[(long)id * wd + l31] = acc[mt][0][r] * s;
                    pp[(long)id * wd + 32 + l31] = acc[mt][1][r] * s;
                }
            }
        }
    }
}

// ---- the whole input front in one pass -----------------------------------------------------------------------------
// in_linear(cat_f(table_f[tok_f] * sqrt(w_f))) = sum_f P_f[tok_f] + b  with the PROJECTED tables
// P_f = sqrt(w_f) * table_f . W_in[:, cols_f]^T  (n_f x D; 339 rows in all at the repo's vocabularies, built by the
// caller with six small GEMMs per step), so embedding, concatenation, in_linear, positional encoding and dropout
// (dqn_policy/model.py:206-223 + 90-92) are ONE pass that reads 48 bytes of ids per token and writes the (rows, D)
// activations: the (rows, 1216) concatenated embeddings and the 1216 -> 512 GEMM over all token rows are gone.
// A thread owns V = 16-byte-of-output adjacent columns of a row (the posenc kernel's shape, so that the dropout stream
// is the one cwlt_posenc_dropout draws for the same seed: key = element offset r * D + c).  The tables are read from L2.
constexpr int EP_TILE = 32;   // token rows per block iteration
template <typename T, typename TT>
__global__ __launch_bounds__(256) void cw_embed_proj_fwd_kernel(const int64_t* __restrict__ tokens, const TT* __restrict__ tp,
                                                                EmbedArgs a, const float* __restrict__ bias,
                                                                const float* __restrict__ pe, T* __restrict__ out,
                                                                long rows, int Tlen, int D, uint32_t thresh,
                                                                float keep_scale, uint64_t seed,
                                                                const uint64_t* __restrict__ seed_base) {
    // element offsets of the EP_TILE x n_attr table rows of a tile: the ids are read once, coalesced, and the table loads
    // of a row do not wait behind an id load of their own (two dependent memory round trips per row otherwise)
    __shared__ int ofs[EP_TILE * CWLT_MAX_ATTR];
    if (seed_base) seed += *seed_base;
    constexpr int V = VecIO<T>::N;
    const int nd = D / V, A = a.n_attr;
    const long ntile = (rows + EP_TILE - 1) / EP_TILE;
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long r0 = tile * EP_TILE;
        const int nr = (int)min((long)EP_TILE, rows - r0);
        __syncthreads();                                   // the previous tile's offsets are no longer read
        if ((int)threadIdx.x < nr * A) {
            const int f = threadIdx.x % A;
            ofs[threadIdx.x] = a.tabofs[f] + clamp_id(tokens[r0 * A + threadIdx.x], a.nrows[f]) * D;
        }
        __syncthreads();
#pragma unroll 2
        for (int i = threadIdx.x; i < nr * nd; i += 256) {
            const int row = i / nd, c = (i - row * nd) * V;
            const long r = r0 + row;
            float v[CWLT_MAX_ATTR][V];
#pragma unroll
            for (int f = 0; f < CWLT_MAX_ATTR; ++f)
                if (f < A) VecIO<TT>::load(tp + ofs[row * A + f] + c, v[f]);
            float t[V];
            loadf<V>(bias + c, t);
            if (pe) {
                float q[V];
                loadf<V>(pe + (r % Tlen) * (long)D + c, q);
#pragma unroll
                for (int j = 0; j < V; ++j) t[j] += q[j];
            }
#pragma unroll
            for (int f = 0; f < CWLT_MAX_ATTR; ++f)
                if (f < A) {
#pragma unroll
                    for (int j = 0; j < V; ++j) t[j] += v[f][j];
                }
            const long off = r * D + c;
            if (thresh) {
                const uint32_t km = dropout_mask<V>(seed, off, thresh);
#pragma unroll
                for (int j = 0; j < V; ++j) t[j] = ((km >> j) & 1u) ? t[j] * keep_scale : 0.f;
            }
            VecIO<T>::store(out + off, t);
        }
    }
}

// dP of the projected tables (cwlt_cw_embed_proj_bwd, bf16): every attribute's one-hot GEMM contracts the SAME 64-column
// slab of dpre, so one workgroup stages the slab once and its four waves share it.  The work is cut into UNITS = (attribute,
// 32-id tile) pairs -- 13 at the repo's vocabularies (2 + 5 + 1 + 3 + 1 + 1) -- dealt to the waves in order, at most
// PB_MAXU per wave; per 64-row step a wave reads the slab's 8 transposed fragments once and issues 8 MFMAs per unit.
// (cw_embed_bwd_mfma_kernel run once per attribute reads the slab six times and builds every tile in every workgroup:
// 0.71 ms at the bench shape; this kernel: see HISTORY 4.3a.)
constexpr int PB_NW = 4;            // waves per workgroup
constexpr int PB_MAXU = 4;          // units per wave
constexpr int PB_CHUNK = 1024;      // token rows whose ids (all attributes, 16 bits each) sit in LDS at a time
constexpr int PB_STEP = 64;         // token rows per step
constexpr int PB_MAXSPLITS = 64;    // row splits (workgroups per column slab): 8 x 64 = 512 workgroups at D = 512

// eight one-hot bf16 values (1.0 where the 16-bit id equals m) from four id pairs: per pair x = id - m, min(x, 1) - 1 =
// 0xffff where equal, & 0x3f80 -- four packed 16-bit instructions instead of two compares, two selects and a pack
__device__ __forceinline__ uint32_t onehot_pair(uint32_t idp, uint32_t mp) {
    uint32_t r;
    const uint32_t one = 0x00010001u, bf1 = 0x3f803f80u;
    asm("v_pk_sub_u16 %0, %1, %2\n\t"               // 0 where the id is this lane's row
        "v_pk_min_u16 %0, %0, %3\n\t"               // 0 equal, 1 not
        "v_pk_sub_u16 %0, %0, %3\n\t"               // 0xffff equal, 0 not
        "v_and_b32 %0, %4, %0"                         // bf16 1.0 / 0.0
        : "=&v"(r)
        : "v"(idp), "v"(mp), "s"(one), "s"(bf1));
    return r;
}

__global__ __launch_bounds__(256) void cw_embed_proj_bwd_mfma_kernel(const int64_t* __restrict__ tokens, EmbedArgs a,
                                                                     const bf16_t* __restrict__ dpre,
                                                                     float* __restrict__ part, long rows, long ldd, int D) {
    using namespace b16;
    __shared__ __attribute__((aligned(16))) bf16_t ts[2][PB_STEP * LD];          // dpre tile [64 token rows][64 cols] x 2
    __shared__ __attribute__((aligned(16))) unsigned short ids[CWLT_MAX_ATTR][PB_CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = blockIdx.x, A = a.n_attr;
    // row split of this workgroup, in whole steps
    const long nstep_all = (rows + PB_STEP - 1) / PB_STEP;
    const long sper = (nstep_all + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * sper * PB_STEP, r1 = min(rows, r0 + sper * PB_STEP);
    // this wave's units
    int total_u = 0;
    for (int f = 0; f < A; ++f) total_u += (a.nrows[f] + 31) >> 5;
    const int upw = (total_u + PB_NW - 1) / PB_NW;        // <= PB_MAXU (launcher)
    int uf[PB_MAXU];                                      // attribute of unit u (-1: no unit)
    uint32_t um[PB_MAXU];                                 // this lane's id row of unit u, in both 16-bit halves
    {
        int u = 0;
#pragma unroll
        for (int k = 0; k < PB_MAXU; ++k) uf[k] = -1, um[k] = 0;
        for (int f = 0; f < A; ++f)
            for (int mt = 0; mt < ((a.nrows[f] + 31) >> 5); ++mt, ++u) {
                const int k = u - w * upw;
#pragma unroll
                for (int kk = 0; kk < PB_MAXU; ++kk)
                    if (kk == k && k < upw) uf[kk] = f, um[kk] = (uint32_t)(32 * mt + l31) * 0x10001u;
            }
    }
    f32x16 acc[PB_MAXU][2];
#pragma unroll
    for (int k = 0; k < PB_MAXU; ++k) acc[k][0] = zero16(), acc[k][1] = zero16();
    const bf16_t* dp = dpre + cb * 64;
    const int trow = tid >> 2, tcol = (tid & 3) * 16;     // 64 rows x 4 pieces of 16 columns
    uint4 p0, p1;
    for (long c0 = r0; c0 < r1; c0 += PB_CHUNK) {
        const int n = (int)min((long)PB_CHUNK, r1 - c0);
        const int npad = (n + PB_STEP - 1) / PB_STEP * PB_STEP;
        __syncthreads();                                  // the previous chunk's ids and tiles are no longer read
        for (int i = tid; i < npad * A; i += 256) {
            const int r = i / A, f = i - r * A;
            ids[f][r] = r < n ? (unsigned short)clamp_id(tokens[(c0 + r) * A + f], a.nrows[f]) : (unsigned short)0xffffu;
        }
        // this chunk's rows of the 64-column slab as a buffer resource: rows >= n read back as zeros
        const __amdgpu_buffer_rsrc_t dr = make_rsrc(dp + c0 * ldd, (uint32_t)(((long)(n - 1) * ldd + 64) * 2));
#define PB_LOAD(k0)                                                                              \
    {                                                                                            \
        const uint32_t off = ((uint32_t)((k0) + trow) * (uint32_t)ldd + tcol) * 2;               \
        p0 = buf_load16(dr, off);                                                                \
        p1 = buf_load16(dr, off + 16);                                                           \
    }
#define PB_PUT(b)                                                                                \
    {                                                                                            \
        *reinterpret_cast<uint4*>(ts[b] + trow * LD + tcol) = p0;                                \
        *reinterpret_cast<uint4*>(ts[b] + trow * LD + tcol + 8) = p1;                            \
    }
        PB_LOAD(0);
        PB_PUT(0);
        __syncthreads();                                  // tile 0 and the ids
        for (int k0 = 0, b = 0; k0 < npad; k0 += PB_STEP, b ^= 1) {
            const bool more = k0 + PB_STEP < npad;
            if (more) { PB_LOAD(k0 + PB_STEP); }
#pragma unroll
            for (int ks = 0; ks < PB_STEP / 16; ++ks) {
                const bf16x8 b0 = tfrag8(ts[b], 16 * ks, 0, lane), b1 = tfrag8(ts[b], 16 * ks, 32, lane);
#pragma unroll
                for (int k = 0; k < PB_MAXU; ++k) {
                    if (uf[k] >= 0) {
                        const uint4 ip = *reinterpret_cast<const uint4*>(ids[uf[k]] + k0 + 16 * ks + 8 * hf);
                        const uint4 o = make_uint4(onehot_pair(ip.x, um[k]), onehot_pair(ip.y, um[k]),
                                                   onehot_pair(ip.z, um[k]), onehot_pair(ip.w, um[k]));
                        const bf16x8 oh = __builtin_bit_cast(bf16x8, o);
                        acc[k][0] = mfma(oh, b0, acc[k][0]);
                        acc[k][1] = mfma(oh, b1, acc[k][1]);
                    }
                }
            }
            if (more) { PB_PUT(b ^ 1); }
            __syncthreads();                              // next tile written, this one no longer read
        }
#undef PB_LOAD
#undef PB_PUT
    }
    float* pp = part + (long)blockIdx.y * a.total + cb * 64;
#pragma unroll
    for (int k = 0; k < PB_MAXU; ++k) {
        if (uf[k] >= 0) {
            const int nr = a.nrows[uf[k]];
            float* pt = pp + a.tabofs[uf[k]];
            const int m0 = (int)(um[k] & 0xffffu) - l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int id = m0 + acc_row(r, hf);
                if (id < nr) {
                    pt[(long)id * D + l31] = acc[k][0][r];
                    pt[(long)id * D + 32 + l31] = acc[k][1][r];
                }
            }
        }
    }
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

// projected tables: every attribute D columns wide, all reading / writing the SAME D columns
static int fill_proj_args(EmbedArgs& a, const int* nrows, int n_attr, int D) {
    if (!nrows || n_attr <= 0 || n_attr > CWLT_MAX_ATTR || D <= 0 || (D & 63) || D > 2048) return CWLT_ERR_ARG;
    int rows = 0;
    for (int f = 0; f < n_attr; ++f) {
        if (nrows[f] <= 0) return CWLT_ERR_ARG;
        a.tab[f] = nullptr;
        a.width[f] = D;
        a.nrows[f] = nrows[f];
        a.off[f] = 0;
        a.tabofs[f] = rows * D;
        a.scale[f] = 1.0f;
        rows += nrows[f];
    }
    a.n_attr = n_attr;
    a.dcat = n_attr * D;
    a.total = rows * D;
    return CWLT_OK;
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
 * to f32 tables, embedding widths (multiples of 64), vocabulary sizes).  out (rows, sum widths), 16-byte aligned,
 * row stride ldo (a multiple of 8 elements for bf16, of 4 for f32).  Ids outside [0, nrows) are clamped (the reference would raise IndexError). */
int cwlt_cw_embed_fwd(const int64_t* tokens, const void* const* tables, const int* widths, const int* nrows,
                      int n_attr, void* out, int64_t rows, int64_t ldo, int dtype, void* stream) {
    using namespace cwlt;
    EmbedArgs a;
    int e = fill_args(a, tables, widths, nrows, n_attr);
    if (e) return e;
    // a thread stores 8 adjacent elements: 16-byte (bf16) / 2 x 16-byte (f32) accesses
    if (rows < 0 || ldo < a.dcat || (ldo & (dtype == CWLT_BF16 ? 7 : 3))) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    if (!tokens || !out || ((uintptr_t)out & 15)) return CWLT_ERR_ARG;
    const int rpb = 16;
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb)), block(320);
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
    } else if (dtype == CWLT_BF16 && maxr <= 32 * EB_MT && (rows + ns - 1) / ns <= EB_MAXROWS && !(ldd & 7) &&
               !((uintptr_t)dout & 15)) {
        hipLaunchKernelGGL(cw_embed_bwd_mfma_kernel, grid, block, 0, st, tokens, a, (const bf16_t*)dout, part,
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

/* out[r, :] = dropout(sum_f tproj[rowofs_f + tokens[r, f], :] + bias + pe[r % T, :]),  rowofs_f = sum_{g<f} nrows[g]:
 * embedding x6 + cat + in_linear + PositionalEncoding (dqn_policy/model.py:206-223, 90-92) in one pass over the token
 * rows, from the PROJECTED tables tproj = cat_f(sqrt(w_f) table_f . W_in[:, cols_f]^T) (sum nrows, D) that the caller
 * builds per step.  tproj has the dtype of out (bf16 / f32), dense; bias (D) and pe (max_len >= T, D) f32, pe may be NULL;
 * out (rows, D) dense, 16-byte aligned; D % 64 == 0.  The dropout stream is cwlt_posenc_dropout's for the same seed. */
int cwlt_cw_embed_proj_fwd(const int64_t* tokens, const void* tproj, const int* nrows, int n_attr, const float* bias,
                           const float* pe, void* out, int64_t rows, int T, int D, float p, uint64_t seed,
                           const uint64_t* seed_base, int dtype, void* stream) {
    using namespace cwlt;
    EmbedArgs a;
    int e = fill_proj_args(a, nrows, n_attr, D);
    if (e) return e;
    if (rows < 0 || T <= 0 || p < 0.f || p >= 1.f) return CWLT_ERR_ARG;
    if (rows == 0) return CWLT_OK;
    if (!tokens || !tproj || !bias || !out || (((uintptr_t)out | (uintptr_t)tproj | (uintptr_t)bias | (uintptr_t)pe) & 15))
        return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t th = drop_thresh(p);
    const float ks = drop_scale(p);
    long nb = (rows + EP_TILE - 1) / EP_TILE;
    if (nb > 256 * 32) nb = 256 * 32;
    if (dtype == CWLT_F32)
        hipLaunchKernelGGL((cw_embed_proj_fwd_kernel<float, float>), dim3((unsigned)nb), dim3(256), 0, st, tokens,
                           (const float*)tproj, a, bias, pe, (float*)out, (long)rows, T, D, th, ks, seed, seed_base);
    else if (dtype == CWLT_BF16)
        hipLaunchKernelGGL((cw_embed_proj_fwd_kernel<bf16_t, bf16_t>), dim3((unsigned)nb), dim3(256), 0, st, tokens,
                           (const bf16_t*)tproj, a, bias, pe, (bf16_t*)out, (long)rows, T, D, th, ks, seed, seed_base);
    else
        return CWLT_ERR_DTYPE;
    return (int)hipGetLastError();
}

/* dtproj[rowofs_f + id, :] = sum over the token rows r with tokens[r, f] == id of dpre[r, :]  (f32, (sum nrows, D)):
 * the gradient of cwlt_cw_embed_proj_fwd's projected tables from dpre = the gradient in front of the dropout
 * (cwlt_posenc_dropout with pe == NULL applied to dout).  The bias gradient is any one attribute's rows of dtproj summed.
 * part: cwlt_embed_splits(rows) * (sum nrows) * D f32.  dpre (rows, D) with row stride ldd.  Deterministic. */
int cwlt_cw_embed_proj_bwd(const int64_t* tokens, const int* nrows, int n_attr, int D, const void* dpre, float* part,
                           float* dtproj, int64_t rows, int64_t ldd, int dtype, void* stream) {
    using namespace cwlt;
    EmbedArgs a;
    int e = fill_proj_args(a, nrows, n_attr, D);
    if (e) return e;
    if (!dtproj || rows < 0 || ldd < D) return CWLT_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) return (int)hipMemsetAsync(dtproj, 0, sizeof(float) * a.total, st);
    if (!tokens || !dpre || !part) return CWLT_ERR_ARG;
    int maxr = 0, units = 0;
    for (int f = 0; f < n_attr; ++f) {
        maxr = nrows[f] > maxr ? nrows[f] : maxr;
        units += (nrows[f] + 31) / 32;
    }
    const size_t lds = (size_t)maxr * 64 * sizeof(float);
    if (lds > 160 * 1024) return CWLT_ERR_ARG;
    const int ns = cwlt_embed_splits(rows);
    // the six attributes of a column slab are neighbours in the grid (x = attribute-major would put them 8 apart: the
    // same XCD either way under round-robin placement), so the slab of dpre they all read is fetched from HBM once
    const dim3 grid(a.dcat / 64, ns), block(64);
    if (dtype == CWLT_F32) {
        if (lds > 64 * 1024)
            hipFuncSetAttribute((const void*)cw_embed_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        hipLaunchKernelGGL((cw_embed_bwd_kernel<float>), grid, block, lds, st, tokens, a, (const float*)dpre, part,
                           (long)rows, (long)ldd);
    } else if (dtype == CWLT_BF16 && units <= PB_NW * PB_MAXU && maxr < 65535 && !(ldd & 7) && !((uintptr_t)dpre & 15) &&
               (int64_t)PB_CHUNK * ldd * 2 < (1ll << 31)) {
        // fewer, longer row splits than the per-attribute kernels use (the caller's `part` is sized for more)
        const int nsp = ns < PB_MAXSPLITS ? ns : PB_MAXSPLITS;
        hipLaunchKernelGGL(cw_embed_proj_bwd_mfma_kernel, dim3(D / 64, nsp), dim3(64 * PB_NW), 0, st, tokens, a,
                           (const bf16_t*)dpre, part, (long)rows, (long)ldd, D);
        e = (int)hipGetLastError();
        if (e) return e;
        return launch_colsum_finalize(part, dtproj, nsp, (long)a.total, a.total, 1.0f, 0, st);
    } else if (dtype == CWLT_BF16 && maxr <= 32 * EB_MT && (rows + ns - 1) / ns <= EB_MAXROWS && !(ldd & 7) &&
               !((uintptr_t)dpre & 15)) {
        hipLaunchKernelGGL(cw_embed_bwd_mfma_kernel, grid, block, 0, st, tokens, a, (const bf16_t*)dpre, part, (long)rows,
                           (long)ldd);
    } else if (dtype == CWLT_BF16) {
        if (lds > 64 * 1024)
            hipFuncSetAttribute((const void*)cw_embed_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        hipLaunchKernelGGL((cw_embed_bwd_kernel<bf16_t>), grid, block, lds, st, tokens, a, (const bf16_t*)dpre, part,
                           (long)rows, (long)ldd);
    } else {
        return CWLT_ERR_DTYPE;
    }
    e = (int)hipGetLastError();
    if (e) return e;
    return launch_colsum_finalize(part, dtproj, ns, (long)a.total, a.total, 1.0f, 0, st);
}

}  // extern "C"
