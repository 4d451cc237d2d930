// Memory skeleton of the one-sweep attention backward (csrc/cla_bf16.hip, cla_bwd_sweep_bf16_kernel): the same grid
// (one 512-thread workgroup per (sequence, head) stream), the same per-chunk access pattern (64 token rows x 128 bytes of
// each of q, k, v, dout, out + 4 bytes of zinv per row in; 64 x 128 bytes of dq, dk, dv out), the same two-chunks-ahead
// register prefetch and reverse chunk order -- and NO arithmetic, LDS traffic or barriers.  What it measures: the time the
// chip needs to move the sweep's bytes in the sweep's pattern, i.e. the memory roofline of this layout.
//   hipcc --offload-arch=gfx950 -O3 -o scan_skeleton tools/probes/scan_skeleton.hip && ./scan_skeleton [B] [T] [packed]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int C = 64, D = 64;
struct Regs { uint4 q, k, v, g, o; float z; };

// csrc/cla_bf16.hip: all heads of sequence n are neighbours in ONE XCD's dispatch order
__device__ __forceinline__ int stream_of_block(int b, int N, int H) {
    if (N & 7) return b;
    const int x = b & 7, j = b >> 3;
    return ((j / H) * 8 + x) * H + j % H;
}

template <bool XCD>
__global__ __launch_bounds__(512, 1) void skeleton(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* out,
                                                   const uint16_t* dout, const float* zinv, uint16_t* dq, uint16_t* dk,
                                                   uint16_t* dv, int H, int L, long ldq, long ldo) {
    const int tid = threadIdx.x;
    const int sid = XCD ? stream_of_block(blockIdx.x, gridDim.x / H, H) : blockIdx.x;
    const int n = sid / H, h = sid % H;
    const int srow = tid >> 3, scol = (tid & 7) * 8;
    const uint16_t* qb = q + (long)n * L * ldq + h * D;
    const uint16_t* kb = k + (long)n * L * ldq + h * D;
    const uint16_t* vb = v + (long)n * L * ldq + h * D;
    const uint16_t* ob = out + (long)n * L * ldo + h * D;
    const uint16_t* gb = dout + (long)n * L * ldo + h * D;
    const float* zb = zinv + (long)n * L * H + h;
    uint16_t* dqb = dq + (long)n * L * ldo + h * D;
    uint16_t* dkb = dk + (long)n * L * ldo + h * D;
    uint16_t* dvb = dv + (long)n * L * ldo + h * D;
    auto load = [&](Regs& R, int c) {
        const long row = (long)c * C + srow;
        R.q = *reinterpret_cast<const uint4*>(qb + row * ldq + scol);
        R.k = *reinterpret_cast<const uint4*>(kb + row * ldq + scol);
        R.v = *reinterpret_cast<const uint4*>(vb + row * ldq + scol);
        R.g = *reinterpret_cast<const uint4*>(gb + row * ldo + scol);
        R.o = *reinterpret_cast<const uint4*>(ob + row * ldo + scol);
        R.z = zb[row * H];
    };
    auto store = [&](const Regs& R, int c) {
        const long row = (long)c * C + srow;
        const uint32_t z = __float_as_uint(R.z);
        // the sweep's store threads: waves 0-3 write dk and dv, waves 4-7 write dq, two rows each; here every thread
        // writes its own row of all three (same bytes per workgroup and chunk)
        *reinterpret_cast<uint4*>(dqb + row * ldo + scol) = make_uint4(R.q.x ^ R.g.x, R.q.y ^ R.g.y, R.q.z ^ R.g.z, R.q.w ^ z);
        *reinterpret_cast<uint4*>(dkb + row * ldo + scol) = make_uint4(R.k.x ^ R.o.x, R.k.y ^ R.o.y, R.k.z ^ R.o.z, R.k.w ^ R.o.w);
        *reinterpret_cast<uint4*>(dvb + row * ldo + scol) = R.v;
    };
    const int nch = L / C;
    Regs RA, RB;
    load(RA, nch - 1);
    if (nch > 1) load(RB, nch - 2);
    for (int c = nch - 1; c >= 0; c -= 2) {
        store(RA, c);
        if (c >= 2) load(RA, c - 2);
        if (c >= 1) {
            store(RB, c - 1);
            if (c >= 3) load(RB, c - 3);
        }
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, T = argc > 2 ? atoi(argv[2]) : 1024, H = 8;
    const long R = (long)B * T;
    uint16_t *qkv, *out, *dout, *dq, *dk, *dv;
    float* z;
    CK(hipMalloc(&qkv, R * 1536 * 2));
    CK(hipMalloc(&out, R * 512 * 2));
    CK(hipMalloc(&dout, R * 512 * 2));
    CK(hipMalloc(&dq, R * 512 * 2));
    CK(hipMalloc(&dk, R * 512 * 2));
    CK(hipMalloc(&dv, R * 512 * 2));
    CK(hipMalloc(&z, R * H * 4));
    CK(hipMemset(qkv, 1, R * 1536 * 2));
    CK(hipMemset(out, 2, R * 512 * 2));
    CK(hipMemset(dout, 3, R * 512 * 2));
    CK(hipMemset(z, 0, R * H * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const double bytes = (double)R * (8 * 512 * 2 + H * 4);
    for (int mode = 0; mode < 4; ++mode) {
        const bool packed = mode & 1, xcd = mode & 2;
        const long ldq = packed ? 1536 : 512;
        // unpacked: q, k, v as three (R, 512) planes inside the same allocation
        const uint16_t* qp = qkv;
        const uint16_t* kp = packed ? qkv + 512 : qkv + R * 512;
        const uint16_t* vp = packed ? qkv + 1024 : qkv + 2 * R * 512;
        float best = 1e9f, sum = 0.f;
        for (int it = 0; it < 12; ++it) {
            CK(hipEventRecord(a));
            if (xcd)
                hipLaunchKernelGGL(skeleton<true>, dim3(B * H), dim3(512), 0, 0, qp, kp, vp, out, dout, z, dq, dk, dv, H, T, ldq, 512L);
            else
                hipLaunchKernelGGL(skeleton<false>, dim3(B * H), dim3(512), 0, 0, qp, kp, vp, out, dout, z, dq, dk, dv, H, T, ldq, 512L);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (it >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("B=%d T=%d  q/k/v %s, blocks %s:  avg %.1f us  best %.1f us  -> %.0f GB/s (8 streams of 128-byte rows)\n", B, T,
               packed ? "packed (row stride 3072 B)" : "separate (row stride 1024 B)", xcd ? "heads of a sequence on one XCD" : "in stream order",
               sum / 10 * 1e3, best * 1e3, bytes / (sum / 10 * 1e-3) / 1e9);
    }
    return 0;
}
