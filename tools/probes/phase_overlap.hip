// A model of a GEMM whose tiles end in a store burst (HISTORY 4.2c): 512 workgroups of 256 threads, two per CU; each does
// `tiles` times  [ M: read 16 x 24 KiB of an L2-resident slab, the operand pieces of a 16-step main loop | S: store one
// 128-row x 512-byte tile of each of two outputs = 128 KiB ].  Times: M only, S only, M + S with all workgroups starting
// together, M + S with the first two workgroups of every CU started 0..7 x ~4 us apart.  If the last is close to
// max(M only, S only) the store bursts hide under other CUs' main loops once the chip is out of phase; if it is close to
// their sum they do not.
//   hipcc --offload-arch=gfx950 -O3 -o phase_overlap tools/probes/phase_overlap.hip && ./phase_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void model(uint4* __restrict__ out, const uint4* __restrict__ slab, uint32_t* __restrict__ sink,
                                                int tiles, int do_m, int do_s, int spread, int msteps) {
    const int tid = threadIdx.x, wg = blockIdx.x;
    if (spread)
        for (int i = (wg >> 3) & 7; i > 0; --i) __builtin_amdgcn_s_sleep(127);
    const uint4* s = slab + (size_t)(wg & 7) * (1 << 16);
    uint32_t acc = 0;
    const uint4 v = make_uint4(tid, wg, 3, 4);
    for (int t = 0; t < tiles; ++t) {
        if (do_m) {
            for (int k = 0; k < msteps; ++k) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {                        // 6 x 4 KiB = 24 KiB per step
                    const uint4 x = s[(((t * msteps + k) * 6 + i) * 256 + tid) & 0xffff];
                    acc += x.x ^ x.w;
                }
                __syncthreads();                                    // a step's barrier
            }
        }
        if (do_s) {
            uint4* base = out + ((size_t)wg * tiles + t) * (2 * 128 * 4096 / 16);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                uint4* p = base + (size_t)(tid / 32 + 8 * i) * (4096 / 16) + (tid & 31);
                *p = v;
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int tiles = 32, W = 512;
    const int msteps = argc > 1 ? atoi(argv[1]) : 16;
    uint4 *out, *slab;
    uint32_t* sink;
    CK(hipMalloc(&out, (size_t)W * tiles * 2 * 128 * 4096));        // 16 GiB
    CK(hipMalloc(&slab, 8 << 20));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(slab, 1, 8 << 20));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto run = [&](int m, int s, int spread) {
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(model, dim3(W), dim3(256), 0, 0, out, slab, sink, tiles, m, s, spread, msteps);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (it && ms < best) best = ms;
        }
        return best * 1e3f;
    };
    printf("main-loop steps per tile %d, %d tiles per workgroup, %d workgroups\n", msteps, tiles, W);
    printf("M only            %8.1f us\n", run(1, 0, 0));
    printf("S only            %8.1f us   (%.0f GB/s)\n", run(0, 1, 0), W * tiles * 131072.0 / run(0, 1, 0) / 1e3);
    printf("M + S, in phase   %8.1f us\n", run(1, 1, 0));
    printf("M + S, spread     %8.1f us\n", run(1, 1, 1));
    printf("M only, spread    %8.1f us\n", run(1, 0, 1));
    return 0;
}
