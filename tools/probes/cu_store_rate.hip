// How fast can ONE compute unit push stores to HBM, and what do a co-resident workgroup's L2-hit loads get meanwhile?
// (The question behind HISTORY 4.2c: the one-kernel FFN forward is main loop + store drain, nothing overlaps.)
//   store-only:  W workgroups (256 threads) each write `tiles` tiles of 128 rows x 512 bytes (row stride 4 096 B: the FFN
//                forward's output tile), 16 bytes per lane; W = 8 ... 512 -> GB/s in total and per workgroup (= per CU while
//                W <= 256: workgroups are dealt round-robin over the XCDs and their CUs).
//   load-only:   W workgroups each stream an L2-resident 1 MiB slab (the GEMM's operand pieces) over and over.
//   both:        2 W workgroups, even ones store, odd ones load: with W = 256 every CU holds one of each.
//   hipcc --offload-arch=gfx950 -O3 -o cu_store_rate tools/probes/cu_store_rate.hip && ./cu_store_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: every workgroup stores; 1: every workgroup loads; 2: even store / odd load
__global__ __launch_bounds__(256) void probe(uint4* __restrict__ out, const uint4* __restrict__ slab, uint32_t* __restrict__ sink,
                                             int tiles, int mode, int nt, unsigned long long* __restrict__ dur = nullptr) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    const int wg = blockIdx.x;
    const bool storer = mode == 0 || (mode == 2 && !(wg & 1));
    const int me = mode == 2 ? wg >> 1 : wg;
    if (storer) {
        // tile t of this workgroup: rows of 512 B at stride 4 096 B; thread -> (row = tid / 32 + 8 i, 16-byte piece tid % 32)
        const uint4 v = make_uint4(tid, wg, 3, 4);
        for (int t = 0; t < tiles; ++t) {
            uint4* base = out + ((size_t)me * tiles + t) * (128 * 4096 / 16);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                uint4* p = base + (size_t)(tid / 32 + 8 * i) * (4096 / 16) + (tid & 31);
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 vv = {v.x, v.y, v.z, v.w};
                if (nt) __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(p));
                else *p = v;
            }
        }
        if (dur) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) dur[me] = __builtin_amdgcn_s_memrealtime() - t0;
        }
    } else {
        uint32_t acc = 0;
        const uint4* s = slab + (size_t)(me & 7) * (1 << 16);      // 8 slabs of 1 MiB, all L2-resident after the first pass
        for (int t = 0; t < tiles * 2; ++t) {                        // 64 KiB per "tile": 256 threads x 16 x 16 B
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint4 x = s[((t * 16 + i) * 256 + tid) & 0xffff];
                acc += x.x ^ x.w;
            }
        }
        if (acc == 0x12345678u) sink[0] = acc;
    }
}

int main() {
    const int tiles = 64;
    uint4 *out, *slab;
    uint32_t* sink;
    CK(hipMalloc(&out, (size_t)512 * tiles * 128 * 4096));          // 16 GiB
    CK(hipMalloc(&slab, 8 << 20));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(slab, 1, 8 << 20));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto run = [&](int W, int mode, int nt) {
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(probe, dim3(mode == 2 ? 2 * W : W), dim3(256), 0, 0, out, slab, sink, tiles, mode, nt);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (it && ms < best) best = ms;
        }
        return best;
    };
    const double tile_bytes = 128.0 * 512, load_bytes = 2.0 * 65536;
    printf("# stores: tiles of 128 rows x 512 B (row stride 4096 B), %d per workgroup\n", tiles);
    for (int nt = 0; nt < 2; ++nt)
        for (int W : {8, 32, 64, 128, 256, 512}) {
            const float ms = run(W, 0, nt);
            const double gbs = W * tiles * tile_bytes / (ms * 1e-3) / 1e9;
            printf("store-only %s W=%3d: %8.1f us  %7.1f GB/s total  %6.1f GB/s per workgroup\n", nt ? "nt     " : "default", W,
                   ms * 1e3, gbs, gbs / W);
        }
    for (int W : {8, 64, 256}) {
        const float ms = run(W, 1, 0);
        const double gbs = W * tiles * load_bytes / (ms * 1e-3) / 1e9;
        printf("load-only (L2-resident slab) W=%3d: %8.1f us  %7.1f GB/s total  %6.1f GB/s per workgroup\n", W, ms * 1e3, gbs, gbs / W);
    }
    // how long does a STORING workgroup take next to a loading one on its CU?  (100 MHz timestamps, median over workgroups)
    {
        unsigned long long* dur;
        CK(hipMalloc(&dur, 8 * 512));
        static unsigned long long h[512];
        for (int W : {8, 64, 128}) {
            for (int mode : {0, 2}) {
                CK(hipMemset(dur, 0, 8 * 512));
                for (int it = 0; it < 3; ++it)
                    hipLaunchKernelGGL(probe, dim3(mode == 2 ? 2 * W : W), dim3(256), 0, 0, out, slab, sink, tiles, mode, 0, dur);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, dur, 8 * W, hipMemcpyDeviceToHost));
                unsigned long long s2 = 0, mx = 0;
                for (int i = 0; i < W; ++i) { s2 += h[i]; if (h[i] > mx) mx = h[i]; }
                printf("storing workgroup, W=%3d %s: mean %.1f us  max %.1f us  -> %.1f GB/s per workgroup\n", W,
                       mode == 2 ? "next to a loading one" : "alone on its CU      ", s2 / (double)W / 100.0, mx / 100.0,
                       tiles * tile_bytes / (s2 / (double)W / 100.0 * 1e-6) / 1e9);
            }
        }
    }
    for (int nt = 0; nt < 2; ++nt)
        for (int W : {8, 64, 256}) {
            const float ms = run(W, 2, nt);
            printf("both %s W=%3d+%3d: %8.1f us  (store-only took %.1f, load-only %.1f)\n", nt ? "nt     " : "default", W, W, ms * 1e3,
                   run(W, 0, nt) * 1e3, run(W, 1, 0) * 1e3);
        }
    return 0;
}
