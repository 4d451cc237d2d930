// L2 -> LDS DMA (buffer_load_dwordx4 ... lds) throughput of one CU from an L2-resident buffer, against the number of waves
// issuing and the number of 1 KiB pieces each keeps in flight.  The question behind HISTORY 9.7(e): the FFN kernels and
// gemm_ln.hip both take in 40-43 GB/s of operand pieces per CU -- latency x depth, or the path's throughput?
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate tools/probes/ldsdma_rate.hip && ./ldsdma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int DEPTH>
__global__ __launch_bounds__(1024) void dma(const char* __restrict__ src, uint32_t bytes, int pieces, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    u32x4_t rs;
    rs[0] = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)src);
    rs[1] = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)src >> 32));
    rs[2] = bytes;
    rs[3] = 0x00020000u;
    // each wave owns DEPTH 1 KiB slots of LDS and keeps DEPTH pieces in flight
    const uint32_t lbase = (uint32_t)(uintptr_t)(lds_void*)lds + w * DEPTH * 1024;
    uint32_t off = ((blockIdx.x * nw + w) * 1024u * 37u) % (bytes - 65536) & ~1023u;
    const uint32_t voff = lane * 16;
    for (int p = 0; p < pieces; ++p) {
        const uint32_t la = lbase + (p % DEPTH) * 1024;
        unsigned keep;
        // wait until at most DEPTH - 1 pieces are in flight, then issue the next
        asm volatile("s_waitcnt vmcnt(%c5)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\t"
                     "buffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(la), "s"(rs), "s"(off), "n"(DEPTH - 1)
                     : "memory", "scc");
        off += 1024u * 64u;
        if (off >= bytes - 65536) off -= bytes - 65536;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && lds[5] == 77 && lds[1029] == 78) sink[0] = 1;
}

int main() {
    const uint32_t bytes = 2u << 20;                   // 2 MiB: L2-resident (the FFN weight is 2 MiB too)
    char* src;
    uint32_t* sink;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 1, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int pieces = 2048;
    auto run = [&](auto kern, int depth, int waves, int W) {
        float best = 1e9f;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(kern, dim3(W), dim3(64 * waves), depth * 1024 * waves, 0, src, bytes, pieces, sink);
            CK(hipGetLastError());
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (it && ms < best) best = ms;
        }
        const double gb = (double)W * waves * pieces * 1024.0 / (best * 1e-3) / 1e9;
        printf("workgroups %3d x %2d waves, %d pieces in flight per wave (%3d KiB per workgroup): %7.1f GB/s per workgroup, %8.1f GB/s chip\n",
               W, waves, depth, depth * waves, gb / W, gb);
    };
    for (int W : {8, 256}) {
        for (int waves : {4, 8, 16}) {
            run(dma<1>, 1, waves, W);
            run(dma<2>, 2, waves, W);
            run(dma<4>, 4, waves, W);
            run(dma<8>, 8, waves, W);
        }
    }
    return 0;
}
