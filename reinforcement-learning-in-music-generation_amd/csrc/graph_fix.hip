// hipGraph surgery: every MEMSET node of a captured graph becomes a fill KERNEL node.
//
// Why.  On ROCm 7.2 a hipMemsetAsync captured into a hipGraph is replayed with a WRONG fill pattern from the second replay
// on when other work runs between replays (tools/probes/graph_memset_probe.py, profiles/r03_graph_memset_probe.txt: half of
// a 32 MiB buffer reads 0x00800000 instead of 0).  The steps this library replays as graphs (ops.GraphedCall: the RL
// rollout and update steps) contain such nodes without asking for them: PyTorch's reduce_kernel zeroes its cross-block
// semaphores with a memset, hipBLASLt zeroes split-K workspaces with one (tools/diag_memset_sites.py) -- a stale pattern
// there means a reduction that never finds its last block, or a GEMM accumulating onto garbage.  Kernel nodes replay
// correctly, so the captured graph is edited before it is instantiated.  No counterpart in the reference (it has no
// graphs): infrastructure of the drop-in RL loops (IRL_dqn_train.py / ppo_train.py).
#include "cwlt_common.h"
#include <stdlib.h>
#include <string.h>

namespace cwlt {

// rows x width elements of elem_size bytes (1, 2 or 4), row pitch in bytes, every element = the low bytes of value
__global__ __launch_bounds__(256) void graph_fill_kernel(unsigned char* dst, unsigned int value, unsigned int elem_size,
                                                         size_t width, size_t height, size_t pitch) {
    const size_t row_bytes = width * elem_size;
    const size_t n = row_bytes * height;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t r = i / row_bytes, c = i - r * row_bytes;
        dst[r * pitch + c] = (unsigned char)(value >> (8 * (c % elem_size)));
    }
}

}  // namespace cwlt

extern "C" {

/* Replace every memset node of `graph` (a hipGraph_t that has not been instantiated yet) by a kernel node that writes
 * the same bytes, with the same dependencies and dependents.  *replaced (may be NULL) receives the number of nodes
 * replaced.  Child graphs are not entered.  Returns 0 or a hipError_t; on error the graph may be partly edited and must
 * not be replayed. */
int cwlt_graph_replace_memset_nodes(void* graph, int* replaced) {
    using namespace cwlt;
    if (replaced) *replaced = 0;
    if (!graph) return CWLT_ERR_ARG;
    hipGraph_t g = (hipGraph_t)graph;
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(g, nullptr, &n);
    if (e != hipSuccess || n == 0) return (int)e;
    hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(n * sizeof(hipGraphNode_t));
    if (!nodes) return CWLT_ERR_ARG;
    e = hipGraphGetNodes(g, nodes, &n);
    int done = 0;
    for (size_t i = 0; e == hipSuccess && i < n; ++i) {
        hipGraphNodeType t;
        e = hipGraphNodeGetType(nodes[i], &t);
        if (e != hipSuccess || t != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp;
        e = hipGraphMemsetNodeGetParams(nodes[i], &mp);
        if (e != hipSuccess) break;
        if (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4) { e = hipErrorInvalidValue; break; }
        size_t nd = 0, no = 0;
        e = hipGraphNodeGetDependencies(nodes[i], nullptr, &nd);
        if (e != hipSuccess) break;
        e = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &no);
        if (e != hipSuccess) break;
        hipGraphNode_t* deps = (hipGraphNode_t*)malloc((nd + no + 1) * sizeof(hipGraphNode_t));
        if (!deps) { e = hipErrorOutOfMemory; break; }
        hipGraphNode_t* outs = deps + nd;
        if (nd) e = hipGraphNodeGetDependencies(nodes[i], deps, &nd);
        if (e == hipSuccess && no) e = hipGraphNodeGetDependentNodes(nodes[i], outs, &no);
        if (e == hipSuccess) {
            unsigned char* dst = (unsigned char*)mp.dst;
            unsigned int value = mp.value, es = mp.elementSize;
            size_t width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
            if (height == 1) pitch = width * es;
            const size_t bytes = width * es * height;
            size_t blocks = (bytes + 256 * 16 - 1) / (256 * 16);
            if (blocks < 1) blocks = 1;
            if (blocks > 1024) blocks = 1024;
            void* args[] = {&dst, &value, &es, &width, &height, &pitch};
            hipKernelNodeParams kp = {};
            kp.func = (void*)graph_fill_kernel;
            kp.gridDim = dim3((unsigned)blocks);
            kp.blockDim = dim3(256);
            kp.sharedMemBytes = 0;
            kp.kernelParams = args;
            kp.extra = nullptr;
            hipGraphNode_t k;
            e = hipGraphAddKernelNode(&k, g, nd ? deps : nullptr, nd, &kp);
            for (size_t j = 0; e == hipSuccess && j < no; ++j) e = hipGraphAddDependencies(g, &k, &outs[j], 1);
            if (e == hipSuccess) e = hipGraphDestroyNode(nodes[i]);
            if (e == hipSuccess) ++done;
        }
        free(deps);
    }
    free(nodes);
    if (replaced) *replaced = done;
    return (int)e;
}

}  // extern "C"
