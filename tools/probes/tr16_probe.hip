#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, const unsigned short* in) {
    __shared__ __attribute__((aligned(16))) unsigned short t[64 * 72];
    for (int i = threadIdx.x; i < 64 * 72; i += 64) t[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    // block: rows 4*g .. 4*g+3, cols 0..15 -> lane 4q+p supplies &t[row q][4p]
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    lds_s16x4* ptr = (lds_s16x4*)(t + (4 * g + q) * 72 + 4 * p);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short h[64 * 72], *din, *dout, o[256];
    for (int r = 0; r < 64; ++r) for (int c = 0; c < 72; ++c) h[r * 72 + c] = r * 100 + c;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, din);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    return 0;
}
