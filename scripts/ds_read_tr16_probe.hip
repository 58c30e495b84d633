#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    // every lane: 8-byte address of "its" 4 elements, rows of 4: lane i -> elements 4i .. 4i+3
    auto p = reinterpret_cast<s16x4 __attribute__((address_space(3)))*>((__attribute__((address_space(3))) unsigned short*)lds + 4 * threadIdx.x);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16, lane i addresses elements 4i..4i+3 (element id = source lane * 4 + j):\n");
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (%2d,%d)", h[l*4+j] / 4, h[l*4+j] % 4); printf("\n"); }
    return 0;
}
