// What f32-MFMA rate does the part SUSTAIN, and for how long?  Pure v_mfma_f32_16x16x4_f32 streams (4 independent accumulator chains
// per wave, no memory, no LDS, no other vector work), one workgroup per CU on all 256 CUs, launches of ~5 ms ... ~2.5 s back to back;
// prints TFLOP/s and the shader clock each launch ran at (s_memtime cycles / event time).  Companion of f32_tile_structure_probe.hip.
//   hipcc --offload-arch=gfx950 -O3 f32_sustained_rate.hip -o f32_sustained_rate && ./f32_sustained_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k16(float* out, long long* cyc, int iters, float a0, float b0) {
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 * (threadIdx.x % 7), b = b0 * (threadIdx.x % 5);
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));   // the longest wave (the older wave of a SIMD gets the pipe first)
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
    long long* cyc; hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (float scale : {1.f, 0.f}) {
        printf(scale != 0.f ? "-- non-zero operands\n" : "-- all-zero operands\n");
        for (int wpc : {8, 4})
            for (int iters : {2000, 20000, 200000, 1000000, 2000}) {
                hipMemset(cyc, 0, 8);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k16, dim3(256), dim3(wpc * 64), 0, 0, out, cyc, iters, 0.001f * scale, 0.002f * scale);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                const double flop = 16.0 * 2048.0 * iters * wpc * 256.0;
                printf("%d waves/CU  %8d iterations: %9.3f ms  %6.1f TFLOP/s  MFMA-busy %5.1f %% of the longest wave's cycles  clock %.2f GHz\n", wpc, iters, ms,
                       flop / ms / 1e9, 100.0 * 16.0 * iters * 32.0 * (wpc / 4) / (double)c, c / (ms * 1e-3) / 1e9);
            }
    }
    return 0;
}
