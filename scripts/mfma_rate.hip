// Raw f32 MFMA issue-rate probe: how many independent accumulator chains does v_mfma_f32_16x16x4_f32 /
// v_mfma_f32_32x32x2_f32 need to reach the 64 FLOP/clk/SIMD peak?  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / CHAINS; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
__global__ void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8 / CHAINS; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double flop_per_wave_iter, int waves_per_cu, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = flop_per_wave_iter * iters * waves_per_cu * 256.0;
    printf("%-34s %d waves/CU: %8.3f ms  %7.1f TFLOP/s\n", name, waves_per_cu, ms, flop / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int iters = 20000;
    for (int wpc : {4, 8}) {
        const int threads = wpc * 64;
        run("16x16x4 1 chain ", [&] { hipLaunchKernelGGL(k16<1>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, 16 * 2048.0, wpc, iters);
        run("16x16x4 2 chains", [&] { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, 16 * 2048.0, wpc, iters);
        run("16x16x4 4 chains", [&] { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, 16 * 2048.0, wpc, iters);
        run("32x32x2 1 chain ", [&] { hipLaunchKernelGGL(k32<1>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, 8 * 4096.0, wpc, iters);
        run("32x32x2 2 chains", [&] { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, 8 * 4096.0, wpc, iters);
    }
    return 0;
}
