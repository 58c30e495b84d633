// Do f32-input MFMA and ordinary f32 VALU work overlap when issued by two different waves of one SIMD?
//   mode 0: waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 idle
//   mode 1: waves 0-3 idle, waves 4-7 run a VALU (v_fma_f32) loop
//   mode 2: both at once  -> t2 ~ max(t0, t1) if the pipes are independent, ~ t0 + t1 if they are shared
// also with bf16 MFMA for comparison.   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF16>
__global__ __launch_bounds__(512) void k(float* out, int mode, int mfma_iters, int valu_iters) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode == 0 || mode == 2) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            float a = 1.f + threadIdx.x, b = 2.f + threadIdx.x * 0.5f;
            bf16x8 ha = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, hb = {8, 7, 6, 5, 4, 3, 2, 1};
            for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (BF16) {
                        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, a1, 0, 0, 0);
                    } else {
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a1, 0, 0, 0);
                    }
                }
            }
            r = a0[0] + a1[1];
        }
    } else {
        if (mode == 1 || mode == 2) {
            float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
            const float m = 1.0001f, c = 0.5f;
            for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x0 = fmaf(x0, m, c); x1 = fmaf(x1, m, c); x2 = fmaf(x2, m, c); x3 = fmaf(x3, m, c);
                    x4 = fmaf(x4, m, c); x5 = fmaf(x5, m, c); x6 = fmaf(x6, m, c); x7 = fmaf(x7, m, c);
                }
            }
            r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <bool BF16>
float run(float* out, int mode, int mi, int vi) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<BF16>, dim3(256), dim3(512), 0, 0, out, mode, mi, vi); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<BF16>, dim3(256), dim3(512), 0, 0, out, mode, mi, vi);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
    const int mi = 20000, vi = 40000;          // 16 MFMA x 32 cyc = 512 cyc / iter ; 32 FMA x ~2-4 cyc / iter
    float a = run<false>(out, 0, mi, vi), b = run<false>(out, 1, mi, vi), c = run<false>(out, 2, mi, vi);
    printf("f32  MFMA 16x16x4 : mfma-only %.3f ms | valu-only %.3f ms | both %.3f ms  (max %.3f, sum %.3f)\n", a, b, c, a > b ? a : b, a + b);
    a = run<true>(out, 0, mi * 4, vi); b = run<true>(out, 1, mi * 4, vi); c = run<true>(out, 2, mi * 4, vi);
    printf("bf16 MFMA 16x16x32: mfma-only %.3f ms | valu-only %.3f ms | both %.3f ms  (max %.3f, sum %.3f)\n", a, b, c, a > b ? a : b, a + b);
    return 0;
}
