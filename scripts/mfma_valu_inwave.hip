// What does one VALU instruction cost next to f32 MFMAs on gfx950 - in the SAME wave, and with a second wave on the SIMD?
//   loop body = 4 x { 1 MFMA (4 rotating accumulators) ; k filler instructions on independent registers }
//   WPS = waves per SIMD (1: 256-thread block, 2: 512-thread block); every wave runs the same stream.
// Reported: shader cycles (s_memtime) per MFMA slot, per SIMD.  If the f32 matrix pipe were independent of the VALU the
// figure would stay at the MFMA's own issue time (32 / 64 cycles) until k fills the gap; if f32 MFMAs run on the
// vector lanes it grows by the filler's cost from k = 1.  bf16 16x16x32 is the control (separate matrix pipe).
// hipcc --offload-arch=gfx950 -O3 mfma_valu_inwave.hip -o mfma_valu_inwave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

enum { M16 = 0, M32 = 1, MBF = 2, NONE = 3 };
enum { F_FMA = 0, F_MAXI = 1, F_PKFMA = 2, F_BFE = 3, F_DSR = 4 };

template <int MK, int K, int FK>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[1024];
    lds[threadIdx.x & 1023] = threadIdx.x;
    __syncthreads();
    f32x4 a4[4];
    f32x16 a16[2];
    for (int i = 0; i < 4; ++i) a4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) a16[i][j] = 0.f;
    float a = 1.f + threadIdx.x, b = 2.f + threadIdx.x * 0.5f;
    bf16x8 ha = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, hb = {8, 7, 6, 5, 4, 3, 2, 1};
    float x[8];
    unsigned u[8];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p[8];
    for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x + j; u[j] = threadIdx.x * 7 + j; p[j] = f32x2{x[j], x[j] + 1.f}; }
    const float m = 1.0001f, c = 0.5f;
    const f32x2 pm = {m, m}, pc = {c, c};
    int addr = (threadIdx.x & 63) * 4;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (MK == M16) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a4[s]) : "v"(a), "v"(b));
            if (MK == M32) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a16[s & 1]) : "v"(a), "v"(b));
            if (MK == MBF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4[s]) : "v"(ha), "v"(hb));
#pragma unroll
            for (int f = 0; f < K; ++f) {
                const int j = (s * K + f) & 7;
                if (FK == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
                if (FK == F_MAXI) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[j]) : "v"(addr));
                if (FK == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pm), "v"(pc));
                if (FK == F_BFE) asm volatile("v_bfe_i32 %0, %1, 3, 1" : "=v"(u[j]) : "v"(addr));
                if (FK == F_DSR) asm volatile("ds_read_b32 %0, %1" : "=v"(x[j]) : "v"(addr));
            }
        }
        if (FK == F_DSR) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int j = 0; j < 8; ++j) r += x[j] + (float)u[j] + p[j][0] + p[j][1];
    for (int i = 0; i < 4; ++i) r += a4[i][0];
    r += a16[0][0] + a16[1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + lds[threadIdx.x & 1023];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MK, int K, int FK>
void run(float* out, unsigned long long* cyc, int wps, const char* mname, const char* fname) {
    const int iters = 20000;
    hipLaunchKernelGGL((k<MK, K, FK>), dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MK, K, FK>), dim3(256), dim3(256 * wps), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // per SIMD: wps waves each issue 4 * iters slots in h cycles
    const double per_slot_wave = (double)h / (4.0 * iters);
    printf("%-6s %-8s k=%d wps=%d : %7.2f cyc per slot per wave | %7.2f cyc per slot per SIMD | %.3f ms (%.2f GHz)\n", mname,
           fname, K, wps, per_slot_wave, per_slot_wave / wps, ms, (double)h / (ms * 1e6));
}

#define SWEEP(MK, FK, MN, FN)                                                                          \
    for (int wps = 1; wps <= 2; ++wps) {                                                               \
        run<MK, 0, FK>(out, cyc, wps, MN, FN); run<MK, 1, FK>(out, cyc, wps, MN, FN);                    \
        run<MK, 2, FK>(out, cyc, wps, MN, FN); run<MK, 3, FK>(out, cyc, wps, MN, FN);                    \
        run<MK, 4, FK>(out, cyc, wps, MN, FN); run<MK, 6, FK>(out, cyc, wps, MN, FN);                    \
        run<MK, 8, FK>(out, cyc, wps, MN, FN);                                                          \
    }

int main() {
    float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
    unsigned long long* cyc; hipMalloc(&cyc, 8);
    SWEEP(M16, F_FMA, "f32x16", "v_fma")
    SWEEP(M32, F_FMA, "f32x32", "v_fma")
    SWEEP(MBF, F_FMA, "bf16", "v_fma")
    SWEEP(NONE, F_FMA, "none", "v_fma")
    SWEEP(M16, F_MAXI, "f32x16", "v_max_i")
    SWEEP(M16, F_PKFMA, "f32x16", "v_pk_fma")
    SWEEP(M16, F_BFE, "f32x16", "v_bfe")
    SWEEP(M16, F_DSR, "f32x16", "ds_read")
    SWEEP(NONE, F_PKFMA, "none", "v_pk_fma")
    return 0;
}
