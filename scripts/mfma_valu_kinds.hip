// Which kinds of vector instructions of one wave delay an f32 MFMA stream of another wave on the same SIMD (gfx950)?
//   waves 0-3 (one per SIMD): 16x16x4 f32 MFMA loop; waves 4-7: a stream of one instruction kind.
//   t(both) ~ max  -> the kind issues beside the matrix pipe;  t(both) ~ sum -> it takes the pipe's slots.
// hipcc --offload-arch=gfx950 -O3 mfma_valu_kinds.hip -o mfma_valu_kinds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { FMA = 0, INT = 1, CMPSEL = 2, PKFMA = 3, MAXI = 4, LDSR = 5, MOV = 6, NKIND = 7 };
static const char* NAMES[NKIND] = {"v_fma_f32", "v_xor/v_add_u32", "v_cmp+v_cndmask", "v_pk_fma_f32", "v_max_i32", "ds_read_b32", "v_mov_b32"};

template <int KIND, int NOPS = 0>
__global__ __launch_bounds__(512) void k(float* out, int mode, int mfma_iters, int valu_iters) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
    __syncthreads();
    float r = 0.f;
    if (wave < 4) {
        if (mode == 0 || mode == 2) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            float a = 1.f + threadIdx.x, b = 2.f + threadIdx.x * 0.5f;
            for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0);
                    if (NOPS >= 1) asm volatile("s_nop %0" :: "n"(NOPS - 1));
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a1, 0, 0, 0);
                    if (NOPS >= 1) asm volatile("s_nop %0" :: "n"(NOPS - 1));
                }
            }
            r = a0[0] + a1[1];
        }
    } else if (mode == 1 || mode == 2) {
        float x[8];
        unsigned u[8];
        f32x2 p[8];
        for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x + j; u[j] = threadIdx.x * 7 + j; p[j] = f32x2{x[j], x[j] + 1.f}; }
        const float m = 1.0001f, c = 0.5f;
        const f32x2 pm = {m, m}, pc = {c, c};
        int addr = (threadIdx.x & 63) * 4;
        for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
                    if (KIND == INT) { if (rep & 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7])); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(addr)); }
                    if (KIND == CMPSEL) { if (rep & 1) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[j]), "v"(c) : "vcc"); else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(m) : "vcc"); }
                    if (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pm), "v"(pc));
                    if (KIND == MAXI) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[j]) : "v"(addr));
                    if (KIND == LDSR) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(addr)); x[j] = t; }
                    if (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[j]) : "v"(addr));
                }
            if (KIND == LDSR) asm volatile("s_waitcnt lgkmcnt(0)");
        }
        for (int j = 0; j < 8; ++j) r += x[j] + (float)u[j] + p[j][0] + p[j][1];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + lds[threadIdx.x];
}
template <int KIND, int NOPS = 0>
float run(float* out, int mode, int mi, int vi) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NOPS>), dim3(256), dim3(512), 0, 0, out, mode, mi, vi); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NOPS>), dim3(256), dim3(512), 0, 0, out, mode, mi, vi);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int KIND>
void test(float* out) {
    const int mi = 20000, vi = 40000;        // 16 MFMA x 32 clk per iteration ; 32 instructions per iteration
    const float a = run<KIND>(out, 0, mi, vi), b = run<KIND>(out, 1, mi, vi), c = run<KIND>(out, 2, mi, vi);
    printf("%-16s mfma-only %.3f ms | kind-only %.3f ms (%.2f clk/instr at 2.4 GHz) | both %.3f ms  -> MFMA delayed by %.0f %% of the kind's time\n",
           NAMES[KIND], a, b, b * 2.4e6 / (vi * 32.0), c, (c - a) / b * 100.f);
}
template <int KIND, int NOPS>
void test_spaced(float* out) {
    const int mi = 20000, vi = 40000;
    const float a = run<KIND, NOPS>(out, 0, mi, vi), b = run<KIND, NOPS>(out, 1, mi, vi), c = run<KIND, NOPS>(out, 2, mi, vi);
    printf("%-16s each MFMA followed by s_nop %d: mfma-only %.3f ms | kind-only %.3f ms | both %.3f ms  -> MFMA delayed by %.0f %% of the kind's time\n",
           NAMES[KIND], NOPS - 1, a, b, c, (c - a) / b * 100.f);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
    test_spaced<FMA, 1>(out); test_spaced<FMA, 2>(out); test_spaced<FMA, 3>(out); test_spaced<FMA, 4>(out); test_spaced<FMA, 5>(out);
    test_spaced<FMA, 6>(out); test_spaced<FMA, 7>(out); test_spaced<FMA, 8>(out);
    test_spaced<LDSR, 4>(out); test_spaced<CMPSEL, 4>(out);
    test<FMA>(out); test<INT>(out); test<CMPSEL>(out); test<PKFMA>(out); test<MAXI>(out); test<LDSR>(out); test<MOV>(out);
    return 0;
}
