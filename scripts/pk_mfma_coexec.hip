// Do packed-f32 vector instructions compute correctly while the OTHER wave of their SIMD streams bf16 MFMAs - and how many wait
// states does a packed-f32 result need in front of a dependent vector instruction?
// 512-thread workgroups (two waves per SIMD): waves 0-3 run a dependent chain of eight v_pk_mul_f32 / v_pk_fma_f32 (the op_sel forms
// hipcc generates for broadcast weights) with N wait states between producer and consumer (N = 0: back to back; hipcc pads such a
// pair with ONE: s_nop 0), then the same chain in scalar v_mul / v_fma; waves 4-7 idle, stream f32 MFMAs or stream bf16 MFMAs.
// Counts results that differ between the packed and the scalar chain.
//   hipcc --offload-arch=gfx950 -O3 pk_mfma_coexec.hip -o pk_mfma_coexec && ./pk_mfma_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define NOPS0 ""
#define NOPS1 "s_nop 0\n\t"
#define NOPS2 "s_nop 1\n\t"
#define NOPS3 "s_nop 2\n\t"
#define CHAIN(P)                                                                                            \
    asm volatile("s_nop 3\n\t"                                                                              \
                 "v_pk_mul_f32 %0, %1, %9 op_sel:[0,1]\n\t" P                                               \
                 "v_pk_fma_f32 %0, %2, %9, %0 op_sel_hi:[1,0,1]\n\t" P                                      \
                 "v_pk_fma_f32 %0, %3, %9, %0 op_sel:[0,1,0]\n\t" P                                         \
                 "v_pk_fma_f32 %0, %4, %9, %0 op_sel_hi:[1,0,1]\n\t" P                                      \
                 "v_pk_fma_f32 %0, %5, %9, %0 op_sel:[0,1,0]\n\t" P                                         \
                 "v_pk_fma_f32 %0, %6, %9, %0 op_sel_hi:[1,0,1]\n\t" P                                      \
                 "v_pk_fma_f32 %0, %7, %9, %0 op_sel:[0,1,0]\n\t" P                                         \
                 "v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[1,0,1]\n\t"                                        \
                 "s_nop 7"                                                                                  \
                 : "=&v"(s) : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]), "v"(ww))

template <int PARTNER, int N>
__global__ __launch_bounds__(512, 2) void k(const float* in, int* bad, float* sink, int iters) {
    const int wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 256 + (threadIdx.x & 255);
    if (wave >= 4) {
        if (PARTNER == 0) return;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (j + 1)); }
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const float fa = 0.001f * threadIdx.x, fb = 0.5f;
        for (int it = 0; it < iters * 8; ++it) {
            if (PARTNER == 1) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c1, 0, 0, 0);
            }
        }
        sink[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
        return;
    }
    f32x2 t[8];
    for (int j = 0; j < 8; ++j) t[j] = f32x2{in[(i * 16 + 2 * j) & 0xffff], in[(i * 16 + 2 * j + 1) & 0xffff]};
    const float w0 = 0.3f + in[i & 0xffff] * 0.1f, w1 = 0.7f - in[(i + 7) & 0xffff] * 0.1f;
    int nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const f32x2 ww = {w0 * w1, w1 * w1};
        f32x2 s;
        if (N == 0) CHAIN(NOPS0);
        if (N == 1) CHAIN(NOPS1);
        if (N == 2) CHAIN(NOPS2);
        if (N == 3) CHAIN(NOPS3);
        float rx = t[0].x * ww.y, ry = t[0].y * ww.y;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            const float w = (j & 1) ? ww.x : ww.y;
            rx = __builtin_fmaf(t[j].x, w, rx);
            ry = __builtin_fmaf(t[j].y, w, ry);
        }
        nbad += (s.x != rx) + (s.y != ry);
        t[it & 7].x += 1e-3f;      // keep the loop from being hoisted
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int PARTNER, int N>
void run(const char* name, const float* din, int* dbad, float* dsink) {
    hipMemset(dbad, 0, 4);
    const int iters = 40000;
    hipLaunchKernelGGL((k<PARTNER, N>), dim3(256), dim3(512), 0, 0, din, dbad, dsink, iters);
    int bad = 0;
    hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    printf("%-78s wrong: %7d of %lld\n", name, bad, 256ll * 256 * iters * 2);
}

int main() {
    std::vector<float> h(65536);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.25f + (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    float *din, *dsink;
    int* dbad;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dsink, 256 * 512 * 4); hipMalloc(&dbad, 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 0>("partner idle      | dependent packed ops back to back", din, dbad, dsink);
    run<0, 1>("partner idle      | 1 wait state between (what hipcc pads)", din, dbad, dsink);
    run<2, 0>("partner f32 MFMA  | back to back", din, dbad, dsink);
    run<2, 1>("partner f32 MFMA  | 1 wait state", din, dbad, dsink);
    run<1, 0>("partner bf16 MFMA | back to back", din, dbad, dsink);
    run<1, 1>("partner bf16 MFMA | 1 wait state (what hipcc pads)", din, dbad, dsink);
    run<1, 2>("partner bf16 MFMA | 2 wait states", din, dbad, dsink);
    run<1, 3>("partner bf16 MFMA | 3 wait states", din, dbad, dsink);
    return 0;
}
