// Probe for the split-precision decoder tile (round 5): the 32-wide ResNet layers of the ConvONet decoder
// (ConvONet/src/conv_onet/models/decoder.py:83-93) on the bf16 matrix core with f32-exact operand splits.
//
//   x = x1 + x2 + x3, w = w1 + w2 + w3 (bf16 pieces, round to nearest: 24+ mantissa bits together)
//   w x ~= w1 x1 + w1 x2 + w2 x1 + w2 x2 + w1 x3 + w3 x1        ("bf16x6": dropped terms <= 2^-26 |w x|)
//   w x ~= w1 x1 + w1 x2 + w2 x1                                ("bf16x3": 2^-17 |w x|)
//
// Rate question: an f32 MFMA excludes every vector instruction on its SIMD (profiles/r04_pmc_fifo.txt), a bf16 MFMA does
// not - so does the tile's time become max(matrix, vector) instead of their sum?  The splits themselves are made on the
// matrix pipe where possible: r = relu(x) - x1 is ONE v_mfma_f32_16x16x32_bf16 with A = -(selection matrix) and C = relu(x)
// (exact: x1 is within 2^-9 of C), so the vector pipe only converts (v_cvt_pk_bf16_f32, 12 per 8 values).
//
// Modes (template): 0 = f32 v_mfma_f32_16x16x4_f32 (today's tile), 1 = bf16x6 with matrix-pipe residuals,
//                   2 = bf16x6 with vector-pipe residuals, 3 = bf16x3.
// Harness: 512-thread workgroups (2 waves per SIMD), one per CU, two 16-point sub-tiles per wave, 15 layers in LDS,
// x <- relu(W x + x0) chained; shader cycles per layer and sub-tile; accuracy of one layer against float64.
//   hipcc --offload-arch=gfx950 -O3 bf16x6_tile_probe.hip -o bf16x6_tile_probe && ./bf16x6_tile_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int NL = 15;                 // layers resident in LDS
constexpr int F32_LAYER = 16 * 64;     // floats: [s 8][mt 2][lane 64]
constexpr int BF_LAYER = 3 * 2 * 64 * 8;   // u16: [split 3][mt 2][lane 64][8]

__host__ inline u16 f2bf(float f) {
    unsigned int u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__host__ inline float bf2f(u16 h) {
    unsigned int u = (unsigned int)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// channel of accumulator register e = 4 mt + r of lane group g (= B-operand element e of the next layer)
__host__ __device__ constexpr int chan(int g, int e) { return 16 * (e >> 2) + 4 * g + (e & 3); }

struct Split3 {
    bf16x8 p[3];
};

// pieces of the eight values of a lane; VRES: residuals on the vector pipe
__device__ __forceinline__ bf16x8 cvt8(const f32x8& v) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
    return o;
}
__device__ __forceinline__ f32x8 sub8(const f32x8& v, const bf16x8& p) {
    f32x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v[j] - (float)p[j];
    return o;
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void tile_probe(const void* __restrict__ img, const float* __restrict__ x_in,
                                                      float* __restrict__ y_out, unsigned long long* __restrict__ cyc, int iters,
                                                      int one_layer) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    constexpr int IMG_BYTES = MODE == 0 ? NL * F32_LAYER * 4 : NL * BF_LAYER * 2;
    for (int i = threadIdx.x; i < IMG_BYTES / 16; i += blockDim.x)
        reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(img)[i];
    __syncthreads();
    // residual selection matrices: lane (m, g), element j of M-tile mt is -1 iff chan(g, j) == 16 mt + m
    bf16x8 sel[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 8; ++j) sel[mt][j] = (__bf16)((chan(g, j) == 16 * mt + (lane & 15)) ? -1.0f : 0.0f);
    f32x8 x[2], x0[2];
    const size_t pt0 = ((size_t)(blockIdx.x * 8 + wave) * 2) * 16 + (lane & 15);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) x0[t][e] = x[t][e] = x_in[(pt0 + 16 * t) * 32 + chan(g, e)];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int l = 0; l < (one_layer ? 1 : NL); ++l) {
            f32x4 acc[2][2];
            if (MODE == 0) {
                const float* wl = reinterpret_cast<const float*>(lds) + l * F32_LAYER + lane;
                float a[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = wl[i * 64];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x8 r;
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = one_layer ? x[t][e] : fmaxf(x[t][e], 0.f);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[t][mt] = f32x4{x0[t][4 * mt], x0[t][4 * mt + 1], x0[t][4 * mt + 2], x0[t][4 * mt + 3]};
#pragma unroll
                    for (int s = 0; s < 8; ++s)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * s + mt], r[s], acc[t][mt], 0, 0, 0);
                }
            } else {
                const bf16x8* wl = reinterpret_cast<const bf16x8*>(lds) + l * (BF_LAYER / 8) + lane;
                bf16x8 A[3][2];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) A[sp][mt] = wl[(sp * 2 + mt) * 64];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x8 r;
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = one_layer ? x[t][e] : fmaxf(x[t][e], 0.f);
                    bf16x8 p1 = cvt8(r), p2, p3;
                    if (MODE == 1) {
                        f32x4 c0 = {r[0], r[1], r[2], r[3]}, c1 = {r[4], r[5], r[6], r[7]};
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel[0], p1, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel[1], p1, c1, 0, 0, 0);
                        const f32x8 r1 = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                        p2 = cvt8(r1);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel[0], p2, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel[1], p2, c1, 0, 0, 0);
                        const f32x8 r2 = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                        p3 = cvt8(r2);
                    } else {
                        const f32x8 r1 = sub8(r, p1);
                        p2 = cvt8(r1);
                        if (MODE == 2) p3 = cvt8(sub8(r1, p2));
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        f32x4 c = {x0[t][4 * mt], x0[t][4 * mt + 1], x0[t][4 * mt + 2], x0[t][4 * mt + 3]};
                        if (MODE != 3) {     // small terms first
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2][mt], p1, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0][mt], p3, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1][mt], p2, c, 0, 0, 0);
                        }
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1][mt], p1, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0][mt], p2, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0][mt], p1, c, 0, 0, 0);
                        acc[t][mt] = c;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) x[t][e] = acc[t][e >> 2][e & 3];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) y_out[(pt0 + 16 * t) * 32 + chan(g, e)] = x[t][e];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, const void* dimg, size_t img_bytes, const float* dx, float* dy, unsigned long long* dcyc,
                const std::vector<float>& W, const std::vector<float>& X, int P) {
    const int blocks = 256;
    const size_t shm = 100 * 1024;          // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(tile_probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    // accuracy: layer 0 once, y = W x + x (no ReLU), against float64
    hipLaunchKernelGGL(tile_probe<MODE>, dim3(blocks), dim3(512), shm, 0, dimg, dx, dy, dcyc, 1, 1);
    std::vector<float> Y((size_t)P * 32);
    hipMemcpy(Y.data(), dy, Y.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, esum = 0, rms = 0, emax_rel = 0;
    const int np = 4096;
    for (int p = 0; p < np; ++p)
        for (int o = 0; o < 32; ++o) {
            double r = X[(size_t)p * 32 + o], mag = fabs(r);
            for (int c = 0; c < 32; ++c) {
                r += (double)W[o * 32 + c] * (double)X[(size_t)p * 32 + c];
                mag += fabs((double)W[o * 32 + c] * (double)X[(size_t)p * 32 + c]);
            }
            const double e = fabs((double)Y[(size_t)p * 32 + o] - r);
            emax = e > emax ? e : emax;
            emax_rel = e / mag > emax_rel ? e / mag : emax_rel;
            esum += e;
            rms += r * r;
        }
    rms = sqrt(rms / (np * 32));
    printf("%-28s one layer vs float64: max |err| %.3e  mean |err| %.3e  (output rms %.3f: mean relative %.2e; max err / sum|terms| %.2e)\n",
           name, emax, esum / (np * 32), rms, esum / (np * 32) / rms, emax_rel);
    // rate
    const int iters = 40;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(tile_probe<MODE>, dim3(blocks), dim3(512), shm, 0, dimg, dx, dy, dcyc, iters, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    {   // run-to-run determinism of the chained layers
        std::vector<float> Y1((size_t)P * 32), Y2((size_t)P * 32);
        hipMemcpy(Y1.data(), dy, Y1.size() * 4, hipMemcpyDeviceToHost);
        size_t nd = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(tile_probe<MODE>, dim3(blocks), dim3(512), shm, 0, dimg, dx, dy, dcyc, iters, 0);
            hipMemcpy(Y2.data(), dy, Y2.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < Y1.size(); ++i) nd += memcmp(&Y1[i], &Y2[i], 4) != 0;
        }
        printf("%-28s values that differ between repeated launches (5 repeats, %zu values each): %zu\n", name, Y1.size(), nd);
    }
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)cyc / (iters * NL * 2);        // per layer and sub-tile, one wave (two waves share the SIMD)
    const double flop = 2.0 * 32 * 32 * (double)P * iters * NL;
    printf("%-28s %d x %d layers, %d points: %.3f ms -> %.1f f32-equivalent TFLOP/s; %.0f shader cycles per layer and sub-tile per wave = %.0f per SIMD pair\n",
           name, iters, NL, P, ms, flop / ms / 1e9, per, per / 2 * 2);
}

int main() {
    const int blocks = 256, P = blocks * 8 * 2 * 16;
    std::vector<float> W((size_t)NL * 32 * 32), X((size_t)P * 32);
    srand(1);
    for (auto& w : W) w = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.25f;
    for (auto& v : X) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    std::vector<float> img32((size_t)NL * F32_LAYER);
    std::vector<u16> img16((size_t)NL * BF_LAYER);
    for (int l = 0; l < NL; ++l)
        for (int lane = 0; lane < 64; ++lane) {
            const int m = lane & 15, g = lane >> 4;
            // f32 16x16x4: k-step s multiplies register s of the B lane (n, q = g): channel chan(q, s); A lane (m, q): W[16 mt + m][chan(q, s)]
            for (int s = 0; s < 8; ++s)
                for (int mt = 0; mt < 2; ++mt)
                    img32[(size_t)l * F32_LAYER + (2 * s + mt) * 64 + lane] = W[(size_t)l * 1024 + (16 * mt + m) * 32 + chan(g, s)];
            for (int mt = 0; mt < 2; ++mt)
                for (int j = 0; j < 8; ++j) {
                    const float w = W[(size_t)l * 1024 + (16 * mt + m) * 32 + chan(g, j)];
                    const u16 h1 = f2bf(w);
                    const float r1 = w - bf2f(h1);
                    const u16 h2 = f2bf(r1);
                    const float r2 = r1 - bf2f(h2);
                    const u16 h3 = f2bf(r2);
                    const u16 hs[3] = {h1, h2, h3};
                    for (int sp = 0; sp < 3; ++sp) img16[(size_t)l * BF_LAYER + ((sp * 2 + mt) * 64 + lane) * 8 + j] = hs[sp];
                }
        }
    void *d32, *d16;
    float *dx, *dy;
    unsigned long long* dcyc;
    hipMalloc(&d32, img32.size() * 4);
    hipMalloc(&d16, img16.size() * 2);
    hipMalloc(&dx, X.size() * 4);
    hipMalloc(&dy, X.size() * 4);
    hipMalloc(&dcyc, 64);
    hipMemcpy(d32, img32.data(), img32.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d16, img16.data(), img16.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    run<0>("f32 16x16x4 (today)", d32, img32.size() * 4, dx, dy, dcyc, W, X, P);
    run<1>("bf16x6, matrix residuals", d16, img16.size() * 2, dx, dy, dcyc, W, X, P);
    run<2>("bf16x6, vector residuals", d16, img16.size() * 2, dx, dy, dcyc, W, X, P);
    run<3>("bf16x3", d16, img16.size() * 2, dx, dy, dcyc, W, X, P);
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
    return 0;
}
