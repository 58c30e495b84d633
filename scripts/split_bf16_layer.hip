// Probe for the split-precision MFMA mode SURVEY 8f lists under N4 (not part of the product): one 256x256 layer of the
// ONet decoder on 16 points per wave, weights read from LDS in fragment order like onet.hip,
//   (a) f32:     1024 x v_mfma_f32_16x16x4_f32
//   (b) bf16x3:  the activations and the weights split into hi + lo bf16 parts (x = hi + lo + O(2^-17 x)),
//                384 x v_mfma_f32_16x16x32_bf16 for hi*hi + hi*lo + lo*hi, f32 accumulation
// and the error of both against a float64 reference.  hipcc --offload-arch=gfx950 -O3 split_bf16_layer.hip -o split_bf16_layer
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int H = 256;
constexpr int TT = 8;                 // output tiles resident in LDS (8 x 16 rows x 256 x 4 B = 128 KB); the timing loop runs them twice
constexpr int IMG_BYTES = TT * 16 * H * 4;

__device__ __host__ inline u16 f2bf(float f) {        // round to nearest even
    unsigned int u = __builtin_bit_cast(unsigned int, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__device__ __host__ inline float bf2f(u16 h) { return __builtin_bit_cast(float, (unsigned int)h << 16); }

// ---- (a) f32: fragment image [t 16][s4 16][lane 64][4], k-step s = 4 s4 + j multiplies channel 16 (s >> 2) + 4 q + (s & 3)
__global__ __launch_bounds__(512, 2) void layer_f32(const float* __restrict__ img, const float* __restrict__ x, float* __restrict__ y,
                                                     int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    for (int i = threadIdx.x; i < IMG_BYTES / 16; i += blockDim.x) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(img)[i];
    __syncthreads();
    float u[64];
    const float* xp = x + ((size_t)(blockIdx.x * 8 + wave) * 16 + n) * H;
    for (int t = 0; t < 16; ++t)
        for (int r = 0; r < 4; ++r) u[4 * t + r] = xp[16 * t + 4 * q + r];
    f32x4 acc[16];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4* a = reinterpret_cast<const f32x4*>(lds) + (size_t)(t % TT) * 16 * 64 + lane;
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                const f32x4 av = a[s4 * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], u[4 * s4 + j], acc[t], 0, 0, 0);
            }
        }
        if (it + 1 < iters)
            for (int t = 0; t < 16; ++t)
                for (int r = 0; r < 4; ++r) u[4 * t + r] = acc[t][r] * 0.05f + u[4 * t + r] * 0.5f;      // keep the chain alive
    }
    float* yp = y + ((size_t)(blockIdx.x * 8 + wave) * 16 + n) * H;
    for (int t = 0; t < 16; ++t)
        for (int r = 0; r < 4; ++r) yp[16 * t + 4 * q + r] = acc[t][r];
}

// ---- (b) bf16x3: fragment image [t 16][s 8][hi, lo][lane 64][8 bf16]; k-step s, slot j of lane group q multiplies
//      channel 16 (2 s + j / 4) + 4 q + (j % 4)
__global__ __launch_bounds__(512, 2) void layer_bf16x3(const u16* __restrict__ img, const float* __restrict__ x, float* __restrict__ y,
                                                        int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    for (int i = threadIdx.x; i < IMG_BYTES / 16; i += blockDim.x) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(img)[i];
    __syncthreads();
    float u[64];
    const float* xp = x + ((size_t)(blockIdx.x * 8 + wave) * 16 + n) * H;
    for (int t = 0; t < 16; ++t)
        for (int r = 0; r < 4; ++r) u[4 * t + r] = xp[16 * t + 4 * q + r];
    f32x4 acc[16];
    for (int it = 0; it < iters; ++it) {
        bf16x8 uh[8], ul[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = u[8 * s + j];                      // register (t = 2 s + j / 4, r = j % 4)
                const u16 h = f2bf(v);
                const u16 l = f2bf(v - bf2f(h));
                uh[s][j] = __builtin_bit_cast(__bf16, h);
                ul[s][j] = __builtin_bit_cast(__bf16, l);
            }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bf16x8* a = reinterpret_cast<const bf16x8*>(lds) + (size_t)(t % TT) * 8 * 2 * 64 + lane;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bf16x8 ah = a[(2 * s) * 64], al = a[(2 * s + 1) * 64];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, uh[s], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ul[s], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, uh[s], acc[t], 0, 0, 0);
            }
        }
        if (it + 1 < iters)
            for (int t = 0; t < 16; ++t)
                for (int r = 0; r < 4; ++r) u[4 * t + r] = acc[t][r] * 0.05f + u[4 * t + r] * 0.5f;
    }
    float* yp = y + ((size_t)(blockIdx.x * 8 + wave) * 16 + n) * H;
    for (int t = 0; t < 16; ++t)
        for (int r = 0; r < 4; ++r) yp[16 * t + 4 * q + r] = acc[t][r];
}

int main() {
    const int blocks = 256, P = blocks * 8 * 16;
    std::vector<float> W(H * H), X((size_t)P * H);
    srand(1);
    for (auto& w : W) w = ((float)rand() / RAND_MAX * 2.f - 1.f) / 16.f;
    for (auto& v : X) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    std::vector<float> img32(H * H);
    std::vector<u16> img16((size_t)H * H * 2);
    for (int t = 0; t < 16; ++t)
        for (int lane = 0; lane < 64; ++lane) {
            const int m = 16 * t + (lane & 15), q = lane >> 4;
            for (int s4 = 0; s4 < 16; ++s4)
                for (int j = 0; j < 4; ++j) {
                    const int s = 4 * s4 + j, k = 16 * (s >> 2) + 4 * q + (s & 3);
                    img32[(((size_t)t * 16 + s4) * 64 + lane) * 4 + j] = W[m * H + k];
                }
            for (int s = 0; s < 8; ++s)
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * (2 * s + j / 4) + 4 * q + (j % 4);
                    const float w = W[m * H + k];
                    const u16 h = f2bf(w), l = f2bf(w - bf2f(h));
                    img16[((((size_t)t * 8 + s) * 2 + 0) * 64 + lane) * 8 + j] = h;
                    img16[((((size_t)t * 8 + s) * 2 + 1) * 64 + lane) * 8 + j] = l;
                }
        }
    float *dimg32, *dx, *dy;
    u16* dimg16;
    hipMalloc(&dimg32, H * H * 4); hipMalloc(&dimg16, H * H * 4); hipMalloc(&dx, (size_t)P * H * 4); hipMalloc(&dy, (size_t)P * H * 4);
    hipMemcpy(dimg32, img32.data(), H * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(dimg16, img16.data(), H * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, X.data(), (size_t)P * H * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(layer_f32), hipFuncAttributeMaxDynamicSharedMemorySize, IMG_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(layer_bf16x3), hipFuncAttributeMaxDynamicSharedMemorySize, IMG_BYTES);
    std::vector<float> Y((size_t)P * H);
    // accuracy (one application) against float64
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(layer_f32, dim3(blocks), dim3(512), IMG_BYTES, 0, dimg32, dx, dy, 1);
        else hipLaunchKernelGGL(layer_bf16x3, dim3(blocks), dim3(512), IMG_BYTES, 0, dimg16, dx, dy, 1);
        hipMemcpy(Y.data(), dy, (size_t)P * H * 4, hipMemcpyDeviceToHost);
        double emax = 0, esum = 0, ref_rms = 0;
        const int np = 256;
        for (int p = 0; p < np; ++p)
            for (int m = 0; m < TT * 16; ++m) {
                double r = 0;
                for (int k = 0; k < H; ++k) r += (double)W[m * H + k] * (double)X[(size_t)p * H + k];
                const double e = fabs((double)Y[(size_t)p * H + m] - r);
                emax = e > emax ? e : emax; esum += e; ref_rms += r * r;
            }
        ref_rms = sqrt(ref_rms / (np * TT * 16));
        printf("%-8s one layer: max |err| %.3e, mean |err| %.3e  (rms of the outputs %.3f -> relative %.2e)\n", mode ? "bf16x3" : "f32", emax,
               esum / (np * TT * 16), ref_rms, esum / (np * TT * 16) / ref_rms);
    }
    // speed: 200 dependent applications
    const int iters = 200;
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(layer_f32, dim3(blocks), dim3(512), IMG_BYTES, 0, dimg32, dx, dy, iters);
            else hipLaunchKernelGGL(layer_bf16x3, dim3(blocks), dim3(512), IMG_BYTES, 0, dimg16, dx, dy, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * H * H * (double)P * iters;
        printf("%-8s %d layers x %d points: %.2f ms -> %.1f f32-equivalent TFLOP/s (%.2f us per layer per CU)\n", mode ? "bf16x3" : "f32", iters,
               P, ms, flop / ms / 1e9, ms * 1e3 / iters);
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
    return 0;
}
