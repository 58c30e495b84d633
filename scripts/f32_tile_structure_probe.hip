// Structural A/B for the f32 decoder tile (round-4 verdict, item 1a / 1c), on the MLP alone: the 32-wide ResNet layers of the decoder
// (15 layers in LDS, x <- W relu(x) + x0 chained), the same points per CU, three structures:
//   A  8 waves (2 per SIMD, 256 VGPRs), two 16-point sub-tiles per wave, v_mfma_f32_16x16x4_f32           - today's tile
//   B  4 waves (1 per SIMD, 512 VGPRs), four 16-point sub-tiles per wave, v_mfma_f32_16x16x4_f32          - verdict 1a
//   C  8 waves, one 32-point tile per wave on v_mfma_f32_32x32x2_f32 (one M-tile covers all 32 channels) - verdict 1c
//   D  4 waves, two 32-point tiles per wave on v_mfma_f32_32x32x2_f32                                     - 1a + 1c
// Epilogue per value: ReLU (v_max) + a sign-mask word per four values, like the tile.  hipcc schedules each body.
// Reported: shader cycles per layer and 16 points per SIMD (floor: 16 MFMAs x 32 cycles = 512), f32 TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 f32_tile_structure_probe.hip -o f32_tile_structure_probe && ./f32_tile_structure_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NL = 15;
constexpr int LAYER = 16 * 64;      // floats per layer in fragment order (both MFMA shapes: 16 A registers per lane)

template <int NSUB, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void k16(const float* __restrict__ img, const float* __restrict__ xin, float* __restrict__ yout,
                                                               unsigned long long* cyc, unsigned int* msk, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NL * LAYER / 4; i += THREADS) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(img)[i];
    __syncthreads();
    float x[NSUB][8], x0[NSUB][8];
    const size_t p0 = ((size_t)(blockIdx.x * (THREADS / 64) + wave) * NSUB) * 16 * 8 * 4 + lane * 8;
    for (int t = 0; t < NSUB; ++t)
        for (int e = 0; e < 8; ++e) x0[t][e] = x[t][e] = xin[(p0 + t * 512 + e) & 0xfffff];
    unsigned int macc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int l = 0; l < NL; ++l) {
            const float* wl = lds + l * LAYER + lane;
            float a[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = wl[i * 64];
#pragma unroll
            for (int t = 0; t < NSUB; ++t) {
                float r[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = fmaxf(x[t][e], 0.f);
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    macc += __builtin_amdgcn_perm(__float_as_uint(x[t][4 * h]), __float_as_uint(x[t][4 * h + 1]), 0x0b090c0cu) ^
                            __builtin_amdgcn_perm(__float_as_uint(x[t][4 * h + 2]), __float_as_uint(x[t][4 * h + 3]), 0x0c0c0b09u);
                f32x4 c0 = {x0[t][0], x0[t][1], x0[t][2], x0[t][3]}, c1 = {x0[t][4], x0[t][5], x0[t][6], x0[t][7]};
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * s], r[s], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * s + 1], r[s], c1, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[t][e] = c0[e]; x[t][4 + e] = c1[e]; }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int t = 0; t < NSUB; ++t)
        for (int e = 0; e < 8; ++e) sum += x[t][e];
    yout[blockIdx.x * THREADS + threadIdx.x] = sum;
    msk[blockIdx.x * THREADS + threadIdx.x] = macc;
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);        // the LONGEST wave: the older wave of a SIMD finishes early
}

// 32-point tiles on 32x32x2: lane (n = lane & 31, half = lane >> 5) holds 16 channels of point n; k-step s takes register s
template <int NT, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void k32(const float* __restrict__ img, const float* __restrict__ xin, float* __restrict__ yout,
                                                               unsigned long long* cyc, unsigned int* msk, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NL * LAYER / 4; i += THREADS) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(img)[i];
    __syncthreads();
    f32x16 x[NT], x0[NT];
    const size_t p0 = ((size_t)(blockIdx.x * (THREADS / 64) + wave) * NT) * 1024 + lane * 16;
    for (int t = 0; t < NT; ++t)
        for (int e = 0; e < 16; ++e) x0[t][e] = x[t][e] = xin[(p0 + t * 1024 + e) & 0xfffff];
    unsigned int macc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int l = 0; l < NL; ++l) {
            const float* wl = lds + l * LAYER + lane;
            float a[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = wl[i * 64];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float r[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) r[e] = fmaxf(x[t][e], 0.f);
#pragma unroll
                for (int h = 0; h < 4; ++h)
                    macc += __builtin_amdgcn_perm(__float_as_uint(x[t][4 * h]), __float_as_uint(x[t][4 * h + 1]), 0x0b090c0cu) ^
                            __builtin_amdgcn_perm(__float_as_uint(x[t][4 * h + 2]), __float_as_uint(x[t][4 * h + 3]), 0x0c0c0b09u);
                f32x16 c = x0[t];
#pragma unroll
                for (int s = 0; s < 16; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], r[s], c, 0, 0, 0);
                x[t] = c;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int t = 0; t < NT; ++t)
        for (int e = 0; e < 16; ++e) sum += x[t][e];
    yout[blockIdx.x * THREADS + threadIdx.x] = sum;
    msk[blockIdx.x * THREADS + threadIdx.x] = macc;
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);        // the LONGEST wave: the older wave of a SIMD finishes early
}

template <typename K>
void run(const char* name, K kern, int threads, int pts_per_wave, const float* dimg, const float* dx, float* dy, unsigned long long* dc, unsigned int* dm) {
    const int iters = 400, blocks = 256;
    const size_t shm = 100 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(dc, 0, 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), shm, 0, dimg, dx, dy, dc, dm, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    const double pts_per_simd = (double)pts_per_wave * (threads / 256);            // waves per SIMD x points per wave
    const double per16 = (double)cyc / (iters * NL) / (pts_per_simd / 16.0);
    const double flop = 2.0 * 32 * 32 * blocks * (threads / 64) * (double)pts_per_wave * iters * NL;
    printf("%-72s %7.3f ms  %6.1f TFLOP/s  %6.0f cycles per layer and 16 points per SIMD (floor 512)  clock while it ran %.2f GHz\n", name, ms,
           flop / ms / 1e9, per16, (double)cyc / (ms * 1e-3) / 1e9);
}

int main() {
    std::vector<float> W((size_t)NL * LAYER), X(1 << 20);
    srand(1);
    for (auto& w : W) w = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.2f;
    for (auto& v : X) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *dimg, *dx, *dy;
    unsigned long long* dc;
    unsigned int* dm;
    hipMalloc(&dimg, W.size() * 4); hipMalloc(&dx, X.size() * 4); hipMalloc(&dy, 256 * 512 * 4); hipMalloc(&dc, 64); hipMalloc(&dm, 256 * 512 * 4);
    hipMemcpy(dimg, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    for (int zero = 0; zero < 2; ++zero) {
    if (zero) {
        printf("-- the same with all-zero weights and inputs (no switching activity in the multipliers: what the clock does)\n");
        hipMemset(dimg, 0, W.size() * 4);
        hipMemset(dx, 0, X.size() * 4);
    }
    run("A  8 waves x two 16-point sub-tiles, 16x16x4 (today)", k16<2, 512>, 512, 32, dimg, dx, dy, dc, dm);
    run("B  4 waves x four 16-point sub-tiles, 16x16x4 (one wave per SIMD)", k16<4, 256>, 256, 64, dimg, dx, dy, dc, dm);
    run("C  8 waves x one 32-point tile, 32x32x2", k32<1, 512>, 512, 32, dimg, dx, dy, dc, dm);
    run("D  4 waves x two 32-point tiles, 32x32x2 (one wave per SIMD)", k32<2, 256>, 256, 64, dimg, dx, dy, dc, dm);
    run("E  8 waves x two 32-point tiles, 32x32x2", k32<2, 512>, 512, 64, dimg, dx, dy, dc, dm);
    }
    return 0;
}
