// LDS gather throughput on gfx950: 8 waves of a CU reading random 16- / 8- / 4-byte elements of a 16 KB array (the kNN
// phase's access pattern: every lane its own candidate point).  Prints LDS-pipe cycles per wave-level read instruction.
//   hipcc --offload-arch=gfx950 -O3 lds_gather_rate.hip -o lds_gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int BYTES, bool RANDOM>
__global__ __launch_bounds__(512) void k(const int* __restrict__ idx, float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float X[1025 * 4];
    for (int i = threadIdx.x; i < 1025 * 4; i += 512) X[i] = (float)i;
    __syncthreads();
    int j[16];
    for (int e = 0; e < 16; ++e) j[e] = RANDOM ? idx[(threadIdx.x * 16 + e) % 8192] : ((threadIdx.x + e * 64) & 1023);
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (BYTES == 16) { const f32x4 v = *reinterpret_cast<const f32x4*>(X + 4 * j[e]); acc += v.x + v.y + v.z + v.w; }
            if (BYTES == 8) { const f32x2 v = *reinterpret_cast<const f32x2*>(X + 2 * j[e]); acc += v.x + v.y; }
            if (BYTES == 4) { acc += X[j[e]] + X[1024 + j[e]] + X[2048 + j[e]]; }
            j[e] = (j[e] + (RANDOM ? 37 : 64)) & 1023;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int BYTES, bool RANDOM>
void run(const char* name, const int* d_idx, float* d_out, unsigned long long* d_cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<BYTES, RANDOM>), dim3(256), dim3(512), 0, 0, d_idx, d_out, iters, d_cyc);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    const double reads = (double)iters * 16 * 8 * (BYTES == 4 ? 3 : 1);   // wave-level read instructions per CU
    printf("%-34s %8.1f cycles per wave-level read (8 waves/CU), %6.1f per gathered point\n", name, c / reads,
           c / ((double)iters * 16 * 8));
}

int main() {
    std::vector<int> h(8192);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 10) & 1023; }
    int* d_idx; float* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_idx, 8192 * 4); hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_cyc, 8);
    hipMemcpy(d_idx, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    run<16, true>("ds_read_b128, random point", d_idx, d_out, d_cyc);
    run<16, false>("ds_read_b128, consecutive points", d_idx, d_out, d_cyc);
    run<8, true>("ds_read_b64, random element", d_idx, d_out, d_cyc);
    run<8, false>("ds_read_b64, consecutive", d_idx, d_out, d_cyc);
    run<4, true>("3 x ds_read_b32 (SoA), random", d_idx, d_out, d_cyc);
    run<4, false>("3 x ds_read_b32 (SoA), consecutive", d_idx, d_out, d_cyc);
    return 0;
}
