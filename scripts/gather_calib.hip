// Calibration of rocprofv3's FETCH_SIZE for the decoder tile's gather pattern (verdict r01 #4): a kernel that reads a
// KNOWN number of bytes exactly like decoder_tile3 reads its taps - four lanes (q = 0..3) of a 16-lane point group
// take 16 B each at line + 16 q and at line + 64 + 16 q, i.e. one 128-byte line per (point, tap) through two
// buffer_load_dwordx4 wave instructions - from a buffer far larger than the 256 MB Infinity Cache, every line once.
//   hipcc --offload-arch=gfx950 -O3 scripts/gather_calib.hip -o scripts/gather_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o g -- scripts/gather_calib
// prints the bytes it read; scripts/collect_profiles.sh divides by FETCH_SIZE * 1024 and records the factor.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void gather_kernel(const float* __restrict__ buf, unsigned long long n_lines, int iters,
                                                     float* __restrict__ out) {
    const unsigned long long wave = (unsigned long long)blockIdx.x * 8 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        // line index of (wave, it, point n): a stride permutation of [0, n_lines) - every line exactly once, far apart
        const unsigned long long id = (wave * iters + it) * 16 + n;
        const unsigned long long line = (id * 2654435761ull) % n_lines;
        const float* p = buf + line * 32 + 4 * q;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + 16);
        acc += a + b;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    const unsigned long long n_lines = 16777259ull;            // prime > 2^24: 2.1 GB of 128-byte lines
    const int blocks = 2048, iters = 64;                        // 2048 * 8 waves * 64 iterations * 16 lines = 16.8 M lines: each once
    float *buf, *out;
    if (hipMalloc(&buf, n_lines * 128) != hipSuccess || hipMalloc(&out, blocks * 512 * 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, n_lines * 128);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(512), 0, 0, buf, n_lines, iters, out);
        (void)hipDeviceSynchronize();
    }
    const unsigned long long lines = (unsigned long long)blocks * 8 * iters * 16;
    printf("{\"kernel\": \"gather_kernel\", \"lines_per_launch\": %llu, \"bytes_per_launch\": %llu}\n", lines, lines * 128ull);
    return 0;
}
