// Is an already-satisfied s_waitcnt a wait state?  hipcc (ROCm 7.2) pads a packed-f32 producer -> dependent consumer pair
// (v_pk_mul_f32 / v_pk_fma_f32 run two passes through the vector ALU) with one wait state and counts an s_waitcnt that happens to
// stand between them as that state.  This probe issues the pair from inline asm (nothing padded) with  (a) nothing,  (b) an
// s_waitcnt whose condition already holds,  (c) s_nop 0  in between, and counts wrong results.
// RESULT (MI355X): 0 wrong results in every form - the hypothesis is WRONG; the errors it was meant to explain come from packed f32 beside
// a bf16-MFMA partner wave (scripts/pk_mfma_coexec.hip).
//   hipcc --offload-arch=gfx950 -O3 pk_waitcnt_hazard.hip -o pk_waitcnt_hazard && ./pk_waitcnt_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(const float* in, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 x = {in[4 * i], in[4 * i + 1]}, y = {in[4 * i + 2], in[4 * i + 3]};
    f32x2 a = {7.f, 9.f}, b;            // a: stale content the consumer would see
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7" ::: "memory");
    if (MODE == 0)
        asm volatile("v_pk_mul_f32 %0, %2, %3\n\tv_pk_fma_f32 %1, %0, %3, %2" : "+v"(a), "=&v"(b) : "v"(x), "v"(y));
    if (MODE == 1)
        asm volatile("v_pk_mul_f32 %0, %2, %3\n\ts_waitcnt vmcnt(0)\n\tv_pk_fma_f32 %1, %0, %3, %2" : "+v"(a), "=&v"(b) : "v"(x), "v"(y));
    if (MODE == 2)
        asm volatile("v_pk_mul_f32 %0, %2, %3\n\ts_nop 0\n\tv_pk_fma_f32 %1, %0, %3, %2" : "+v"(a), "=&v"(b) : "v"(x), "v"(y));
    if (MODE == 3)
        asm volatile("v_pk_mul_f32 %0, %2, %3\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_fma_f32 %1, %0, %3, %2" : "+v"(a), "=&v"(b) : "v"(x), "v"(y));
    asm volatile("s_nop 7" ::: "memory");
    out[2 * i] = b.x;
    out[2 * i + 1] = b.y;
}

template <int MODE>
void run(const char* name, const float* din, float* dout, const std::vector<float>& h, int n) {
    hipLaunchKernelGGL(k<MODE>, dim3(n / 256), dim3(256), 0, 0, din, dout);
    std::vector<float> o(2 * n);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 2; ++c) {
            const float x = h[4 * i + c], y = h[4 * i + 2 + c];
            const float want = fmaf(x * y, y, x);
            bad += o[2 * i + c] != want;
        }
    printf("%-52s wrong results: %d of %d\n", name, bad, 2 * n);
}

int main() {
    const int n = 1 << 16;
    std::vector<float> h(4 * n);
    for (int i = 0; i < 4 * n; ++i) h[i] = 0.25f + (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    float *din, *dout;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, 2 * n * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("v_pk_mul ; v_pk_fma (dependent), back to back", din, dout, h, n);
    run<1>("v_pk_mul ; s_waitcnt vmcnt(0) [satisfied] ; v_pk_fma", din, dout, h, n);
    run<3>("v_pk_mul ; s_waitcnt lgkmcnt(0) [satisfied] ; v_pk_fma", din, dout, h, n);
    run<2>("v_pk_mul ; s_nop 0 ; v_pk_fma", din, dout, h, n);
    return 0;
}
