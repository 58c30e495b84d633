// SIMD cost of individual vector instruction kinds on gfx950 with TWO waves per SIMD (the optimiser's regime):
//   (a) alone     : both waves run only the instruction (8 independent registers, unrolled x32)
//   (b) with MFMA : both waves run { v_mfma_f32_16x16x4_f32 ; 4 x instruction } - cycles added per instruction over the
//                   bare MFMA stream (32 cycles per MFMA per SIMD)
// Cycles are shader cycles (s_memtime), per block max(end) - min(start) over its 8 waves, median over the blocks.
// hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define KINDS(X)                                                                                                   \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %2, %3", 0)                                                               \
    X(1, "v_mul_f32", "v_mul_f32 %0, %0, %2", 0)                                                                   \
    X(2, "v_add_f32", "v_add_f32 %0, %0, %2", 0)                                                                   \
    X(3, "v_max_f32", "v_max_f32 %0, %0, %2", 0)                                                                   \
    X(4, "v_max_f32 neg", "v_max_f32_e64 %0, -%0, 0", 0)                                                           \
    X(5, "v_mul_f32 clamp", "v_mul_f32_e64 %0, %0, %2 clamp", 0)                                                   \
    X(6, "v_fmac_f32", "v_fmac_f32 %0, %2, %3", 0)                                                                 \
    X(7, "v_mov_b32", "v_mov_b32 %0, %2", 0)                                                                       \
    X(8, "v_max_i32", "v_max_i32 %1, %1, %4", 0)                                                                   \
    X(9, "v_and_b32", "v_and_b32 %1, %1, %4", 0)                                                                   \
    X(10, "v_add_u32", "v_add_u32 %1, %1, %4", 0)                                                                  \
    X(11, "v_alignbit_b32", "v_alignbit_b32 %1, %1, %4, 31", 0)                                                    \
    X(12, "v_bfe_i32", "v_bfe_i32 %1, %4, 3, 1", 0)                                                                \
    X(13, "v_bfi_b32", "v_bfi_b32 %1, %4, 0, %1", 0)                                                               \
    X(14, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %2, vcc", 0)                                                     \
    X(15, "v_cmp_gt_f32", "v_cmp_gt_f32 vcc, %0, %2", 0)                                                           \
    X(16, "v_and_b32_sdwa", "v_and_b32_sdwa %1, sext(%4), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", 0) \
    X(17, "v_mov_b32_sdwa b", "v_mov_b32_sdwa %1, %4 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3", 0) \
    X(18, "v_perm_b32", "v_perm_b32 %1, %1, %4, %4", 0)                                                            \
    X(19, "v_med3_u32", "v_med3_u32 %1, %1, %4, %4", 0)                                                            \
    X(20, "v_med3_f32", "v_med3_f32 %0, %0, %2, %3", 0)                                                            \
    X(21, "v_min_u32", "v_min_u32 %1, %1, %4", 0)                                                                  \
    X(22, "v_min_f32", "v_min_f32 %0, %0, %2", 0)                                                                  \
    X(23, "v_pk_fma_f32", "v_pk_fma_f32 %5, %5, %6, %7", 0)                                                        \
    X(24, "v_pk_mul_f32", "v_pk_mul_f32 %5, %5, %6", 0)                                                            \
    X(25, "v_pk_add_f32", "v_pk_add_f32 %5, %5, %6", 0)                                                            \
    X(26, "v_fma_mix_f32", "v_fma_mix_f32 %0, %0, %2, %3", 0)                                                      \
    X(27, "v_fma_mixlo_f16 clamp", "v_fma_mixlo_f16 %0, %2, %3, 0 clamp", 0)                                       \
    X(28, "v_exp_f32", "v_exp_f32 %0, %0", 0)                                                                      \
    X(29, "v_rcp_f32", "v_rcp_f32 %0, %0", 0)                                                                      \
    X(30, "v_sqrt_f32", "v_sqrt_f32 %0, %0", 0)                                                                    \
    X(31, "v_cvt_i32_f32", "v_cvt_i32_f32 %1, %2", 0)                                                              \
    X(32, "v_floor_f32", "v_floor_f32 %0, %2", 0)                                                                  \
    X(33, "v_sub_f32", "v_sub_f32 %0, %0, %2", 0)                                                                  \
    X(34, "v_lshl_add_u32", "v_lshl_add_u32 %1, %1, 2, %4", 0)                                                     \
    X(35, "v_mul_f32 dpp", "v_mul_f32_dpp %0, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 0)           \
    X(36, "v_mov_b32 dpp", "v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 0)               \
    X(37, "v_cvt_f32_fp8 b1", "v_cvt_f32_fp8_sdwa %0, %4 src0_sel:BYTE_1", 0)                                      \
    X(38, "v_max3_f32", "v_max3_f32 %0, %0, %2, %3", 0)                                                            \
    X(39, "v_mad_u32_u24", "v_mad_u32_u24 %1, %1, %4, %4", 0)                                                      \
    X(40, "v_lshrrev_b32", "v_lshrrev_b32 %1, 3, %1", 0)                                                           \
    X(41, "v_xor_b32", "v_xor_b32 %1, %1, %4", 0)                                                                  \
    X(42, "v_mul_u32_u24", "v_mul_u32_u24 %1, %1, %4", 0)                                                          \
    X(43, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %4", 0)                                                              \
    X(44, "v_log_f32", "v_log_f32 %0, %0", 0)                                                                      \
    X(45, "v_mul_legacy_f32", "v_mul_legacy_f32 %0, %0, %2", 0)                                                    \
    X(46, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %4", 0)                                                              \
    X(47, "v_fract_f32", "v_fract_f32 %0, %2", 0)

constexpr int NKINDS = 48;

template <int KIND, bool WITH_MFMA>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* tt, int iters) {
    f32x4 a4[4];
    for (int i = 0; i < 4; ++i) a4[i] = f32x4{0, 0, 0, 0};
    float a = 1.f + threadIdx.x, b = 2.f + threadIdx.x * 0.5f;
    float x[8];
    unsigned u[8];
    f32x2 p[8];
    for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x + j; u[j] = threadIdx.x * 7 + j; p[j] = f32x2{x[j], x[j] + 1.f}; }
    float m = 1.0001f, c = 0.5f;
    unsigned ui = (threadIdx.x & 63) * 4 + 1;
    f32x2 pm = {m, m}, pc = {c, c};
    asm volatile("" : "+v"(m), "+v"(c), "+v"(ui), "+v"(pm), "+v"(pc));
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (WITH_MFMA) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a4[s & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int j = (s * 4 + f) & 7;
#define X(ID, NAME, ASM, Z) if (KIND == ID) asm volatile(ASM : "+v"(x[j]), "+v"(u[j]) : "v"(m), "v"(c), "v"(ui), "v"(p[j]), "v"(pm), "v"(pc) : "vcc");
                KINDS(X)
#undef X
                if (KIND == 23 || KIND == 24 || KIND == 25) asm volatile("" : "+v"(p[j]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int j = 0; j < 8; ++j) r += x[j] + (float)u[j] + p[j][0] + p[j][1];
    for (int i = 0; i < 4; ++i) r += a4[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        tt[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t0;
        tt[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = t1;
    }
}

template <int KIND, bool WITH_MFMA>
double run(float* out, unsigned long long* tt, int iters) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, WITH_MFMA>), dim3(256), dim3(512), 0, 0, out, tt, iters);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256 * 16);
    (void)hipMemcpy(h.data(), tt, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per;
    for (int b = 0; b < 256; ++b) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < 8; ++w) { lo = std::min(lo, h[(b * 8 + w) * 2]); hi = std::max(hi, h[(b * 8 + w) * 2 + 1]); }
        per.push_back((double)(hi - lo));
    }
    std::sort(per.begin(), per.end());
    return per[128];
}

template <int KIND>
void test(float* out, unsigned long long* tt, const char* name, double base) {
    const int iters = 4000;
    const double slots = 2.0 * 8 * iters;                 // per SIMD: two waves x 8 slots per iteration
    const double alone = run<KIND, false>(out, tt, iters) / (slots * 4);
    const double with = run<KIND, true>(out, tt, iters) / slots;
    printf("%-24s alone %5.2f cyc/instr/SIMD | beside MFMA: %6.2f cyc per {MFMA + 4 instr} -> +%5.2f cyc per instr\n", name, alone,
           with, (with - base) / 4.0);
}

template <int KIND>
struct Sweep {
    static void go(float* out, unsigned long long* tt, double base) {
        const char* names[NKINDS] = {
#define X(ID, NAME, ASM, Z) NAME,
            KINDS(X)
#undef X
        };
        test<KIND>(out, tt, names[KIND], base);
        Sweep<KIND + 1>::go(out, tt, base);
    }
};
template <>
struct Sweep<NKINDS> {
    static void go(float*, unsigned long long*, double) {}
};

// bare MFMA stream: the kernel with zero fillers
__global__ __launch_bounds__(512) void kbase(float* out, unsigned long long* tt, int iters) {
    f32x4 a4[4];
    for (int i = 0; i < 4; ++i) a4[i] = f32x4{0, 0, 0, 0};
    float a = 1.f + threadIdx.x, b = 2.f + threadIdx.x * 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a4[s & 3]) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a4[0][0] + a4[1][0] + a4[2][0] + a4[3][0];
    if ((threadIdx.x & 63) == 0) {
        tt[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t0;
        tt[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = t1;
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    unsigned long long* tt; (void)hipMalloc(&tt, 256 * 16 * 8);
    const int iters = 4000;
    double base = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kbase, dim3(256), dim3(512), 0, 0, out, tt, iters);
        (void)hipDeviceSynchronize();
    }
    {
        std::vector<unsigned long long> h(256 * 16);
        (void)hipMemcpy(h.data(), tt, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> per;
        for (int b = 0; b < 256; ++b) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < 8; ++w) { lo = std::min(lo, h[(b * 8 + w) * 2]); hi = std::max(hi, h[(b * 8 + w) * 2 + 1]); }
            per.push_back((double)(hi - lo));
        }
        std::sort(per.begin(), per.end());
        base = per[128] / (2.0 * 8 * iters);
    }
    printf("bare MFMA stream: %.2f cycles per MFMA per SIMD\n", base);
    Sweep<0>::go(out, tt, base);
    return 0;
}
