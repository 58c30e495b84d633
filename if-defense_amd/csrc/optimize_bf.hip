// The persistent optimiser with the split-precision decoder tiles (tile_bf.h): optimize_kernel<8, S, PREC> for PREC = 1 (bf16x6:
// six products of bf16 pieces per weight x activation - f32-equivalent) and PREC = 2 (bf16x3: three products - 2^-17), in a
// translation unit of their own (each instantiation is ~190 KB of code and a minute of compile time).
#include "optimize_kernel.h"

namespace ifd {

// 158,752 B (the moments are in global memory; the last 2 KB: one 256-byte landing strip per wave for the tap prefetches of tile_bf.h)
constexpr size_t OPT_LDS_BF = (size_t)BF_IMG_BYTES + MAXK * 16 * 3 + 16 + MAXK * 12 + 128 * 4 + 8 * 256;
static_assert(OPT_LDS_BF <= 160 * 1024, "LDS budget");
constexpr size_t LARGE_OCC_LDS_BF = (size_t)BF_IMG_BYTES + 8 * 256;     // large_occupancy3_kernel<1 | 2>: the piece image + the waves' landing strips

hipError_t configure_optimize_bf_kernels() {
    hipError_t e = hipSuccess;
#define IFD_CFG(S_, P_)                                                                                                      \
    if (e == hipSuccess)                                                                                                     \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(optimize_kernel<8, S_, P_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)OPT_LDS_BF);
    IFD_CFG(1, 1) IFD_CFG(2, 1) IFD_CFG(4, 1) IFD_CFG(1, 2) IFD_CFG(2, 2) IFD_CFG(4, 2)
#undef IFD_CFG
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(large_occupancy3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LARGE_OCC_LDS_BF);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(large_occupancy3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LARGE_OCC_LDS_BF);
    return e;
}

// the occupancy half of the launch-per-step path (clouds of more than MAXK points) on the split-precision tile
hipError_t launch_large_occupancy_bf(int prec, const float* dec_img, const float* planes, const float* p, int B, int parts, int K,
                                     const int32_t* lbpc, int loss_batch, float thr, int want_loss, f32x4* G, DecConst dc, hipStream_t s) {
    if (prec == 1)
        hipLaunchKernelGGL(large_occupancy3_kernel<1>, dim3(B, parts), dim3(OPT_THREADS), LARGE_OCC_LDS_BF, s, dec_img, planes, p, K, lbpc,
                           loss_batch, thr, want_loss, G, dc);
    else
        hipLaunchKernelGGL(large_occupancy3_kernel<2>, dim3(B, parts), dim3(OPT_THREADS), LARGE_OCC_LDS_BF, s, dec_img, planes, p, K, lbpc,
                           loss_batch, thr, want_loss, G, dc);
    return hipGetLastError();
}

template <int S, int PREC>
static hipError_t launch_bf(const float* dec_img, const float* planes, float* p, float* m, float* v, float* loss, const int32_t* lb,
                            uint16_t* knn_lists, unsigned long long* counters, const float* adam_tab, int grid, int K, const OptArgs& a,
                            CoopWs* coop, int n, void* mv_ws, hipStream_t s) {
    hipLaunchKernelGGL((optimize_kernel<8, S, PREC>), dim3(grid), dim3(512), OPT_LDS_BF, s, dec_img, planes, p, m, v, loss, lb, knn_lists,
                       counters, adam_tab, K, a, coop, n, static_cast<f32x4*>(mv_ws));
    return hipGetLastError();
}

// one launch of optimize.hip's launch_part (pointers already offset to the launch's first cloud)
hipError_t launch_part_bf(int S, int prec, const float* dec_img, const float* planes, float* p, float* m, float* v, float* loss,
                          const int32_t* lb, uint16_t* knn_lists, unsigned long long* counters, const float* adam_tab, int grid, int K,
                          const OptArgs& a, CoopWs* coop, int n, void* mv_ws, hipStream_t s) {
#define IFD_GO(S_, P_) return launch_bf<S_, P_>(dec_img, planes, p, m, v, loss, lb, knn_lists, counters, adam_tab, grid, K, a, coop, n, mv_ws, s)
    if (prec == 1) {
        if (S == 1) IFD_GO(1, 1);
        if (S == 2) IFD_GO(2, 1);
        IFD_GO(4, 1);
    }
    if (S == 1) IFD_GO(1, 2);
    if (S == 2) IFD_GO(2, 2);
    IFD_GO(4, 2);
#undef IFD_GO
}

}  // namespace ifd
