// Host-side internal interface between the C ABI (api.cpp) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ifd_device.h"

namespace ifd {

struct OptArgs {
    int steps, t0, loss_batch, normalize, knn_scan_every_step;
    int shared_planes;   // experiment hook (env IFD_SHARED_PLANES): every cloud reads cloud 0's planes
    float lr, rep_weight, threshold, rep_radius, rep_h, rep_eps;
    DecConst dc;
};

hipError_t configure_optimize_kernels();
hipError_t launch_optimize(const float* dec_img, const float* planes, float* p, float* m, float* v, float* loss,
                           uint16_t* knn_lists, unsigned long long* counters, int B, int K, const OptArgs& a,
                           hipStream_t s);
// bytes of context workspace ifd_optimize needs for B clouds (certified neighbour lists)
size_t knn_list_bytes(int B);
hipError_t launch_decode(const float* dec_img, const float* planes, const float* p, int B, int K, float* logits,
                         float* dlogit_dp, DecConst dc, hipStream_t s);
hipError_t launch_repulsion(const float* p, int B, int K, float* loss, float* grad, int32_t* knn_idx, float radius,
                            float h, float eps, hipStream_t s);
hipError_t launch_normalize(float* p, int B, int K, hipStream_t s);

}  // namespace ifd
