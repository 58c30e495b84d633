// Host-side internal interface between the C ABI (api.cpp) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ifd_device.h"

namespace ifd {

struct OptArgs {
    int steps, t0, loss_batch, normalize, knn_scan_every_step, planes_shared;
    float lr, rep_weight, threshold, rep_radius, rep_h, rep_eps;
    DecConst dc;
    unsigned int coop_timeout_ticks;   // bound of a split cloud's cross-CU waits, in ticks of the 100 MHz wall clock (knn_device.h coop_wait)
    int test_drop_member;              // test hook (env IFD_TEST_COOP_DROP): this member of every split cloud never arrives; -1 = off
    int precision;                     // ifd_opt_params.precision: 0 f32 MFMA tiles, 1 bf16x6, 2 bf16x3 (tile_bf.h); the caller passes the matching image
};

// The context's device counter buffer (api.cpp d_counters, unsigned long long): IFD_N_COUNTERS public diagnostics of the last
// optimise call, the wave trace of -DIFD_TRACE builds, then two STICKY status words that no optimise call clears
// (ifd_optimize_status reads and resets them): points whose fixed-point repulsion sums came within a factor two of wrapping,
// and cross-CU waits of split clouds that gave up.
constexpr int DEV_COUNTERS = 16 + 8 * 32;
constexpr int STATUS_OVERFLOW = DEV_COUNTERS;
constexpr int STATUS_TIMEOUT = DEV_COUNTERS + 1;
// ... and the time-out word of the CURRENT optimise call: cleared by every ifd_optimize / ifd_onet_optimize in front of its launches.
// A waiter that gives up raises this word and the sticky one; the other waiters fall out on THIS word only (round-4 advisor: when
// they looked at the sticky word, one time-out made the first wait of every later launch on the context fail too, until the host
// had called ifd_optimize_status).
constexpr int STATUS_TIMEOUT_CUR = DEV_COUNTERS + 2;
constexpr int TRACE2_BASE = DEV_COUNTERS + 3;        // -DIFD_TRACE2 builds: [8 waves][128] stamps inside one decoder tile per wave
constexpr int DEV_COUNTERS_TOTAL = TRACE2_BASE + 8 * 128;

// offsets (floats) of the point-net tensors inside the canonical weight vector (include/ifd.h order)
struct EncPointOffsets {
    int pos_w, pos_b, fc0_w[5], fc0_b[5], fc1_w[5], fc1_b[5], sc_w[5], fcc_w, fcc_b;
};

hipError_t configure_encoder_kernels();
int enc_image_floats();
void build_enc_image(const float* w, const EncPointOffsets& eo, float* img);      // host: aligned point-net weights (encoder.hip)
// zero-fills planes itself
hipError_t launch_encode_points(const float* w, const EncPointOffsets& eo, const float* enc_img, const float* sel,
                                const int* t_per_cloud, int B, int Tmax, float* planes, float* c_out, DecConst dc, hipStream_t s);

// device pointers to the re-packed ([tap][Cin][Cout]) U-Net weights
struct UNetWeights {
    const float *down_w[4][2], *down_b[4][2];
    const float *up_t_w[3], *up_t_b[3], *up_w[3][2], *up_b[3][2];
    const float *fin_w, *fin_b;
    // the 3x3 layers once more in the Winograd F(2x2, 3x3) domain: U = G g G^T, [Cin / 16][16 xi][Cout][16] (unet.hip wino_kernel)
    const float *down_u[4][2], *up_u[3][2];
};
hipError_t configure_unet_kernels();
size_t unet_workspace_floats(int n_img);
hipError_t launch_unet(const UNetWeights& W, const float* x, float* out, float* ws, int n_img, hipStream_t s);

struct PrepArgs {
    int cloud_base;            // global index of cloud 0 of this call (RNG counter)
    int n_sel, n_opt;          // encoder subset size (600), optimised points per cloud (1024)
    float padding_scale, init_sigma;
    uint32_t seed_lo, seed_hi;
    int no_morton;             // measurement hook (IFD_TEST_NO_MORTON with IFD_ENABLE_TEST_HOOKS=1, read at create): the library's own draws keep their draw order
};
constexpr int PREP_MAXK = 10000;     // largest input cloud (points) of ifd_sor / ifd_prepare (prep.hip: LDS of prepare_kernel)
hipError_t configure_prep_kernels();
hipError_t launch_sor(const float* pc, int B, int K, int k_nn, double alpha, uint8_t* keep, double* value, hipStream_t s);
hipError_t launch_prepare(const float* pc, const uint8_t* keep, int B, int K, const PrepArgs& a, const int32_t* sel_idx,
                          const int32_t* init_idx, const float* noise, float* sel, int32_t* t_per_cloud, float* init,
                          int32_t* n_kept, float* proc_out, hipStream_t s);

hipError_t configure_optimize_kernels();
// ws: optimize_ws_bytes(B) of context workspace; split: ifd_opt_params.split; n_cu: compute units of the device
hipError_t launch_optimize(const float* dec_img, const float* planes, float* p, float* m, float* v, float* loss,
                           const int32_t* loss_batch_per_cloud, void* ws, unsigned long long* counters,
                           const float* adam_tab, int B, int K, const OptArgs& a, int split, int n_cu, hipStream_t s);
size_t optimize_ws_bytes(int B);
// per-step Adam bias corrections {lr / (1 - beta1^t), sqrt(1 - beta2^t)}, t = t0 + 1 ... t0 + steps -> tab[steps][2]
hipError_t launch_adam_table(float* tab, int t0, int steps, float lr, hipStream_t s);
// bytes of context workspace ifd_optimize needs for B clouds (certified neighbour lists)
size_t knn_list_bytes(int B);
hipError_t launch_decode(const float* dec_img, const float* planes, const float* p, int B, int K, float* logits,
                         float* dlogit_dp, DecConst dc, hipStream_t s);
// ... in a split-precision mode (decode_bf.hip: the optimiser's own tile in MODE_SUM); dec_img_bf = the bf16 piece image
hipError_t configure_decode_bf_kernels();
hipError_t launch_decode_bf(int prec, const float* dec_img_bf, const float* planes, const float* p, int B, int K, float* logits, float* dlogit_dp,
                            DecConst dc, int n_cu, hipStream_t s);
hipError_t launch_repulsion(const float* p, int B, int K, float* loss, float* grad, int32_t* knn_idx, float radius,
                            float h, float eps, hipStream_t s);
hipError_t launch_normalize(float* p, int B, int K, hipStream_t s);
// clouds of MAXK < K <= LARGE_MAXK optimised points: two launches per Adam step (optimize.hip, "large" section)
size_t large_ws_bytes(int B, int K, bool own_moments);
// dec_img: the image of a.precision (f32: the optimiser's copy; 1 / 2: the bf16 piece image)
hipError_t launch_large_occupancy(int precision, const float* dec_img, const float* planes, const float* p, int B, int parts, int K,
                                  const int32_t* loss_batch_per_cloud, int loss_batch, float thr, int want_loss, void* G, DecConst dc,
                                  hipStream_t s);
size_t large_list_bytes(int B, int K);        // certified neighbour lists of the launch-per-step path (0 beyond LARGE_LDS_MAXK points)
void* large_list_ws(void* ws, int B, int K, bool own_moments);                                      // ... inside ws (nullptr beyond)
size_t large_f_bytes(int B, int K);           // global repulsion accumulators of clouds beyond LARGE_LDS_MAXK points (0 below)
hipError_t large_f_prepare(void* ws, int B, int K, bool own_moments, void** f_ws, hipStream_t s);   // ... at the end of ws, zeroed
hipError_t launch_large_repulsion(const float* p, int B, int K, float* loss, float* grad, int32_t* knn_idx, float radius,
                                  float h, float eps, void* f_ws, hipStream_t s);
// one Adam step from the occupancy gradients G ([B][K] float4: d loss / d xyz, BCE term): exact 5-NN + repulsion + Adam
hipError_t launch_large_step(float* p, float* m, float* v, const void* G, int B, int K, const float* adam_tab, int step,
                             const int32_t* loss_batch_per_cloud, const OptArgs& a, float* loss, void* f_ws, void* list_ws,
                             unsigned long long* counters, hipStream_t s);
hipError_t launch_large_normalize(float* p, int B, int K, hipStream_t s);

// ---- ONet-Opt (onet.hip) --------------------------------------------------------------------------------
// offsets (floats) into the canonical ONet weight vector (include/ifd.h order)
struct OnetEncOffsets {
    int pos_w, pos_b, fc0_w[5], fc0_b[5], fc1_w[5], fc1_b[5], sc_w[5], fcc_w, fcc_b;
};
struct OnetDecOffsets {
    int cbn_gamma_w[11], cbn_gamma_b[11], cbn_beta_w[11], cbn_beta_b[11], cbn_mean[11], cbn_var[11], fc0_b[5];
};
hipError_t configure_onet_kernels();
hipError_t configure_onet_bf_kernels();
// split-precision ONet-Opt launch (onet_bf.hip): img_bf = the bf16 piece image of api.cpp onet_fragment_image_bf, precision 1 | 2
hipError_t launch_onet_optimize_bf(int precision, const float* img_bf, const float* small, const float* ab, float* p, float* m, float* v,
                                   float* loss, const int32_t* loss_batch_per_cloud, uint16_t* knn_lists, unsigned long long* counters,
                                   const float* adam_tab, int B, int K, const OptArgs& a, hipStream_t s);
size_t onet_encode_ws_floats(int B, int Tmax);
int onet_small_floats();
hipError_t launch_onet_encode(const float* w, const OnetEncOffsets& eo, const float* sel, const int* t_per_cloud, int B,
                              int Tmax, float* ws, float* c_out, hipStream_t s);
hipError_t launch_onet_cbn(const float* w, const OnetDecOffsets& od, const float* c, int B, float* gb, float* ab,
                           hipStream_t s);
hipError_t launch_onet_decode(const float* img, const float* small, const float* ab, const float* p, int B, int K,
                              float* logits, float* dlogit_dp, hipStream_t s);
hipError_t launch_onet_large_occupancy_bf(int precision, const float* img_bf, const float* small, const float* ab, const float* p, int B,
                                          int parts, int K, const int32_t* loss_batch_per_cloud, int loss_batch, float thr, void* G,
                                          hipStream_t s);                               // onet_bf.hip
hipError_t launch_onet_decode_bf(int precision, const float* img_bf, const float* small, const float* ab, const float* p, int B, int K,
                                 float* logits, float* dlogit_dp, hipStream_t s);      // onet_bf.hip
// clouds of MAXK < K <= LARGE_MAXK points (ONet/opt_defense.py:27 has no limit): two launches per Adam step, ws as
// large_ws_bytes
hipError_t launch_onet_large_optimize(const float* img, const float* small, const float* ab, float* p, float* m, float* v,
                                      float* loss, const int32_t* loss_batch_per_cloud, void* ws, unsigned long long* counters,
                                      const float* adam_tab, int B, int K, const OptArgs& a, hipStream_t s);
hipError_t launch_onet_optimize(const float* img, const float* small, const float* ab, float* p, float* m, float* v,
                                float* loss, const int32_t* loss_batch_per_cloud, uint16_t* knn_lists,
                                unsigned long long* counters, const float* adam_tab, int B, int K, const OptArgs& a,
                                hipStream_t s);

// ---- ONet-Mesh (mesh.hip) --------------------------------------------------------------------------------
// MISE state of a batch of clouds as dense arrays (per-cloud strides: P3 for val / known, pend_stride for pend,
// sub_total for sub / mix, cap for list)
struct MiseGrid {
    int res0, depth, P, P3, cap, sub_total, sub_off[4];
    size_t pend_stride;
    double threshold;
    float* val;          // [B][P3]   decoder logits at the grid points
    uint8_t* known;      // [B][P3]
    uint8_t* pend;       // [B][pend_stride]  queued for evaluation (byte flags)
    uint8_t* sub;        // [B][sub_total]    voxel of level l subdivided (levels concatenated)
    uint8_t* mix;        // [B][sub_total]    scratch of one update round
    int* list;           // [B][cap]  grid-point indices to evaluate this round
    int* count;          // [B]      queue length of this round (filled by the previous round's mise_apply_kernel)
    int* prev;           // [B]      points evaluated in the round mise_update is closing (0: nothing new is known, the cloud is skipped)
    int* plan;           // [B + 1]  exclusive prefix of the clouds' 128-point decoder passes of this round (grid_plan_kernel)
};
hipError_t launch_mise_init(const MiseGrid& g, int B, hipStream_t s);
hipError_t launch_mise_update(const MiseGrid& g, int B, hipStream_t s);
hipError_t launch_mise_fill(const MiseGrid& g, int B, hipStream_t s);
hipError_t launch_onet_grid_eval(const float* img, const float* small, const float* ab, const MiseGrid& g, int B,
                                 int n_blocks, float box, hipStream_t s);
hipError_t launch_onet_grid_eval_bf(int precision, const float* img_bf, const float* small, const float* ab, const MiseGrid& g, int B,
                                    int n_blocks, float box, hipStream_t s);
hipError_t mc_upload_table();
void mc_host_table(int8_t (*tri)[16], uint8_t* ntri);
hipError_t launch_marching_cubes(const float* val, int B, int P, double iso, float box, int* cube_offs, int* ntri_total,
                                 int cap, float* tris, double* area, hipStream_t s);
hipError_t launch_sample_surface(const float* tris, const double* cum_area, const int* ntri_total, int B, int cap, int n,
                                 uint64_t seed, int cloud_base, float* out, hipStream_t s);

}  // namespace ifd
