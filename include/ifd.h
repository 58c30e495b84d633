/*
 * ifd.h - C ABI of libifd.so: the MI355X (gfx950) implementation of IF-Defense's
 * ConvONet-Opt restoration hot path.
 *
 * The reference (Wuziyi616/IF-Defense) has no plugin / FFI interface: the path
 * sits behind Python call seams.  Each entry point below names the reference
 * call it replaces (paths relative to the reference repo).  INTEGRATION.md shows
 * the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only, no torch / C++ types.
 *   - every function returns an int status (IFD_OK == 0); nothing throws or
 *     calls exit() across the ABI (the reference's OOM loops call exit(-1):
 *     ConvONet/defense/repulsion_loss.py:30-32, SOR.py:57-59).
 *   - all array arguments are DEVICE pointers owned by the caller unless the
 *     name ends in _host.  float = IEEE binary32.  Arrays are dense, row-major.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls
 *     only enqueue work on it and return; the host blocks in: ifd_create /
 *     ifd_onet_create / ifd_destroy, ifd_get_counters, ifd_onet_mesh_sample (waits
 *     on one event per MISE round, a round behind what it has enqueued), and in a call that needs more
 *     context workspace than any earlier call on that context (a one-time
 *     device synchronisation while the workspace grows).
 *   - a context is bound to the device given to ifd_create: every entry point
 *     makes that device current for the duration of the call and restores the
 *     caller's (launches and workspace never land on another GPU).  One thread
 *     at a time per context; different contexts are independent (no hidden globals).
 *   - a context owns mutable scratch that its calls use on the caller's stream: the
 *     optimise calls (ifd_optimize / ifd_onet_optimize) share the neighbour lists, the
 *     Adam table and the counters (and, with ifd_onet_decode / ifd_onet_mesh_sample, the
 *     folded CBN coefficients of an ONet context); the encoder calls (ifd_encode_* /
 *     ifd_unet / ifd_onet_encode) share the encoder scratch and nothing else.  Calls of one kind on one context
 *     must therefore be stream-ordered with each other (one stream per kind, or
 *     events between them); an optimise call and an encoder call may overlap on
 *     two streams (pipeline.defend_stream does exactly that).
 *
 * Plane layout used on the device ("channel-last"):
 *     planes[b][plane][row][col][ch],  plane in {0:xz, 1:xy, 2:yz}, row = u1 cell,
 *     col = u0 cell, ch in [0,32)   ->  float[B][3][64][64][32]  (1.5 MiB / cloud)
 * i.e. reference tensor c[plane][b, ch, row, col]
 * (ConvONet/src/encoder/pointnet.py:79-80) permuted so that one bilinear tap is
 * one contiguous 128-byte line.
 */
#ifndef IFD_H
#define IFD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ifd_opt_params grew (split, planes_shared - round 3), ifd_optimize_status and the two status codes below were added */
/* 3: ifd_opt_params.precision (round 5) */
/* 4: ifd_mesh_params.precision (round 5) */
/* 5: ifd_decode_ex, ifd_onet_decode_ex (round 6; nothing else changed: a caller of ABI 4 only needs the new version number) */
#define IFD_ABI_VERSION 5

enum {
    IFD_OK = 0,
    IFD_ERR_ARG = -1,        /* bad argument (NULL, size out of the supported range) */
    IFD_ERR_HIP = -2,        /* a HIP runtime call failed; see ifd_last_error() */
    IFD_ERR_UNSUPPORTED = -3,/* valid in the reference but not built here (e.g. 'grid' planes) */
    IFD_ERR_NOMEM = -4,
    IFD_ERR_TIMEOUT = -5,    /* ifd_optimize_status: a cross-CU wait of a split cloud gave up (results of that launch invalid) */
    IFD_ERR_OVERFLOW = -6    /* ifd_optimize_status: fixed-point repulsion sums reached half their range (results invalid) */
};

/* Resolved hyper-parameters (ConvONet/configs/convonet_3plane_mn40.yaml + default.yaml).
 * Only the shipped 3-plane configuration is supported; ifd_create validates it. */
typedef struct ifd_config {
    int32_t struct_size;      /* sizeof(ifd_config), for forward compatibility */
    int32_t plane_resolution; /* 64   (yaml: model.encoder_kwargs.plane_resolution) */
    int32_t c_dim;            /* 32   (yaml: model.c_dim) */
    int32_t hidden_dim;       /* 32   (encoder hidden_dim == decoder hidden_size) */
    int32_t n_blocks;         /* 5    (decoder.py:23, pointnet.py:33 defaults) */
    int32_t unet_depth;       /* 4 */
    int32_t unet_start_filts; /* 32 */
    float   padding;          /* 0.1  (default.yaml data.padding) */
} ifd_config;

typedef struct ifd_ctx ifd_ctx;

/* Number of floats ifd_create expects in `weights_host`, and the canonical
 * order: the reference checkpoint's state_dict tensors, each in its native
 * torch layout, concatenated in this order (names as in pretrain/convonet.pth):
 *   decoder.fc_p.{weight[32,3],bias[32]}
 *   decoder.fc_c.{0..4}.{weight[32,32],bias[32]}
 *   decoder.blocks.{0..4}.{fc_0.weight[32,32],fc_0.bias,fc_1.weight[32,32],fc_1.bias}
 *   decoder.fc_out.{weight[1,32],bias[1]}
 *   encoder.fc_pos.{weight[64,3],bias[64]}
 *   encoder.blocks.{0..4}.{fc_0.weight[32,64],fc_0.bias,fc_1.weight[32,32],fc_1.bias,shortcut.weight[32,64]}
 *   encoder.fc_c.{weight[32,32],bias[32]}
 *   encoder.unet.down_convs.{0..3}.{conv1.weight,conv1.bias,conv2.weight,conv2.bias}
 *   encoder.unet.up_convs.{0..2}.{upconv.weight,upconv.bias,conv1.weight,conv1.bias,conv2.weight,conv2.bias}
 *   encoder.unet.conv_final.{weight[32,32,1,1],bias[32]}
 * Replaces: model.load_state_dict(torch.load(...)) (ConvONet/opt_defense.py:64-65). */
size_t ifd_weight_count(void);

int  ifd_abi_version(void);

/* Replaces config.get_model + load_state_dict + get_generator
 * (ConvONet/opt_defense.py:56-73).  `weights_host` is HOST memory. */
ifd_ctx* ifd_create(const float* weights_host, size_t n_weights, const ifd_config* cfg, int device);
void     ifd_destroy(ifd_ctx* ctx);
const char* ifd_last_error(const ifd_ctx* ctx);   /* ctx may be NULL: error of the last failed ifd_create */

/* ---- pre-processing ------------------------------------------------------------ */

/* sor_process / SORDefense.outlier_removal (ConvONet/opt_defense.py:86-111, defense/SOR.py:22-49):
 * pc [B,K,3] -> keep_mask [B,K] (1 = kept: value <= mean + alpha * std, float64 like the reference);
 * value (optional) [B,K] float64 = mean of the k nearest squared distances.  2 <= K <= 10000, k <= 7. */
int ifd_sor(ifd_ctx* ctx, const float* pc, int B, int K, int k, float alpha, uint8_t* keep_mask,
            double* value, void* stream);

typedef struct ifd_prep_params {
    int32_t  struct_size;      /* sizeof(ifd_prep_params) */
    int32_t  n_sel;            /* cfg data.pointcloud_n: encoder subset size, 600 */
    int32_t  n_opt;            /* --sample_npoint: optimised points per cloud, 1024 */
    float    padding_scale;    /* --padding_scale, 0.9 */
    float    init_sigma;       /* --init_sigma, 0.01 */
    uint64_t seed;             /* the reference never seeds; draws here = f(seed, global cloud index, draw index) */
    int64_t  cloud_index_base; /* global index of cloud 0 of this call (sharding-invariant random draws) */
} ifd_prep_params;

/* preprocess_pc + init_points (ConvONet/opt_defense.py:114-146, 149-179) on the points SOR kept.
 * keep_mask may be NULL (--sor=False).  Explicit draws (all optional, for parity tests): sel_idx [B,n_sel]
 * (np.random.choice without replacement), init_idx [B,n_opt] (torch.randint), noise [B,n_opt,3] ~ N(0,1).
 * With the library's own draws (init_idx == NULL and noise == NULL) and n_opt <= 4096 the rows of init_points are written in MORTON
 * ORDER of their coordinates: the reference's draws are i.i.d. (torch.randint / torch.randn, opt_defense.py:166-176), so the row order
 * carries no information, and neighbouring rows being neighbours in space makes the optimiser's plane gathers local (0.8 % / 4.9 % of
 * its launch in f32 / bf16x6).  Explicit draws keep their order.
 * Outputs: sel [B,n_sel,3] (rows >= t_per_cloud[b] are zero), t_per_cloud [B] = min(n_kept, n_sel),
 * init_points [B,n_opt,3], n_kept (optional) [B], proc (optional) [B,K,3]: the processed kept points,
 * first n_kept[b] rows valid.  K <= 10000, n_sel <= 1024. */
int ifd_prepare(ifd_ctx* ctx, const float* pc, const uint8_t* keep_mask, int B, int K, const ifd_prep_params* prm,
                const int32_t* sel_idx, const int32_t* init_idx, const float* noise, float* sel,
                int32_t* t_per_cloud, float* init_points, int32_t* n_kept, float* proc, void* stream);

/* ---- encoder ------------------------------------------------------------------ */

/* Point-wise half of generator.model.encode_inputs (ConvONet/opt_defense.py:300 ->
 * src/encoder/pointnet.py:124-156 + generate_plane_features :68-80 up to, not including, the U-Net):
 * fc_pos, 5 ResnetBlockFC with pool_local (scatter_max/gather over the 3 planes), fc_c, scatter_mean.
 * sel [B,Tmax,3] (the 600-point encoder subsets); t_per_cloud [B] int32 or NULL (= Tmax for every cloud):
 * clouds may hold fewer than Tmax valid points (the reference's torch.cat at :284 cannot express that).
 * planes_pre [B,3,64,64,32] channel-last, fully written (zero where no point falls).
 * c_points (optional) [B,Tmax,32]: the per-point features c.   1 <= Tmax <= 1024. */
int ifd_encode_points(ifd_ctx* ctx, const float* sel, const int32_t* t_per_cloud, int B, int Tmax,
                      float* planes_pre, float* c_points, void* stream);

/* The shared 2-D U-Net applied to the three planes of every cloud (pointnet.py:82-84 -> src/encoder/unet.py:225-239):
 * planes_pre [B,3,64,64,32] -> planes [B,3,64,64,32], both channel-last.  Uses context-owned scratch
 * (about 15 MB per cloud, grown on demand).  The 3x3 convolutions (unet.py:48-57) are evaluated in the Winograd
 * F(2x2, 3x3) domain in float32 (within 1e-6 of the planes' maximum of the plain nine-tap sum; fixed summation order,
 * independent of B); environment IFD_UNET_DIRECT=1 selects the nine-tap implicit-GEMM kernels instead (validation). */
int ifd_unet(ifd_ctx* ctx, const float* planes_pre, int B, float* planes, void* stream);

/* generator.model.encode_inputs(x) (ConvONet/opt_defense.py:300 -> src/conv_onet/models/__init__.py:52 ->
 * LocalPoolPointnet.forward): ifd_encode_points followed by ifd_unet.  sel [B,Tmax,3], t_per_cloud as above,
 * planes [B,3,64,64,32] channel-last = {'xz','xy','yz': [B,32,64,64]} of the reference, permuted. */
int ifd_encode_planes(ifd_ctx* ctx, const float* sel, const int32_t* t_per_cloud, int B, int Tmax,
                      float* planes, void* stream);

/* ---- decoder / losses / optimiser: the 501-step hot loop --------------------- */

/* generator.model.decode(p, c).logits (ConvONet/opt_defense.py:212 ->
 * src/conv_onet/models/decoder.py:69-95).  p [B,K,3]; planes channel-last;
 * logits [B,K]; dlogit_dp (optional, may be NULL) [B,K,3] = d(sum logits)/dp. */
int ifd_decode(ifd_ctx* ctx, const float* planes, const float* p, int B, int K,
               float* logits, float* dlogit_dp, void* stream);
/* ... in the arithmetic ifd_opt_params.precision names (0: ifd_decode itself; 1 = bf16x6, 2 = bf16x3: the optimiser's own
 * split-precision tile evaluating sum(logits) instead of the loss), so that the seam and ifd_optimize agree in every mode.
 * (The reference has one arithmetic; a maintainer binds ifd_decode.) */
int ifd_decode_ex(ifd_ctx* ctx, const float* planes, const float* p, int B, int K, int precision,
                  float* logits, float* dlogit_dp, void* stream);

/* repulsion_loss(p) (ConvONet/opt_defense.py:221 -> defense/repulsion_loss.py:18-54,
 * defense/pn_utils.py:64-83).  loss [B] = mean over K*5 of (r-d)*exp(-(d/h)^2);
 * grad (optional) [B,K,3] = d(sum_b loss_b)/dp;  knn_idx (optional) [B,K,5] int32,
 * neighbours sorted by increasing distance. */
int ifd_repulsion(ifd_ctx* ctx, const float* p, int B, int K,
                  float* loss, float* grad, int32_t* knn_idx, void* stream);

typedef struct ifd_opt_params {
    int32_t struct_size;  /* sizeof(ifd_opt_params) */
    int32_t steps;        /* number of Adam steps = --iterations + 1 (opt_defense.py:210) */
    int32_t t0;           /* steps already taken (0 for a fresh run); Adam bias correction uses t0+1.. */
    int32_t loss_batch;   /* the reference's batch size B in the two torch.mean() (1/B factor, :215,:222) */
    int32_t normalize;    /* !=0: finish with normalize_batch_pc (opt_defense.py:76-83,:238) */
    float   lr;           /* --lr, 1e-3 */
    float   rep_weight;   /* --rep_weight, 500 */
    float   threshold;    /* cfg test.threshold, 0.2 */
    float   rep_radius;   /* 0.07  (RepulsionLoss defaults, repulsion_loss.py:9-10) */
    float   rep_h;        /* 0.03 */
    float   rep_eps;      /* 1e-12 */
    int32_t knn_scan_every_step; /* validation only: !=0 disables the certified neighbour lists and runs the
                                    exact brute-force 5-NN scan at every step (same results, slower) */
    int32_t split;        /* workgroups (CUs) per cloud in the persistent kernel (ifd_optimize, K <= 1024).  0 = automatic:
                             one per cloud in whole rounds of one cloud per CU, and the clouds of the last partial round
                             split over 2 or 4 CUs each when that fills the GPU (fewer clouds than CUs, e.g. one GPU's shard
                             of a file spread over 8 GPUs, would otherwise cost a whole round); 1 = never split; 2 / 4 =
                             every cloud split (validation).  The results do not depend on it, bit for bit.  The
                             members of a split cloud wait for each other on the GPU: do not run two processes with
                             split launches on one GPU at the same time (set 1 there). */
    int32_t planes_shared;/* measurement only: != 0 makes every cloud read the planes of cloud 0 (the tap gathers then hit in
                             L2: scripts/ab_planes.py prices the gather traffic this way) */
    int32_t knn_reference_form; /* validation only: != 0 ranks the neighbours exactly like the reference - float32 expanded form
                             |a|^2 + |b|^2 - 2 a.b in torch's accumulation order, top-6, column 0 dropped whatever it is
                             (ConvONet/defense/pn_utils.py:72-83), by brute force every step - instead of the exact 5-NN of direct
                             differences.  The reference's form swaps candidates closer than its ~1e-7 noise and, for pairs of points
                             closer than ~1.5e-4, keeps "self" as a neighbour and drops the pair's term; the product path does not.
                             (topk's order among EQUAL distances is restated for rows of >= 384 values - torch's partial_sort path,
                             which 1024-point clouds take; shorter rows go through nth_element there and ties may differ.) */
    int32_t precision;    /* arithmetic of the decoder's 32 x 32 layers (the 256 x 256 ones of ifd_onet_optimize; every K):
                             0 = f32 MFMA (v_mfma_f32_16x16x4_f32: bit-equal to an fmaf chain) - the default;
                             1 = "bf16x6": both operands split exactly into three bf16 pieces, six piece products on the bf16
                                 matrix core, f32 accumulation - f32-equivalent (dropped terms <= 2^-26 of a product, one
                                 rounding per 32-term sum: measured closer to float64 than the f32 MFMA chain);
                             2 = "bf16x3": two pieces, three products - reduced precision, 2^-17 of a product.
                             Everything else (sampling, fc_p, fc_out, loss, kNN, repulsion, Adam) is f32 in every mode. */
} ifd_opt_params;

/* optimize_points(opt_points, z, c, rep_weight, iterations) (ConvONet/opt_defense.py:182-239).
 * p [B,K,3]: in = initial points, out = optimised (and normalised if requested).
 * m, v (optional, both or neither) [B,K,3]: Adam moments, read when t0 > 0 and always
 * written back - lets a caller resume or teacher-force single steps.
 * loss (optional) [B,2]: {sum_k BCE_k, repulsion_loss_b} evaluated at the last step's
 * pre-update points (what the reference prints at :229-236, before 1/B and weights).
 * loss_batch_per_cloud (optional) [B] int32: overrides prm->loss_batch per cloud, so clouds that belong to
 * different reference batches (the last batch of a file is shorter) can share one launch.
 * 6 <= K <= 10000.  Up to 1024 points a cloud runs in the persistent one-launch kernel; 1025 ... 10000 points take two
 * launches per Adam step (decoder gradient on the persistent kernel's tile, in the requested precision; exact 5-NN +
 * repulsion + Adam), same arithmetic.  Up to 4096 points the second launch keeps certified neighbour lists of its own
 * (bit-identical to the brute-force scan, which knn_scan_every_step selects and which clouds beyond 4096 points always
 * take); knn_reference_form applies at every size.  Batches of 32 clouds and more go as up to four groups of clouds on streams
 * of the context beside `stream` (a launch ends with its slowest cloud; the groups fill each other's idle CUs): forked from and
 * joined into `stream`, so the call is ordered on `stream` like every other, and bit-identical to one group.  Counters of this path: [0] whole-cloud list builds (one per cloud
 * and call), [5] point-steps whose certificate did not hold and that were resolved by the exact query. */
int ifd_optimize(ifd_ctx* ctx, const float* planes, float* p, int B, int K,
                 const ifd_opt_params* prm, const int32_t* loss_batch_per_cloud,
                 float* m, float* v, float* loss, void* stream);

/* Diagnostics (no reference counterpart): counters of the most recent ifd_optimize on this context, copied
 * to HOST memory; synchronises the device.  [0] wave-level neighbour-list rebuilds, [1] wave-level certificate
 * failures served by the exact brute-force scan, [2] extra wave-level rebuild work (exact scans for the list
 * radii + re-passes for overflowed balls), [3] shader-clock cycles cloud 0 spent in the optimiser kernel
 * (effective clock = cycles / kernel time), [4] wave-steps that had to evaluate the back ring of the lists,
 * [5] wave-steps with a near-tie settled by an exact per-point query (and the loss-reporting step), [6] wave-steps with individual
 * list refreshes, [7] lists built in total; of the most recent ifd_onet_mesh_sample: [8] grid points evaluated,
 * [9] MISE rounds (summed over chunks).  n <= IFD_N_COUNTERS (slots beyond it exist only for the wave trace of
 * diagnostic -DIFD_TRACE builds). */
#define IFD_N_COUNTERS 16
int ifd_get_counters(ifd_ctx* ctx, uint64_t* out_host, int n);

/* Device-side failures of the optimise calls issued on this context since the last call of this function (either model).
 * The reference's own failure path in this loop is exit(-1) (ConvONet/defense/repulsion_loss.py:25-39, SOR.py:57-59); here
 * nothing exits or hangs: the kernels raise sticky status words and this call reports them.  Synchronises `stream` (call it
 * where the results are consumed), resets the words, returns
 *   IFD_OK
 *   IFD_ERR_TIMEOUT   a member workgroup of a split cloud (ifd_opt_params.split != 1) did not arrive within the bound of the
 *                     cross-CU waits (30 s of wall-clock time; environment IFD_COOP_TIMEOUT_MS, read when the context is
 *                     created) - CUs masked or held by another process's split
 *                     launch.  Every waiter falls out, the launch ends early, its output is invalid.
 *   IFD_ERR_OVERFLOW  the repulsion gradient of some point summed to |x| >= 128 in a step: the 32-bit fixed-point accumulators
 *                     (2^-23 units, wrap at 256) are within a factor two of wrapping; with the reference's radius / h the sum
 *                     is bounded by 91.  The output of that launch is invalid.
 * ifd_last_error() carries the counts. */
int ifd_optimize_status(ifd_ctx* ctx, void* stream);

/* normalize_batch_pc (ConvONet/opt_defense.py:76-83) in place on p [B,K,3]. */
int ifd_normalize_unit_sphere(ifd_ctx* ctx, float* p, int B, int K, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ONet-Opt variant (ONet/opt_defense.py; BASELINE config #1).
 *
 * Same restoration loop with the Occupancy-Network model of ONet/configs/onet_mn40.yaml: encoder
 * ResnetPointnet(c_dim 512, hidden 512) on pointcloud_n = 300 points, decoder DecoderCBatchNorm(z_dim 0, c_dim 512,
 * hidden 256) in eval mode.  `diff ConvONet/opt_defense.py ONet/opt_defense.py` is the config path, the
 * decode(p, z, c) call (:212) and the save name, so ifd_sor / ifd_prepare (n_sel = 300) / ifd_repulsion /
 * ifd_normalize_unit_sphere / ifd_get_counters serve both kinds of context; the entry points below replace
 * ifd_encode_planes / ifd_decode / ifd_optimize, with the latent code c [B,512] in place of the planes (z is empty).
 *
 * Canonical weight order = the reference checkpoint's state_dict order (ONet `pretrain/onet.pth`) without the
 * BatchNorm num_batches_tracked scalars, every tensor row-major float32, Conv1d kernels squeezed:
 *   decoder.fc_p.{weight[256,3],bias}; for i in 0..4: decoder.block{i}.bn_0.{conv_gamma.weight[256,512],
 *   conv_gamma.bias, conv_beta.weight[256,512], conv_beta.bias, bn.running_mean, bn.running_var},
 *   decoder.block{i}.bn_1.{same six}, decoder.block{i}.fc_0.{weight[256,256],bias}, decoder.block{i}.fc_1.{weight,
 *   bias}; decoder.bn.{same six}; decoder.fc_out.{weight[256],bias[1]};
 *   encoder.fc_pos.{weight[1024,3],bias}; for i in 0..4: encoder.block_{i}.{fc_0.weight[512,1024], fc_0.bias,
 *   fc_1.weight[512,512], fc_1.bias, shortcut.weight[512,1024]}; encoder.fc_c.{weight[512,512],bias}
 *   (10,379,521 floats). */
#define IFD_MODEL_CONVONET 0
#define IFD_MODEL_ONET 1
size_t ifd_onet_weight_count(void);

/* config.get_model + load_state_dict + model.eval() (ONet/opt_defense.py:64-73).  NULL on failure
 * (ifd_last_error(NULL)). */
ifd_ctx* ifd_onet_create(const float* weights_host, size_t n_weights, int device);

/* generator.model.encode_inputs(x) (ONet/opt_defense.py:300; im2mesh/encoder/pointnet.py:86-113):
 * sel [B,Tmax,3] (+ optional t_per_cloud [B]: valid points per cloud) -> c [B,512]. */
int ifd_onet_encode(ifd_ctx* ctx, const float* sel, const int32_t* t_per_cloud, int B, int Tmax, float* c,
                    void* stream);

/* generator.model.decode(p, z, c).logits (ONet/opt_defense.py:212; onet/models/decoder.py:115-133):
 * c [B,512], p [B,K,3] -> logits [B,K]; dlogit_dp (optional) [B,K,3] = d(sum of logits)/dp. */
int ifd_onet_decode(ifd_ctx* ctx, const float* c, const float* p, int B, int K, float* logits,
                    float* dlogit_dp, void* stream);
/* ... with ifd_opt_params.precision's arithmetic (see ifd_decode_ex). */
int ifd_onet_decode_ex(ifd_ctx* ctx, const float* c, const float* p, int B, int K, int precision, float* logits,
                       float* dlogit_dp, void* stream);

/* optimize_points (ONet/opt_defense.py:182-239); arguments as ifd_optimize with c [B,512] for the planes. */
int ifd_onet_optimize(ifd_ctx* ctx, const float* c, float* p, int B, int K, const ifd_opt_params* prm,
                      const int32_t* loss_batch_per_cloud, float* m, float* v, float* loss, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ONet-Mesh path (ONet/remesh_defense.py; BASELINE config #4), on an ONet context.
 *
 * reconstruct_mesh + resample_points (remesh_defense.py:128-170): generator.generate_from_latent(z, c)
 * (im2mesh/onet/generation.py:88-178: MISE occupancy grid, resolution0 32 and 2 upsampling steps -> 129^3, marching
 * cubes on the -1e6-padded grid) followed by trimesh.sample.sample_surface(mesh, n_sample).
 *   c [B,512] -> points [B,n_sample,3] (NOT yet normalised: call ifd_normalize_unit_sphere, remesh_defense.py:262),
 *   n_triangles [B] (device int32; 0 = empty mesh: the cloud's rows of `points` are left untouched and the caller
 *   applies the reference's fallback, remesh_defense.py:160-170).
 * Optional outputs for inspection / tests: grid [B,P,P,P] float32 (P = resolution0 * 2^upsampling_steps + 1; MISE's
 * to_dense()), triangles [B,max_triangles,9] float32 (three xyz vertices per triangle in the decoder's frame; the
 * first n_triangles[b] rows are valid).  The surface samples use the counter-based generator of ifd_prepare, keyed by
 * (seed, cloud_index_base + b, sample index) - the reference's are unseeded numpy draws.
 * The MISE loop is driven from the device (queue lengths, the split of a round's decoder passes over the CUs, finished clouds);
 * the host enqueues round r before it has looked at what round r - 1 queued and waits on one event per round only to learn when
 * every queue has run empty - the GPU never waits for the host. */
typedef struct ifd_mesh_params {
    int32_t struct_size;       /* sizeof(ifd_mesh_params) */
    int32_t resolution0;       /* cfg generation.resolution_0 (32) */
    int32_t upsampling_steps;  /* cfg generation.upsampling_steps (2); 0..2, resolution0 << steps <= 128 */
    int32_t n_sample;          /* args.sample_npoint (1024) */
    int32_t max_triangles;     /* capacity per cloud of the triangle buffer (e.g. 400000) */
    float padding;             /* Generator3D padding (0.1): box_size = 1 + padding */
    double threshold;          /* cfg test.threshold (0.2), as a probability; the iso-value is its logit */
    uint64_t seed;
    int64_t cloud_index_base;
    int32_t precision;         /* arithmetic of the decoder layers in the grid evaluation, as ifd_opt_params.precision: 0 = f32 (default; the
                                  grid is bit-identical to the reference's MISE class on the same decoder values), 1 = bf16x6 (f32-equivalent
                                  values; a grid point within rounding of the threshold may fall on the other side), 2 = bf16x3 (reduced) */
    int32_t reserved;          /* 0 */
} ifd_mesh_params;

int ifd_onet_mesh_sample(ifd_ctx* ctx, const float* c, int B, const ifd_mesh_params* prm, float* points,
                         int32_t* n_triangles, float* grid, float* triangles, void* stream);

/* The marching-cubes polygonisation table this library uses (host memory: tri [256*16] edge triples terminated by -1,
 * ntri [256]): csrc/mc_table_data.h, a committed constant table that records what the REFERENCE's compiled libmcubes emits for
 * each of the 256 single-cube sign configurations - edge triples in its order and winding - obtained by running that library
 * (scripts/probe_mc_table.py; fixture tests/golden/mc_table_ref.npz; it is the classic Lorensen-Cline table in libmcubes'
 * numbering).  Nothing is generated at start-up.  Corner / edge numbering as in libmcubes (marchingcubes.h:44-64). */
int ifd_mc_table(int8_t* tri, uint8_t* ntri);

#ifdef __cplusplus
}
#endif
#endif /* IFD_H */
