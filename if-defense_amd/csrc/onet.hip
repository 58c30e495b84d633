// ONet-Opt (ONet/opt_defense.py; BASELINE config #1): the same restoration loop as
// ConvONet-Opt with the Occupancy-Network model - a global 512-d latent code from a PointNet with ResNet blocks
// (im2mesh/encoder/pointnet.py:60-113) and a decoder of five conditional-batch-norm ResNet blocks, hidden 256
// (im2mesh/onet/models/decoder.py:77-133, im2mesh/layers.py:51-107,193-242).
//
// Decoder = 1.31 MMAC per point and pass direction (42x ConvONet's), all of it in ten 256x256 layers: GEMM-shaped,
// MFMA-bound.  One workgroup (8 waves) owns a cloud for all steps, like optimize.hip; per step the 1024 points go
// through the decoder in passes of 128 (16 per wave):
//   * activations live in registers in the accumulator layout of v_mfma_f32_16x16x4_f32 (M = channel, N = point):
//     lane (n = l & 15, q = l >> 4), tile t, register r  <->  channel 16 t + 4 q + r of point n.  With the k-steps of
//     the next layer ordered s = 4 t + r that register IS the B operand of k-step s - no transposes, no LDS traffic.
//   * the 2.6 MB of layer weights (+ their transposes for the backward chain) cannot live in LDS.  They are
//     pre-packed on the host in *fragment order* ([layer][tile t][4 k-steps][lane][4]) so that a 32 KB chunk
//     (2 output tiles x K = 256) is one linear copy, staged global -> LDS with global_load_lds_dwordx4 (no VGPRs),
//     double-buffered, one barrier per chunk; every wave reads each A fragment with one ds_read_b128 per 4 MFMAs.
//     All CUs stream the same 5.2 MB, which stays in L2 / Infinity Cache.
//   * CBN in eval mode is a per-cloud, per-channel affine map a x + b (folded once per cloud by cbn_fold_kernel);
//     ReLU masks are kept as bit-masks; parameters are frozen, so the backward pass is the transposed chain only.
//   * kNN repulsion, Adam and the neighbour lists are the shared code of knn_device.h.
// The encoder is 0.3 % of the FLOPs: plain LDS-tiled f32 MFMA GEMMs (gemm_kernel) + max-pool, one launch per layer.
#include "ifd_device.h"
#include "ifd_internal.h"
#include "knn_device.h"

namespace ifd {

// ---------------------------------------------------------------------------------------------
// encoder: generic row-major GEMM  C[M,N] (+)= act(A[M,K]) W[N,K]^T + bias   on v_mfma_f32_32x32x2_f32
// ---------------------------------------------------------------------------------------------
constexpr int GBM = 128, GBN = 64, GBK = 16, GPAD = GBK + 1;

struct GemmArgs {
    const float* A; int lda;          // [M][lda], K columns used
    const float* W; int ldw;          // [N][ldw], K columns used (pointer may be offset to a column block)
    const float* bias;                // [N] (rows_per_group == 0), [M / rows_per_group][N] otherwise, or nullptr
    float* C; int ldc;
    int M, N, K;
    int rows_per_group;
    int relu_a;                       // apply ReLU to A while staging
    int accumulate;                   // C += result
};

__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ float As[GBM * GPAD];
    __shared__ float Bs[GBN * GPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * GBM, col0 = blockIdx.y * GBN;
    const int m = lane & 31, kh = lane >> 5;
    f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const int ar = tid >> 1, ak = (tid & 1) * 8;                  // A staging: 2 threads per row, 8 floats each
    const int br = tid >> 2, bk = (tid & 3) * 4;                  // W staging: 4 threads per row, 4 floats each
    const float* ap = g.A + (size_t)min(row0 + ar, g.M - 1) * g.lda + ak;
    const float* bp = g.W + (size_t)(col0 + br) * g.ldw + bk;
    for (int k0 = 0; k0 < g.K; k0 += GBK) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + k0), a1 = *reinterpret_cast<const f32x4*>(ap + k0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + k0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            As[ar * GPAD + ak + j] = g.relu_a ? fmaxf(a0[j], 0.f) : a0[j];
            As[ar * GPAD + ak + 4 + j] = g.relu_a ? fmaxf(a1[j], 0.f) : a1[j];
            Bs[br * GPAD + bk + j] = b0[j];
        }
        __syncthreads();
        const float* pa = As + (wave * 32 + m) * GPAD + kh;
        const float* pb = Bs + m * GPAD + kh;
#pragma unroll
        for (int ks = 0; ks < GBK / 2; ++ks) {
            const float av = pa[2 * ks];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pb[2 * ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pb[32 * GPAD + 2 * ks], acc1, 0, 0, 0);
        }
    }
    // D: lane = column n, register r = row (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = col0 + t * 32 + m;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (row < g.M) {
                float v = t ? acc1[r] : acc0[r];
                if (g.bias) v += g.rows_per_group ? g.bias[(size_t)(row / g.rows_per_group) * g.N + col] : g.bias[col];
                float* c = g.C + (size_t)row * g.ldc + col;
                *c = g.accumulate ? *c + v : v;
            }
        }
    }
}

static hipError_t gemm(hipStream_t s, const float* A, int lda, const float* W, int ldw, const float* bias, int rows_per_group,
                       float* C, int ldc, int M, int N, int K, bool relu_a, bool accumulate) {
    GemmArgs g{A, lda, W, ldw, bias, C, ldc, M, N, K, rows_per_group, relu_a ? 1 : 0, accumulate ? 1 : 0};
    hipLaunchKernelGGL(gemm_kernel, dim3((M + GBM - 1) / GBM, N / GBN), dim3(256), 0, s, g);
    return hipGetLastError();
}

// fc_pos (encoder/pointnet.py:89): out[M][1024] = p[M][3] Wpos^T + b
__global__ __launch_bounds__(256) void fc_pos_kernel(const float* __restrict__ p, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ out, int M, int N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int row = (int)(i / N), col = (int)(i % N);
    const float* pp = p + (size_t)row * 3;
    out[i] = fmaf(w[col * 3 + 2], pp[2], fmaf(w[col * 3 + 1], pp[1], fmaf(w[col * 3], pp[0], b[col])));
}

// pool (encoder/pointnet.py:92,96,...,107): max over the valid points of each cloud
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, const int* __restrict__ t_per_cloud,
                                                       int Tmax, int N, float* __restrict__ out) {
    const int b = blockIdx.x;
    const int T = t_per_cloud ? min(t_per_cloud[b], Tmax) : Tmax;
    for (int c = threadIdx.x; c < N; c += 256) {
        float mx = -INFINITY;
        const float* px = x + (size_t)b * Tmax * N + c;
        for (int t = 0; t < T; ++t) mx = fmaxf(mx, px[(size_t)t * N]);
        out[(size_t)b * N + c] = mx;
    }
}

hipError_t launch_onet_encode(const float* w, const OnetEncOffsets& eo, const float* sel, const int* t_per_cloud, int B,
                              int Tmax, float* ws, float* c_out, hipStream_t s) {
    // scratch (floats): x0 [M][1024] | h [M][512] | netA [M][512] | netB [M][512] | pooled, v0, vs [B][512] each
    const int M = B * Tmax, H = ONET_ENC_H;
    float* x0 = ws;
    float* h = x0 + (size_t)M * 2 * H;
    float* netA = h + (size_t)M * H;
    float* netB = netA + (size_t)M * H;
    float* pooled = netB + (size_t)M * H;
    float* v0 = pooled + (size_t)B * H;
    float* vs = v0 + (size_t)B * H;
    hipError_t e;
    hipLaunchKernelGGL(fc_pos_kernel, dim3((unsigned)(((size_t)M * 2 * H + 255) / 256)), dim3(256), 0, s, sel, w + eo.pos_w,
                       w + eo.pos_b, x0, M, 2 * H);
    // block_0 on the 1024-d fc_pos output (layers.py:39-48)
    if ((e = gemm(s, x0, 2 * H, w + eo.fc0_w[0], 2 * H, w + eo.fc0_b[0], 0, h, H, M, H, 2 * H, true, false)) != hipSuccess) return e;
    if ((e = gemm(s, x0, 2 * H, w + eo.sc_w[0], 2 * H, nullptr, 0, netA, H, M, H, 2 * H, false, false)) != hipSuccess) return e;
    if ((e = gemm(s, h, H, w + eo.fc1_w[0], H, w + eo.fc1_b[0], 0, netA, H, M, H, H, true, true)) != hipSuccess) return e;
    for (int i = 1; i < 5; ++i) {
        // cat([net, pooled]) (pointnet.py:92-94): the pooled half of fc_0 / shortcut is one vector per cloud
        hipLaunchKernelGGL(maxpool_kernel, dim3(B), dim3(256), 0, s, netA, t_per_cloud, Tmax, H, pooled);
        if ((e = gemm(s, pooled, H, w + eo.fc0_w[i] + H, 2 * H, w + eo.fc0_b[i], 0, v0, H, B, H, H, true, false)) != hipSuccess) return e;
        if ((e = gemm(s, pooled, H, w + eo.sc_w[i] + H, 2 * H, nullptr, 0, vs, H, B, H, H, false, false)) != hipSuccess) return e;
        if ((e = gemm(s, netA, H, w + eo.fc0_w[i], 2 * H, v0, Tmax, h, H, M, H, H, true, false)) != hipSuccess) return e;
        if ((e = gemm(s, netA, H, w + eo.sc_w[i], 2 * H, vs, Tmax, netB, H, M, H, H, false, false)) != hipSuccess) return e;
        if ((e = gemm(s, h, H, w + eo.fc1_w[i], H, w + eo.fc1_b[i], 0, netB, H, M, H, H, true, true)) != hipSuccess) return e;
        float* t = netA; netA = netB; netB = t;
    }
    hipLaunchKernelGGL(maxpool_kernel, dim3(B), dim3(256), 0, s, netA, t_per_cloud, Tmax, H, pooled);
    return gemm(s, pooled, H, w + eo.fcc_w, H, w + eo.fcc_b, 0, c_out, ONET_C, B, ONET_C, H, true, false);   // fc_c(relu(.))
}

size_t onet_encode_ws_floats(int B, int Tmax) {
    return (size_t)B * Tmax * (2 * ONET_ENC_H + 3 * ONET_ENC_H) + 3 * (size_t)B * ONET_ENC_H;
}

// ---------------------------------------------------------------------------------------------
// CBN fold: gamma / beta (1x1 convs of c, layers.py:234-235) + eval-mode BatchNorm -> a x + b per cloud and channel.
// gb [B][22][256] = {gamma_0, beta_0, gamma_1, ...} (22 small GEMMs); ab [B][11][2][256].  The bias of the fc_0 that
// feeds bn_1 is folded into bn_1's offset: a (h + bias) + b.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cbn_fold_kernel(const float* __restrict__ gb, const float* __restrict__ w,
                                                        OnetDecOffsets od, float* __restrict__ ab) {
    const int b = blockIdx.x, ch = threadIdx.x;
    for (int n = 0; n < ONET_NCBN; ++n) {
        const float gamma = gb[((size_t)b * 2 * ONET_NCBN + 2 * n) * ONET_H + ch];
        const float beta = gb[((size_t)b * 2 * ONET_NCBN + 2 * n + 1) * ONET_H + ch];
        const float a = gamma * (1.0f / sqrtf(w[od.cbn_var[n] + ch] + 1e-5f));
        float off = beta - a * w[od.cbn_mean[n] + ch];
        if (n < 10 && (n & 1)) off = fmaf(a, w[od.fc0_b[n >> 1] + ch], off);       // bn_1 of block n/2 follows fc_0
        ab[(((size_t)b * ONET_NCBN + n) * 2 + 0) * ONET_H + ch] = a;
        ab[(((size_t)b * ONET_NCBN + n) * 2 + 1) * ONET_H + ch] = off;
    }
}

hipError_t launch_onet_cbn(const float* w, const OnetDecOffsets& od, const float* c, int B, float* gb, float* ab, hipStream_t s) {
    for (int n = 0; n < ONET_NCBN; ++n) {
        hipError_t e = gemm(s, c, ONET_C, w + od.cbn_gamma_w[n], ONET_C, w + od.cbn_gamma_b[n], 0, gb + (size_t)(2 * n) * ONET_H,
                            2 * ONET_NCBN * ONET_H, B, ONET_H, ONET_C, false, false);
        if (e != hipSuccess) return e;
        e = gemm(s, c, ONET_C, w + od.cbn_beta_w[n], ONET_C, w + od.cbn_beta_b[n], 0, gb + (size_t)(2 * n + 1) * ONET_H,
                 2 * ONET_NCBN * ONET_H, B, ONET_H, ONET_C, false, false);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(cbn_fold_kernel, dim3(B), dim3(ONET_H), 0, s, gb, w, od, ab);
    return hipGetLastError();
}

#include "onet_kernel.h"


__global__ __launch_bounds__(OPT_THREADS, 2) void onet_decode_kernel(const float* __restrict__ img, const float* __restrict__ small,
                                                                      const float* __restrict__ ab, const float* __restrict__ p,
                                                                      int K, float* __restrict__ logits,
                                                                      float* __restrict__ dlogit_dp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    onet_prologue(img, small, ab + (size_t)cloud * ONET_NCBN * 2 * ONET_H, smem, tid, OPT_THREADS, wave, lane);
    const float* pc = p + (size_t)cloud * K * 3;
    const int npass = (K + 127) >> 7;
    for (int g = 0; g < npass; ++g) {
        const int pt = g * 128 + wave * 16 + (lane & 15), tp = min(pt, K - 1);
        float logit, bce, dx[3];
        if (dlogit_dp != nullptr)
            onet_pass<OMODE_SUM, true>(img, smem, wave, lane, pc[3 * tp], pc[3 * tp + 1], pc[3 * tp + 2], 0.f, 1.f, logit, bce, dx);
        else
            onet_pass<OMODE_SUM, false>(img, smem, wave, lane, pc[3 * tp], pc[3 * tp + 1], pc[3 * tp + 2], 0.f, 1.f, logit, bce, dx);
        if (lane < 16 && pt < K) {
            logits[(size_t)cloud * K + pt] = logit;
            if (dlogit_dp != nullptr) {
                float* o = dlogit_dp + ((size_t)cloud * K + pt) * 3;
                o[0] = dx[0]; o[1] = dx[1]; o[2] = dx[2];
            }
        }
    }
}

hipError_t launch_onet_grid_eval(const float* img, const float* small, const float* ab, const MiseGrid& g, int B, int n_blocks,
                                 float box, hipStream_t s) {
    hipLaunchKernelGGL(grid_plan_kernel<0>, dim3(1), dim3(64), 0, s, g, B);
    hipLaunchKernelGGL(onet_grid_eval_kernel<0>, dim3(n_blocks), dim3(OPT_THREADS), ONET_DEC_LDS, s, img, small, ab, g, B, box);
    return hipGetLastError();
}

// ---- clouds of more than MAXK optimised points (--sample_npoint up to LARGE_MAXK; ONet/opt_defense.py:27 has no limit) --
// The persistent kernel keeps a cloud's optimiser state in one CU's LDS (1024 points).  Larger clouds take two launches per
// Adam step, like ConvONet's (optimize.hip, "large" section): this kernel - the decoder pass of onet_optimize_kernel, BCE seed
// included, spread over (cloud, part) workgroups - writes every point's occupancy gradient; large_step_kernel (shared) does
// the exact brute-force 5-NN, the repulsion terms and Adam.  Same arithmetic term by term, no certified lists.
__global__ __launch_bounds__(OPT_THREADS, 2) void onet_large_occupancy_kernel(
    const float* __restrict__ img, const float* __restrict__ small, const float* __restrict__ ab, const float* __restrict__ p,
    int K, const int32_t* __restrict__ loss_batch_per_cloud, int loss_batch, float thr, f32x4* __restrict__ G) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    onet_prologue(img, small, ab + (size_t)cloud * ONET_NCBN * 2 * ONET_H, smem, tid, OPT_THREADS, wave, lane);
    const float* pc = p + (size_t)cloud * K * 3;
    const int lb = loss_batch_per_cloud ? loss_batch_per_cloud[cloud] : loss_batch;
    const float inv_lb = 1.0f / (float)lb;
    const int npass = (K + 127) >> 7;
    for (int g = blockIdx.y; g < npass; g += gridDim.y) {          // (block-uniform trip count: onet_pass syncs the block)
        const int pt = g * 128 + wave * 16 + (lane & 15), tp = min(pt, K - 1);
        float logit, bce, dx[3];
        onet_pass<OMODE_OPT, true>(img, smem, wave, lane, pc[3 * tp], pc[3 * tp + 1], pc[3 * tp + 2], thr, inv_lb, logit, bce, dx);
        if (lane < 16 && pt < K) G[(size_t)cloud * K + pt] = f32x4{dx[0], dx[1], dx[2], bce};
    }
}

// (a.precision != 0: img = the bf16 piece image, the occupancy half on onet_bf.hip's split-precision pass)
hipError_t launch_onet_large_optimize(const float* img, const float* small, const float* ab, float* p, float* m, float* v,
                                      float* loss, const int32_t* loss_batch_per_cloud, void* ws, unsigned long long* counters,
                                      const float* adam_tab, int B, int K, const OptArgs& a, hipStream_t s) {
    f32x4* G = static_cast<f32x4*>(ws);
    void* f_ws = nullptr;                  // global repulsion accumulators beyond LARGE_LDS_MAXK points (end of ws)
    void* list_ws = large_list_ws(ws, B, K, m == nullptr);      // certified neighbour lists up to LARGE_LDS_MAXK points (optimize.hip)
    {
        hipError_t e = large_f_prepare(ws, B, K, m == nullptr, &f_ws, s);
        if (e != hipSuccess) return e;
    }
    if (m == nullptr) {                    // own moments behind G (large_ws_bytes), zeroed
        m = reinterpret_cast<float*>(G + (size_t)B * K);
        v = m + (size_t)B * K * 3;
        hipError_t e = hipMemsetAsync(m, 0, (size_t)B * K * 3 * 4 * 2, s);
        if (e != hipSuccess) return e;
    }
    const int npass = (K + 127) >> 7;
    const int parts = B >= 256 ? 1 : min(npass, max(1, 512 / B));        // enough workgroups to fill the GPU with few clouds
    for (int step = 0; step < a.steps; ++step) {
        hipError_t e = hipSuccess;
        if (a.precision != 0)
            e = launch_onet_large_occupancy_bf(a.precision, img, small, ab, p, B, parts, K, loss_batch_per_cloud, a.loss_batch, a.threshold, G, s);
        else
            hipLaunchKernelGGL(onet_large_occupancy_kernel, dim3(B, parts), dim3(OPT_THREADS), ONET_DEC_LDS, s, img, small, ab, p, K,
                               loss_batch_per_cloud, a.loss_batch, a.threshold, G);
        if (e != hipSuccess) return e;
        e = launch_large_step(p, m, v, G, B, K, adam_tab, step, loss_batch_per_cloud, a,
                                         step == a.steps - 1 ? loss : nullptr, f_ws, list_ws, counters, s);
        if (e != hipSuccess) return e;
    }
    return a.normalize ? launch_large_normalize(p, B, K, s) : hipGetLastError();
}

hipError_t configure_onet_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(onet_decode_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ONET_DEC_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(onet_large_occupancy_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)ONET_DEC_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(onet_grid_eval_kernel<0>),          // (per device: see onet_bf.hip)
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)ONET_DEC_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(onet_optimize_kernel<0>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ONET_OPT_LDS);
}

hipError_t launch_onet_decode(const float* img, const float* small, const float* ab, const float* p, int B, int K,
                              float* logits, float* dlogit_dp, hipStream_t s) {
    hipLaunchKernelGGL(onet_decode_kernel, dim3(B), dim3(OPT_THREADS), ONET_DEC_LDS, s, img, small, ab, p, K, logits, dlogit_dp);
    return hipGetLastError();
}

hipError_t launch_onet_optimize(const float* img, const float* small, const float* ab, float* p, float* m, float* v, float* loss,
                                const int32_t* loss_batch_per_cloud, uint16_t* knn_lists, unsigned long long* counters,
                                const float* adam_tab, int B, int K, const OptArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(onet_optimize_kernel<0>, dim3(B), dim3(OPT_THREADS), ONET_OPT_LDS, s, img, small, ab, p, m, v, loss,
                       loss_batch_per_cloud, knn_lists, counters, adam_tab, K, a);
    return hipGetLastError();
}

int onet_small_floats() { return ONET_SMALL_FLOATS; }

}  // namespace ifd
