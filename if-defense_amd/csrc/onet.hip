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

// ---------------------------------------------------------------------------------------------
// decoder pass: 128 points (16 per wave) forward + input-backward through the ten 256x256 layers
// ---------------------------------------------------------------------------------------------
constexpr int CHUNK_FLOATS = 2 * 64 * 64;            // 2 output tiles x 64 k-steps x 64 lanes = 32 KB
constexpr int CHUNKS_PER_LAYER = 8;
constexpr int N_IMG = 20;                            // fwd fc_0/fc_1 of 5 blocks, then their transposes in backward order
constexpr int N_CHUNKS = N_IMG * CHUNKS_PER_LAYER;   // 160 per pass
// LDS layout (floats)
constexpr int OL_CHUNK = 0;                                          // [2][CHUNK_FLOATS]
constexpr int OL_AB = OL_CHUNK + 2 * CHUNK_FLOATS;                   // [11][2][256]  CBN a, b of this cloud
constexpr int OL_FCP = OL_AB + ONET_NCBN * 2 * ONET_H;               // [256][4]      fc_p {w0, w1, w2, bias}
constexpr int OL_B1 = OL_FCP + ONET_H * 4;                           // [5][256]      fc_1 biases
constexpr int OL_WOUT = OL_B1 + 5 * ONET_H;                          // [256] + bout (+3 pad)
constexpr int OL_END = OL_WOUT + ONET_H + 4;
constexpr int ONET_SMALL_FLOATS = OL_END - OL_FCP;                   // what the host packs behind the CBN block

__device__ __forceinline__ void stage_chunk(const float* __restrict__ img, float* __restrict__ lds, int g, int wave, int lane) {
    // chunk g of the circular weight stream -> buffer g & 1; 32 pieces of 1 KiB, 4 per wave
    const float* src = img + (size_t)g * CHUNK_FLOATS + wave * 1024 + lane * 4;
    float* dst = lds + OL_CHUNK + (g & 1) * CHUNK_FLOATS + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                         (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
}

// One 256x256 layer on the wave's 16 points: out[t] (+)= sum_s A_frag(t, s) * in[s].  g = index of the layer's first
// chunk in the weight stream; chunk g is resident in LDS on entry, chunk g + 8 on exit.
__device__ __forceinline__ void layer_mfma(const float* __restrict__ img, float* __restrict__ lds, int g, int wave, int lane,
                                           const float (&in)[64], f32x4 (&out)[16]) {
#pragma unroll
    for (int c = 0; c < CHUNKS_PER_LAYER; ++c) {
        int gn = g + c + 1;
        if (gn >= N_CHUNKS) gn -= N_CHUNKS;
        stage_chunk(img, lds, gn, wave, lane);                       // prefetch the next chunk into the other buffer
        const f32x4* a = reinterpret_cast<const f32x4*>(lds + OL_CHUNK + ((g + c) & 1) * CHUNK_FLOATS) + lane;
        // A fragments one k-group ahead; the sched_barrier keeps the scheduler from hoisting all 32 fragment reads of
        // the chunk to its top (128 VGPRs -> spills)
        f32x4 a0 = a[0], a1 = a[16 * 64];
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            f32x4 n0 = a0, n1 = a1;
            if (s4 + 1 < 16) { n0 = a[(s4 + 1) * 64]; n1 = a[(16 + s4 + 1) * 64]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                out[2 * c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], in[4 * s4 + j], out[2 * c], 0, 0, 0);
                out[2 * c + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], in[4 * s4 + j], out[2 * c + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0; a1 = n1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // my pieces of the next chunk have landed
        __builtin_amdgcn_s_barrier();                                // everybody's have, and everybody is done with this one
    }
}

// ReLU masks of the 64 channels a lane holds: bit (31 - k % 32) of word k / 32 = "value k is NOT positive".  One
// v_alignbit per value going in (the sign of bits(relu(v)) - 1), v_bfe_i32 + v_bfi per value coming out - every
// vector instruction costs SIMD time next to the MFMAs.
struct Mask64 {
    unsigned int w[2];
};
// u = relu(a x + b) for the 64 channels this lane holds; returns the ReLU mask
__device__ __forceinline__ Mask64 cbn_relu(const float* __restrict__ ab, int q, const f32x4 (&x)[16], float (&u)[64]) {
    Mask64 m = {{0u, 0u}};
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ab + 16 * t + 4 * q);
        const f32x4 b = *reinterpret_cast<const f32x4*>(ab + ONET_H + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = fmaf(a[r], x[t][r], b[r]);
            const int ri = max(__float_as_int(v), 0);                  // ReLU on the bits (-0.0 and negatives -> +0)
            u[4 * t + r] = __int_as_float(ri);
            m.w[(4 * t + r) >> 5] = __builtin_amdgcn_alignbit(m.w[(4 * t + r) >> 5], (unsigned int)ri - 1u, 31);
        }
    }
    asm("" : "+v"(m.w[0]), "+v"(m.w[1]));     // keep them bit-masks (not volatile: dropped with the masks in forward-only passes)
    return m;
}
// val where value k passed its ReLU, else +0
__device__ __forceinline__ float relu_gate(const Mask64& m, int k, float val) {
    const int dead = (int)(m.w[k >> 5] << (k & 31)) >> 31;            // v_bfe_i32: 0 / -1
    unsigned int o;
    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(o) : "v"(dead), "v"(__float_as_uint(val)));      // val & ~dead
    return __uint_as_float(o);
}

constexpr int OMODE_SUM = 0;      // d(sum of logits)/dp, loss = logits          (ifd_onet_decode)
constexpr int OMODE_OPT = 1;      // BCE-with-logits against `threshold`, scaled by inv_lb  (optimiser)

// Forward + backward of one 16-point sub-tile per wave.  (x0, x1, x2) = the point of lane n (all four q-lanes of
// a point pass the same coordinates).  All 8 waves of the block call this together (it contains barriers).
template <int MODE, bool WANT_GRAD>
__device__ __forceinline__ void onet_pass(const float* __restrict__ img, float* __restrict__ lds, int wave, int lane,
                                          float x0, float x1, float x2, float threshold, float inv_lb, float& logit,
                                          float& bce, float (&dx)[3]) {
    const int q = lane >> 4;
    const float* ab = lds + OL_AB;
    f32x4 x[16], acc[16];
    float u[64];
    Mask64 m0[5], m1[5], mf;
    // fc_p (decoder.py:118)
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(lds + OL_FCP + (16 * t + 4 * q + r) * 4);
            x[t][r] = fmaf(w.z, x2, fmaf(w.y, x1, fmaf(w.x, x0, w.w)));
        }
    // five CResnetBlockConv1d (layers.py:97-107): x += fc_1(relu(bn_1(fc_0(relu(bn_0(x))))))
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        m0[i] = cbn_relu(ab + (2 * i) * 2 * ONET_H, q, x, u);
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};          // fc_0 bias is folded into bn_1
        layer_mfma(img, lds, (2 * i) * CHUNKS_PER_LAYER, wave, lane, u, acc);
        m1[i] = cbn_relu(ab + (2 * i + 1) * 2 * ONET_H, q, acc, u);
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] += *reinterpret_cast<const f32x4*>(lds + OL_B1 + i * ONET_H + 16 * t + 4 * q);
        layer_mfma(img, lds, (2 * i + 1) * CHUNKS_PER_LAYER, wave, lane, u, x);
    }
    // fc_out(relu(bn(x)))  (decoder.py:130)
    mf = cbn_relu(ab + 10 * 2 * ONET_H, q, x, u);
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(lds + OL_WOUT + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(w[r], u[4 * t + r], part);
    }
    part = add_lane_xor32(add_lane_xor16(part));
    logit = part + lds[OL_WOUT + ONET_H];
    float dl;
    if (MODE == OMODE_OPT) {
        // F.binary_cross_entropy_with_logits (opt_defense.py:213): max(x,0) - x t + log1p(exp(-|x|))
        const float e = expf(-fabsf(logit));
        bce = fmaxf(logit, 0.f) - logit * threshold + log1pf(e);
        const float sig = logit >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        dl = (sig - threshold) * inv_lb;
    } else {
        bce = logit;
        dl = 1.f;
    }
    dx[0] = dx[1] = dx[2] = 0.f;
    if (!WANT_GRAD) {
        // the next pass starts at chunk 0 again: skip the ten backward images (chunk 80, staged by the last layer,
        // is not read by anybody)
        stage_chunk(img, lds, 0, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        return;
    }
    // ---- backward (parameters frozen: only d/dx).  gx = dL/dx of the residual stream, in x[] ---------------------
    {
        const float* a = ab + 10 * 2 * ONET_H;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a + 16 * t + 4 * q);
            const f32x4 w = *reinterpret_cast<const f32x4*>(lds + OL_WOUT + 16 * t + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[t][r] = relu_gate(mf, 4 * t + r, dl * w[r] * av[r]);
        }
    }
#pragma unroll
    for (int i = 4; i >= 0; --i) {
        const int gi = 10 + 2 * (4 - i);                                           // image index of fc_1[i]^T
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) u[4 * t + r] = x[t][r];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        layer_mfma(img, lds, gi * CHUNKS_PER_LAYER, wave, lane, u, acc);          // g_u2 = W1^T gx
        {
            const float* a = ab + (2 * i + 1) * 2 * ONET_H;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(a + 16 * t + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) u[4 * t + r] = relu_gate(m1[i], 4 * t + r, acc[t][r] * av[r]);
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        layer_mfma(img, lds, (gi + 1) * CHUNKS_PER_LAYER, wave, lane, u, acc);    // g_u = W0^T g_h
        {
            const float* a = ab + (2 * i) * 2 * ONET_H;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(a + 16 * t + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    x[t][r] += relu_gate(m0[i], 4 * t + r, acc[t][r] * av[r]);
            }
        }
    }
    // d/dp through fc_p
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(lds + OL_FCP + (16 * t + 4 * q + r) * 4);
            g0 = fmaf(w.x, x[t][r], g0); g1 = fmaf(w.y, x[t][r], g1); g2 = fmaf(w.z, x[t][r], g2);
        }
    g0 = add_lane_xor32(add_lane_xor16(g0));
    g1 = add_lane_xor32(add_lane_xor16(g1));
    g2 = add_lane_xor32(add_lane_xor16(g2));
    dx[0] = g0; dx[1] = g1; dx[2] = g2;
}

// per-cloud prologue: CBN a/b + the small parameters into LDS, first weight chunk in flight and landed
__device__ __forceinline__ void onet_prologue(const float* __restrict__ img, const float* __restrict__ small,
                                              const float* __restrict__ ab_cloud, float* __restrict__ lds, int tid, int nthreads,
                                              int wave, int lane) {
    for (int i = tid; i < ONET_NCBN * 2 * ONET_H; i += nthreads) lds[OL_AB + i] = ab_cloud[i];
    for (int i = tid; i < ONET_SMALL_FLOATS; i += nthreads) lds[OL_FCP + i] = small[i];
    stage_chunk(img, lds, 0, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

constexpr size_t ONET_DEC_LDS = (size_t)OL_END * 4;
constexpr int OL_G = OL_END;                                        // optimiser state behind the decoder's
constexpr size_t ONET_OPT_LDS = ONET_DEC_LDS + MAXK * 16 * 2 + 16 + MAXK * 12 + 128 * 4;
static_assert(ONET_OPT_LDS <= 160 * 1024, "LDS budget");

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

// Occupancy at the queued grid points of the MISE rounds (mesh.hip; generation.py:112-127 eval_points), forward only.
// DEVICE-DRIVEN (round 5): the host does not know the queue lengths.  grid_plan_kernel turns the clouds' counts into an exclusive
// prefix of 128-point decoder passes; the evaluation kernel is a fixed launch of one workgroup per CU (98 KB of LDS: one fits),
// workgroup j takes the contiguous range [j T / G, (j + 1) T / G) of the round's T passes - whatever cloud they belong to; the
// cloud's folded CBN coefficients are re-read into LDS when the range crosses into the next cloud.  Rounds 1-4 launched
// (segment of 2048 points, cloud) blocks sized by a max count the host read back every round: a round of e.g. 1152 blocks on 256
// CUs ran as 4.5 waves of 5 ms each and the last wave was half empty (onet_grid_eval_kernel 0.72 of the f32-MFMA peak against
// 0.86 for the optimiser's passes on the same code).
__global__ void grid_plan_kernel(MiseGrid g, int B) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int acc = 0;
    for (int b = 0; b < B; ++b) {
        g.plan[b] = acc;
        acc += (min(g.count[b], g.cap) + 127) >> 7;
    }
    g.plan[B] = acc;
}

__global__ __launch_bounds__(OPT_THREADS, 2) void onet_grid_eval_kernel(const float* __restrict__ img, const float* __restrict__ small,
                                                                         const float* __restrict__ ab, MiseGrid g, int B, float box) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long T = g.plan[B];
    int pass = (int)((T * blockIdx.x) / gridDim.x);
    const int pass_end = (int)((T * (blockIdx.x + 1)) / gridDim.x);
    if (pass >= pass_end) return;                                              // block-uniform
    // the cloud of the first pass: largest b with plan[b] <= pass (clouds without points have empty ranges and are stepped over)
    int lo = 0, hi = B - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (g.plan[mid] <= pass) lo = mid; else hi = mid - 1;
    }
    int cloud = lo;
    const float inv_r = (float)(g.P - 1);
    bool first = true;
    while (pass < pass_end) {
        while (g.plan[cloud + 1] <= pass) ++cloud;                             // (uniform; at most B steps over the whole range)
        const float* abc = ab + (size_t)cloud * ONET_NCBN * 2 * ONET_H;
        if (first) {
            onet_prologue(img, small, abc, smem, tid, OPT_THREADS, wave, lane);
            first = false;
        } else {
            // every wave has left the previous cloud's last pass (it ends in a workgroup barrier): replace the CBN block only
            for (int i = tid; i < ONET_NCBN * 2 * ONET_H; i += OPT_THREADS) smem[OL_AB + i] = abc[i];
            __syncthreads();
        }
        const int n = min(g.count[cloud], g.cap);
        const int* list = g.list + (size_t)cloud * g.cap;
        const int stop = min(pass_end, g.plan[cloud + 1]);
        for (; pass < stop; ++pass) {
            const int base = (pass - g.plan[cloud]) * 128;
            const int i = base + wave * 16 + (lane & 15);
            const int idx = list[min(i, n - 1)];
            const int x = idx / (g.P * g.P), y = (idx / g.P) % g.P, z = idx % g.P;
            // pointsf / resolution, box_size * (pointsf - 0.5) in float32 (generation.py:117-121)
            const float px = ((float)x / inv_r - 0.5f) * box, py = ((float)y / inv_r - 0.5f) * box, pz = ((float)z / inv_r - 0.5f) * box;
            float logit, bce, dx[3];
            onet_pass<OMODE_SUM, false>(img, smem, wave, lane, px, py, pz, 0.f, 1.f, logit, bce, dx);
            if (lane < 16 && i < n) {
                g.val[(size_t)cloud * g.P3 + idx] = logit;
                g.known[(size_t)cloud * g.P3 + idx] = 1;
            }
        }
    }
}

hipError_t launch_onet_grid_eval(const float* img, const float* small, const float* ab, const MiseGrid& g, int B, int n_blocks,
                                 float box, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(onet_grid_eval_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ONET_DEC_LDS);
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(grid_plan_kernel, dim3(1), dim3(64), 0, s, g, B);
    hipLaunchKernelGGL(onet_grid_eval_kernel, dim3(n_blocks), dim3(OPT_THREADS), ONET_DEC_LDS, s, img, small, ab, g, B, box);
    return hipGetLastError();
}

// The ONet-Opt optimiser (ONet/opt_defense.py:182-239): same skeleton as optimize_kernel, decoder passes instead
// of plane tiles.  The kNN phase is ~1 % of a step here, so all waves simply run it first.
__global__ __launch_bounds__(OPT_THREADS, 2) void onet_optimize_kernel(
    const float* __restrict__ img, const float* __restrict__ small, const float* __restrict__ ab, float* __restrict__ p,
    float* __restrict__ m_io, float* __restrict__ v_io, float* __restrict__ loss_out,
    const int32_t* __restrict__ loss_batch_per_cloud, uint16_t* knn_lists, unsigned long long* __restrict__ counters,
    const float* __restrict__ adam_tab, int K, OptArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* G = reinterpret_cast<f32x4*>(smem + OL_G);                // occupancy gradient (+ BCE term in .w)
    f32x4* X = G + MAXK;                                             // current points; X[MAXK] = far-away dummy
    const RepAcc F = {reinterpret_cast<long long*>(X + MAXK + 1),    // fixed-point repulsion-gradient scatter (knn_device.h)
                      reinterpret_cast<int*>(reinterpret_cast<long long*>(X + MAXK + 1) + MAXK)};
    float* scratch = reinterpret_cast<float*>(F.z + MAXK);           // 128 floats
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* pc = p + (size_t)cloud * K * 3;
    const int pa = tid, pb = tid + OPT_THREADS;
    const unsigned long long t_begin = __builtin_readcyclecounter();

    AdamState ast;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pt = q ? pb : pa;
#pragma unroll
        for (int a = 0; a < 3; ++a) ast.mm[3 * q + a] = ast.vv[3 * q + a] = 0.f;
        if (pt < K) {
            X[pt] = f32x4{pc[3 * pt], pc[3 * pt + 1], pc[3 * pt + 2], 0.f};
            if (A.t0 > 0 && m_io != nullptr) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    ast.mm[3 * q + a] = m_io[((size_t)cloud * K + pt) * 3 + a];
                    ast.vv[3 * q + a] = v_io[((size_t)cloud * K + pt) * 3 + a];
                }
            }
        }
    }
    for (int i = tid; i < MAXK; i += OPT_THREADS) { F.xy[i] = 0; F.z[i] = 0; }
    if (tid == 0) X[MAXK] = f32x4{1e18f, 1e18f, 1e18f, 0.f};

    const RepConst rc = {A.rep_radius, A.rep_h, A.rep_eps};
    const int loss_batch = loss_batch_per_cloud ? loss_batch_per_cloud[cloud] : A.loss_batch;
    const float inv_lb = 1.0f / (float)loss_batch;
    const float rep_scale = A.rep_weight / ((float)loss_batch * (float)K * 5.f);
    const bool use_rep = A.rep_weight > 0.f;
    float rep_loss_a = 0.f, rep_loss_b = 0.f;
    uint16_t* La = knn_lists + ((size_t)cloud * MAXK + (pa & (MAXK - 1))) * LIST_M;
    uint16_t* Lb = knn_lists + ((size_t)cloud * MAXK + (pb & (MAXK - 1))) * LIST_M;
    KnnPt ka = {0, -1, 0.f, 0.f, 2.0f, 7.0f, f32x4{0.f, 0.f, 0.f, 0.f}, 0.f, 0.f, false, false};
    KnnPt kb = ka;
    uint16_t* cloud_lists = knn_lists + (size_t)cloud * MAXK * LIST_M;
    float* dmaxbuf = scratch + 32;
    float* movebuf = scratch + 64;
    volatile int* rebuild_flag = reinterpret_cast<volatile int*>(scratch + 28);
    unsigned int* lcnt = reinterpret_cast<unsigned int*>(scratch + 96);           // [CN_COUNT] event counters
    KnnCounters cn{lcnt, lane};
    const KnnShared ksh = {dmaxbuf, movebuf, rebuild_flag};
    if (tid < 2) rebuild_flag[tid] = 0;
    if (tid < CN_COUNT) lcnt[tid] = 0u;
    if (tid < 2 * MAX_WAVES) { dmaxbuf[tid] = 0.f; movebuf[tid] = 2.f * A.lr; }
    onet_prologue(img, small, ab + (size_t)cloud * ONET_NCBN * 2 * ONET_H, smem, tid, OPT_THREADS, wave, lane);   // syncs

    const int npass = (K + 127) >> 7;
    for (int step = 0; step < A.steps; ++step) {
        const bool last = step == A.steps - 1;
        if (use_rep)
            knn_phase(X, F, K, pa, pb, wave, lane, step, last, A.knn_scan_every_step, La, Lb, cloud_lists, ka, kb, ksh,
                      rc, rep_loss_a, rep_loss_b, cn);
#pragma unroll 1
        for (int g = 0; g < npass; ++g) {
            const int pt = g * 128 + wave * 16 + (lane & 15), tp = min(pt, K - 1);
            const f32x4 x = X[tp];
            float logit, bce, dx[3];
            onet_pass<OMODE_OPT, true>(img, smem, wave, lane, x.x, x.y, x.z, A.threshold, inv_lb, logit, bce, dx);
            if (lane < 16 && pt < K) G[tp] = f32x4{dx[0], dx[1], dx[2], bce};
        }
        __syncthreads();
        if (last && loss_out != nullptr) {   // losses at the pre-update points of the last step
            float occ = (pa < K ? G[pa].w : 0.f) + (pb < K ? G[pb].w : 0.f);
            float rep = (pa < K ? rep_loss_a : 0.f) + (pb < K ? rep_loss_b : 0.f);
            occ = wave_sum(occ);
            rep = wave_sum(rep);
            if (lane == 0) { scratch[wave] = occ; scratch[MAX_WAVES + wave] = rep; }
            __syncthreads();
            if (tid == 0) {
                float so = 0.f, sr = 0.f;
                for (int w = 0; w < OPT_THREADS / 64; ++w) { so += scratch[w]; sr += scratch[MAX_WAVES + w]; }
                loss_out[2 * cloud + 0] = so;
                loss_out[2 * cloud + 1] = sr / ((float)K * 5.f);
            }
        }
        float xnew[2][3], mv2;
        adam_phase(X, G, F, K, pa, pb, adam_tab[2 * step], adam_tab[2 * step + 1], rep_scale, ast, xnew, mv2, counters);
        adam_displacement(K, pa, pb, wave, lane, step, xnew, mv2, ka, kb, ksh);
        if (tid == 0) rebuild_flag[step & 1] = 0;
        __syncthreads();
    }

    if (counters != nullptr) {
        if (tid < CN_COUNT) {
            constexpr int SLOT[CN_COUNT] = {0, 1, 2, 4, 5, 6, 7};
            atomicAdd(counters + SLOT[tid], (unsigned long long)lcnt[tid]);
        }
        if (tid == 0 && cloud == 0) counters[3] = __builtin_readcyclecounter() - t_begin;
    }
    if (A.normalize) normalize_in_lds(X, K, scratch);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pt = q ? pb : pa;
        if (pt < K) {
            const f32x4 x = X[pt];
            pc[3 * pt] = x.x; pc[3 * pt + 1] = x.y; pc[3 * pt + 2] = x.z;
            if (m_io != nullptr) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    m_io[((size_t)cloud * K + pt) * 3 + a] = ast.mm[3 * q + a];
                    v_io[((size_t)cloud * K + pt) * 3 + a] = ast.vv[3 * q + a];
                }
            }
        }
    }
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

hipError_t launch_onet_large_optimize(const float* img, const float* small, const float* ab, float* p, float* m, float* v,
                                      float* loss, const int32_t* loss_batch_per_cloud, void* ws, unsigned long long* counters,
                                      const float* adam_tab, int B, int K, const OptArgs& a, hipStream_t s) {
    f32x4* G = static_cast<f32x4*>(ws);
    void* f_ws = nullptr;                  // global repulsion accumulators beyond LARGE_LDS_MAXK points (end of ws)
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
        hipLaunchKernelGGL(onet_large_occupancy_kernel, dim3(B, parts), dim3(OPT_THREADS), ONET_DEC_LDS, s, img, small, ab, p, K,
                           loss_batch_per_cloud, a.loss_batch, a.threshold, G);
        hipError_t e = launch_large_step(p, m, v, G, B, K, adam_tab, step, loss_batch_per_cloud, a,
                                         step == a.steps - 1 ? loss : nullptr, f_ws, counters, s);
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
    return hipFuncSetAttribute(reinterpret_cast<const void*>(onet_optimize_kernel),
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
    hipLaunchKernelGGL(onet_optimize_kernel, dim3(B), dim3(OPT_THREADS), ONET_OPT_LDS, s, img, small, ab, p, m, v, loss,
                       loss_batch_per_cloud, knn_lists, counters, adam_tab, K, a);
    return hipGetLastError();
}

int onet_small_floats() { return ONET_SMALL_FLOATS; }

}  // namespace ifd
