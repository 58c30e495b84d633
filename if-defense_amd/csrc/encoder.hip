// ConvONet encoder, point-wise half (LocalPoolPointnet.forward, ConvONet/src/encoder/pointnet.py:124-168):
// fc_pos -> 5 ResnetBlockFC with local max-pooling over the three 64x64 planes -> fc_c -> scatter-mean of
// the point features into the three (channel-last) planes that feed the U-Net.
//
// One workgroup per cloud, one thread per input point (T <= 1024), everything between the xyz read and the
// plane write stays in LDS / registers:
//   * pool_local (pointnet.py:104-122) never materialises the [32,4096] scatter_max grids: a point's pooled
//     feature is the channel-wise max over the points that share its cell, reached through a per-plane ring of
//     the cell's points built once per cloud (cells are sparsely occupied: ~600 points over ~1700 occupied cells).
//   * scatter_mean (pointnet.py:75-80) is evaluated gather-side in ascending point order by the first point
//     of each cell: deterministic (no float atomics) and the same summation order as an index-ordered
//     scatter_add.  The plane buffer must be zero-filled by the caller (empty cells stay 0).
//   * the weights are wave-uniform -> scalar loads + SGPR-operand FMAs, no LDS traffic.  (Staging each block's 20 KB
//     of weights in LDS and reading them back with uniform-address loads was measured: 5.4 ms against 3.6 ms.)
#include "ifd_device.h"
#include "ifd_internal.h"

namespace ifd {

constexpr int NET_STRIDE = 33;          // LDS row stride of the [T][32] feature matrix (bank-conflict pad)

// out[o] = b[o] + sum_k W[o][k] * in[k]   (W row-major [NOUT][NIN], uniform address -> s_load)
template <int NOUT, int NIN, bool RELU_IN, bool HAS_BIAS>
__device__ __forceinline__ void linear(const float* __restrict__ W, const float* __restrict__ b, const float (&in)[NIN],
                                       float (&out)[NOUT]) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float acc = HAS_BIAS ? b[o] : 0.f;
#pragma unroll
        for (int k = 0; k < NIN; ++k) acc = fmaf(W[o * NIN + k], RELU_IN ? fmaxf(in[k], 0.f) : in[k], acc);
        out[o] = acc;
    }
}

// ResnetBlockFC(64 -> 32, hidden 32) (src/layers.py:39-48): x_s + fc_1(relu(fc_0(relu(x))))
__device__ __forceinline__ void resnet_block(const float* __restrict__ w, const EncPointOffsets& eo, int blk,
                                             const float (&x)[64], float (&out)[32]) {
    float h[32], dx[32], xs[32];
    linear<32, 64, true, true>(w + eo.fc0_w[blk], w + eo.fc0_b[blk], x, h);
    linear<32, 32, true, true>(w + eo.fc1_w[blk], w + eo.fc1_b[blk], h, dx);
    linear<32, 64, false, false>(w + eo.sc_w[blk], nullptr, x, xs);
#pragma unroll
    for (int o = 0; o < 32; ++o) out[o] = xs[o] + dx[o];
}

// [threads][33] features + [3][threads] cell rings; the ring construction's scratch (head table int[4096] + unordered
// chains u16[3][threads] = 22,528 B at 1024 threads) aliases the feature matrix, which is first written after it
constexpr size_t enc_lds(int threads) { return (size_t)threads * NET_STRIDE * 4 + 3 * (size_t)threads * 2; }   // 141,312 B @1024

template <int ENC_THREADS>
__global__ __launch_bounds__(ENC_THREADS) void encode_points_kernel(const float* __restrict__ w, EncPointOffsets eo,
                                                                     const float* __restrict__ sel,
                                                                     const int* __restrict__ t_per_cloud, int Tmax,
                                                                     float* __restrict__ planes,
                                                                     float* __restrict__ c_out, DecConst dc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* net = smem;                                                     // [ENC_THREADS][NET_STRIDE]
    unsigned short* ring = reinterpret_cast<unsigned short*>(net + ENC_THREADS * NET_STRIDE);
                                                                           // [3][ENC_THREADS] next point of my cell

    const int b = blockIdx.x, i = threadIdx.x;
    const int T = t_per_cloud ? min(t_per_cloud[b], Tmax) : Tmax;
    const bool live = i < T;
    const float* ps = sel + ((size_t)b * Tmax + (live ? i : 0)) * 3;
    const float p[3] = {ps[0], ps[1], ps[2]};

    // normalize_coordinate + coordinate2index (src/common.py:250-257, 309-311): cell = int(u0*64) + 64*int(u1*64)
    int ci[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float u = p[a] / dc.sdiv + 0.5f;
        if (u >= 1.f) u = dc.uclamp;
        if (u < 0.f) u = 0.f;
        ci[a] = (int)(u * (float)RES);
    }
    const int mycell[3] = {ci[0] + RES * ci[2], ci[0] + RES * ci[1], ci[1] + RES * ci[2]};   // xz, xy, yz

    // ---- the points of a cell as a ring in ascending point order (per plane) ----------------------------------
    // Both pool_local and scatter_mean only ever combine the points that share a cell (~600 points over ~1700
    // occupied cells of 4096), so the cloud's cell structure is resolved ONCE: an exchange on the cell's head
    // slot chains the points of a cell in arrival order, the oldest closes the chain into a ring, and every point
    // then picks its ascending successor (smallest larger index, else the ring's smallest).  The result does not
    // depend on the arrival order.  (Round 1 re-scanned the cloud's 600 cell ids in each of the four pooling
    // stages and again for the mean: 13 of the kernel's 16 ms.)
    {
        int* head = reinterpret_cast<int*>(net);                                           // [RES*RES] (aliases net)
        unsigned short* chain = reinterpret_cast<unsigned short*>(head + RES * RES);       // [3][ENC_THREADS]
#pragma unroll
        for (int P = 0; P < 3; ++P) {
            for (int k = i; k < RES * RES; k += ENC_THREADS) head[k] = -1;
            __syncthreads();
            int prev = -1;
            if (live) prev = atomicExch(&head[mycell[P]], i);
            __syncthreads();
            chain[P * ENC_THREADS + i] = (unsigned short)(!live ? i : prev >= 0 ? prev : head[mycell[P]]);
            __syncthreads();
        }
#pragma unroll
        for (int P = 0; P < 3; ++P) {
            const unsigned short* ch = chain + P * ENC_THREADS;
            int mn = i, up = 0x7fffffff;
            for (int j = ch[i]; j != i; j = ch[j]) {
                mn = min(mn, j);
                if (j > i) up = min(up, j);
            }
            ring[P * ENC_THREADS + i] = (unsigned short)(up != 0x7fffffff ? up : mn);
        }
        __syncthreads();
    }

    float x[64], cur[32];
    linear<64, 3, false, true>(w + eo.pos_w, w + eo.pos_b, p, x);          // fc_pos
    resnet_block(w, eo, 0, x, cur);
#pragma unroll
    for (int o = 0; o < 32; ++o) net[i * NET_STRIDE + o] = cur[o];
    __syncthreads();

    for (int blk = 1; blk < 5; ++blk) {
        // pool_local (pointnet.py:104-122): sum over planes of (max over the points of my cell), self included
        float pooled[32];
#pragma unroll
        for (int o = 0; o < 32; ++o) pooled[o] = 0.f;
#pragma unroll 1
        for (int P = 0; P < 3; ++P) {
            float mx[32];
#pragma unroll
            for (int o = 0; o < 32; ++o) mx[o] = cur[o];
            const unsigned short* rp = ring + P * ENC_THREADS;
            for (int j = rp[i]; j != i; j = rp[j]) {
#pragma unroll
                for (int o = 0; o < 32; ++o) mx[o] = fmaxf(mx[o], net[j * NET_STRIDE + o]);
            }
#pragma unroll
            for (int o = 0; o < 32; ++o) pooled[o] += mx[o];
        }
#pragma unroll
        for (int o = 0; o < 32; ++o) { x[o] = cur[o]; x[32 + o] = pooled[o]; }    // torch.cat([net, pooled], dim=2)
        resnet_block(w, eo, blk, x, cur);
        __syncthreads();                     // every thread has finished reading the old features
#pragma unroll
        for (int o = 0; o < 32; ++o) net[i * NET_STRIDE + o] = cur[o];
        __syncthreads();
    }

    float c[32];
    linear<32, 32, false, true>(w + eo.fcc_w, w + eo.fcc_b, cur, c);        // fc_c
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 32; ++o) net[i * NET_STRIDE + o] = c[o];
    if (c_out != nullptr && live) {
        float* co = c_out + ((size_t)b * Tmax + i) * 32;
#pragma unroll
        for (int o = 0; o < 32; ++o) co[o] = c[o];
    }
    __syncthreads();

    // scatter_mean into the zero-filled channel-last planes: the first point of a cell walks its ring, which adds the
    // cell's features in ascending point order (the order of an index-ordered scatter_add); later points of the cell
    // meet a smaller index on their first wrap and drop out
    if (live) {
#pragma unroll
        for (int P = 0; P < 3; ++P) {
            const unsigned short* rp = ring + P * ENC_THREADS;
            float sum[32];
#pragma unroll
            for (int o = 0; o < 32; ++o) sum[o] = c[o];
            float cnt = 1.f;
            bool first = true;
            for (int j = rp[i]; j != i; j = rp[j]) {
                if (j < i) { first = false; break; }
                cnt += 1.f;
#pragma unroll
                for (int o = 0; o < 32; ++o) sum[o] += net[j * NET_STRIDE + o];
            }
            if (first) {
                float* dst = planes + (((size_t)b * 3 + P) * RES * RES + mycell[P]) * CH;
#pragma unroll
                for (int o = 0; o < 32; o += 4)
                    *reinterpret_cast<f32x4*>(dst + o) =
                        f32x4{sum[o] / cnt, sum[o + 1] / cnt, sum[o + 2] / cnt, sum[o + 3] / cnt};
            }
        }
    }
}

hipError_t configure_encoder_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(encode_points_kernel<640>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds(640));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(encode_points_kernel<1024>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds(1024));
}

hipError_t launch_encode_points(const float* w, const EncPointOffsets& eo, const float* sel, const int* t_per_cloud,
                                int B, int Tmax, float* planes, float* c_out, DecConst dc, hipStream_t s) {
    // 640 threads (10 waves, 168 VGPRs) cover the shipped pointcloud_n = 600; larger subsets use 1024 threads
    if (Tmax <= 640)
        hipLaunchKernelGGL(encode_points_kernel<640>, dim3(B), dim3(640), enc_lds(640), s, w, eo, sel, t_per_cloud,
                           Tmax, planes, c_out, dc);
    else
        hipLaunchKernelGGL(encode_points_kernel<1024>, dim3(B), dim3(1024), enc_lds(1024), s, w, eo, sel,
                           t_per_cloud, Tmax, planes, c_out, dc);
    return hipGetLastError();
}

}  // namespace ifd
