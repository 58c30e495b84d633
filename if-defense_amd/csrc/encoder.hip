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
#include <cstdlib>
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


// ---------------------------------------------------------------------------------------------
// Round 4: the same encoder with its linear layers on the matrix pipe (v_mfma_f32_16x16x4_f32).
//
// The thread-per-point kernel above issues 26.8 k scalar-weight FMAs per point - 268 k vector instructions per cloud at a
// quarter of the vector rate (every FMA waits on a scalar weight load; 10 waves per CU cannot hide them): 3.4 ms per 2468
// clouds.  Here a wave takes the cloud's points 16 at a time: lane (n, q) holds, for point n of the tile, the 8 channels
// 16 mt + 4 q + r of a 32-channel vector (the decoder tile's layout, optimize.hip): the accumulator layout of one layer IS the
// B-operand layout of the next, so features never leave registers between the layers of a block, and the A operands (weights)
// are 16-byte loads straight from an aligned copy of the point-net's weights (api.cpp pack_encoder: L1 / L2 resident, the same
// 109 KB for every cloud) - k-slot (s, kq) <-> input channel 16 (s >> 2) + 4 kq + (s & 3) on both operands.
// pool_local and scatter_mean walk the same per-cell rings as before, every lane for its own 8 channels (the four lanes of a
// point walk in step).  The cloud's planes are zero-filled by the kernel itself at its start (the stores drain under the
// compute; ifd_encode_points issued a 3.9 GB hipMemsetAsync per pass in front of the kernel).
// Summation order: four input channels per MFMA, k-steps in ascending s - not the sequential order of `linear` above:
// results agree with it to float32 rounding (the fixtures hold both to 1e-5), independent of the batch as before.
// ---------------------------------------------------------------------------------------------
constexpr int ENCI_POS = 0;                          // [64][4] = {Wpos[o][0..2], bpos[o]}
constexpr int ENCI_BLK0 = 256;                       // per block: W0 [32][64], b0 [32], W1 [32][32], b1 [32], Ws [32][64]
constexpr int ENCI_W0 = 0, ENCI_B0 = 2048, ENCI_W1 = 2080, ENCI_B1 = 3104, ENCI_WS = 3136, ENCI_BLK = 5184;
constexpr int ENCI_WC = ENCI_BLK0 + 5 * ENCI_BLK;    // [32][32]
constexpr int ENCI_BC = ENCI_WC + 1024;              // [32]
constexpr int ENCI_FLOATS = ENCI_BC + 32;            // 27,232
constexpr int ENC_NSTR = 36;                         // LDS row of the [points][32] feature matrix (floats): 16-byte pieces, conflict-free writes

struct Acc8 { f32x4 t[2]; };                         // 8 channels of a point: t[mt][r] <-> channel 16 mt + 4 q + r

// acc[mt] += W[16 mt + m][cidx(s, kq)] * in[s] over the NS k-steps; W row-major [32][4 NS]; RELU_IN: relu on the B operand
template <int NS, bool RELU_IN>
__device__ __forceinline__ void enc_dense(const float* __restrict__ W, int m, int kq, const float (&in)[NS], Acc8& acc) {
    constexpr int NIN = 4 * NS;
    f32x4 a[2][NS / 4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < NS / 4; ++g) a[mt][g] = *reinterpret_cast<const f32x4*>(W + (16 * mt + m) * NIN + 16 * g + 4 * kq);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float b = RELU_IN ? fmaxf(in[s], 0.f) : in[s];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc.t[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][s >> 2][s & 3], b, acc.t[mt], 0, 0, 0);
    }
}
__device__ __forceinline__ Acc8 enc_bias(const float* __restrict__ b, int q) {
    Acc8 r;
    r.t[0] = *reinterpret_cast<const f32x4*>(b + 4 * q);
    r.t[1] = *reinterpret_cast<const f32x4*>(b + 16 + 4 * q);
    return r;
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void encode_points_mfma_kernel(const float* __restrict__ img, const float* __restrict__ sel,
                                                                      const int* __restrict__ t_per_cloud, int Tmax,
                                                                      float* __restrict__ planes, float* __restrict__ c_out,
                                                                      DecConst dc) {
    constexpr int NT = NW * 64;                                            // threads = points the LDS is sized for
    constexpr int TPW = (NT / 16 + NW - 1) / NW;                           // 16-point tiles per wave: 4
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* net = smem;                                                     // [NT][ENC_NSTR]
    unsigned short* ring = reinterpret_cast<unsigned short*>(net + NT * ENC_NSTR);   // [3][NT] next point of my cell
    unsigned short* cellb = ring + 3 * NT;                                 // [3][NT] cell of the point per plane

    const int b = blockIdx.x, i = threadIdx.x, lane = i & 63, wave = i >> 6;
    const int T = t_per_cloud ? min(t_per_cloud[b], Tmax) : Tmax;
    // ---- zero-fill of the cloud's three planes (empty cells stay 0); complete before the scatter at the end -------------
    {
        f32x4* pz = reinterpret_cast<f32x4*>(planes + (size_t)b * CLOUD_PLANE_FLOATS);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int k = i; k < CLOUD_PLANE_FLOATS / 4; k += NT) __builtin_nontemporal_store(z, pz + k);
    }
    // ---- cells and per-cell rings, one thread per point (as in the kernel above) -------------------------------------------
    {
        const bool live = i < T;
        const float* ps = sel + ((size_t)b * Tmax + (live ? i : 0)) * 3;
        const float p[3] = {ps[0], ps[1], ps[2]};
        int ci[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float u = p[a] / dc.sdiv + 0.5f;
            if (u >= 1.f) u = dc.uclamp;
            if (u < 0.f) u = 0.f;
            ci[a] = (int)(u * (float)RES);
        }
        const int mycell[3] = {ci[0] + RES * ci[2], ci[0] + RES * ci[1], ci[1] + RES * ci[2]};   // xz, xy, yz
        int* head = reinterpret_cast<int*>(net);                                           // [RES*RES] (aliases net)
        unsigned short* chain = reinterpret_cast<unsigned short*>(head + RES * RES);       // [3][NT]
#pragma unroll
        for (int P = 0; P < 3; ++P) {
            cellb[P * NT + i] = (unsigned short)mycell[P];
            for (int k = i; k < RES * RES; k += NT) head[k] = -1;
            __syncthreads();
            int prev = -1;
            if (live) prev = atomicExch(&head[mycell[P]], i);
            __syncthreads();
            chain[P * NT + i] = (unsigned short)(!live ? i : prev >= 0 ? prev : head[mycell[P]]);
            __syncthreads();
        }
#pragma unroll
        for (int P = 0; P < 3; ++P) {
            const unsigned short* ch = chain + P * NT;
            int mn = i, up = 0x7fffffff;
            for (int j = ch[i]; j != i; j = ch[j]) {
                mn = min(mn, j);
                if (j > i) up = min(up, j);
            }
            ring[P * NT + i] = (unsigned short)(up != 0x7fffffff ? up : mn);
        }
        __syncthreads();
    }

    // ---- the point-net on 16-point tiles: tile k of this wave covers the points 16 (wave + k NW) ... + 15 ------------------
    const int n = lane & 15, q = lane >> 4;
    const int ntiles = (T + 15) >> 4;
    Acc8 cur[TPW];
    auto tile_point = [&](int k) { return 16 * (wave + k * NW) + n; };
    // fc_pos (K = 4: x, y, z and a 1 that carries the bias) and block 0, whose input is fc_pos' 64 channels
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        if (wave + k * NW >= ntiles) { cur[k].t[0] = cur[k].t[1] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }      // (wave-uniform)
        const int pt = min(tile_point(k), T - 1);
        const float xq = q < 3 ? sel[((size_t)b * Tmax + pt) * 3 + q] : 1.f;
        float x[16];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x4f32(img[ENCI_POS + (16 * mt + n) * 4 + q], xq, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[4 * mt + r] = o[r];
        }
        const float* B0 = img + ENCI_BLK0;
        Acc8 h = enc_bias(B0 + ENCI_B0, q);
        enc_dense<16, true>(B0 + ENCI_W0, n, q, x, h);
        Acc8 xs;
        xs.t[0] = xs.t[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        enc_dense<16, false>(B0 + ENCI_WS, n, q, x, xs);
        float hv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { hv[r] = h.t[0][r]; hv[4 + r] = h.t[1][r]; }
        Acc8 dx = enc_bias(B0 + ENCI_B1, q);
        enc_dense<8, true>(B0 + ENCI_W1, n, q, hv, dx);
        cur[k].t[0] = xs.t[0] + dx.t[0];
        cur[k].t[1] = xs.t[1] + dx.t[1];
    }
    auto put_net = [&](const Acc8 (&v)[TPW]) {
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            if (wave + k * NW >= ntiles) continue;
            float* d = net + tile_point(k) * ENC_NSTR + 4 * q;
            *reinterpret_cast<f32x4*>(d) = v[k].t[0];
            *reinterpret_cast<f32x4*>(d + 16) = v[k].t[1];
        }
    };
    put_net(cur);
    __syncthreads();

    for (int blk = 1; blk < 5; ++blk) {
        const float* Bk = img + ENCI_BLK0 + blk * ENCI_BLK;
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            if (wave + k * NW >= ntiles) continue;
            const int pi = tile_point(k);
            // pool_local (pointnet.py:104-122): sum over planes of (max over the points of my cell), self included
            Acc8 pooled;
            pooled.t[0] = pooled.t[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int P = 0; P < 3; ++P) {
                Acc8 mx = cur[k];
                const unsigned short* rp = ring + P * NT;
                for (int j = rp[pi]; j != pi; j = rp[j]) {
                    const float* s = net + j * ENC_NSTR + 4 * q;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(s), a1 = *reinterpret_cast<const f32x4*>(s + 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { mx.t[0][r] = fmaxf(mx.t[0][r], a0[r]); mx.t[1][r] = fmaxf(mx.t[1][r], a1[r]); }
                }
                pooled.t[0] += mx.t[0];
                pooled.t[1] += mx.t[1];
            }
            float x[16];                                                   // torch.cat([net, pooled], dim=2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x[r] = cur[k].t[0][r]; x[4 + r] = cur[k].t[1][r];
                x[8 + r] = pooled.t[0][r]; x[12 + r] = pooled.t[1][r];
            }
            Acc8 h = enc_bias(Bk + ENCI_B0, q);
            enc_dense<16, true>(Bk + ENCI_W0, n, q, x, h);
            Acc8 xs;
            xs.t[0] = xs.t[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            enc_dense<16, false>(Bk + ENCI_WS, n, q, x, xs);
            float hv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { hv[r] = h.t[0][r]; hv[4 + r] = h.t[1][r]; }
            Acc8 dx = enc_bias(Bk + ENCI_B1, q);
            enc_dense<8, true>(Bk + ENCI_W1, n, q, hv, dx);
            cur[k].t[0] = xs.t[0] + dx.t[0];
            cur[k].t[1] = xs.t[1] + dx.t[1];
        }
        __syncthreads();                     // every wave has finished reading the old features
        put_net(cur);
        __syncthreads();
    }

    // fc_c, then the features once more through LDS for the scatter-mean
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        if (wave + k * NW >= ntiles) continue;
        float cv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { cv[r] = cur[k].t[0][r]; cv[4 + r] = cur[k].t[1][r]; }
        Acc8 c = enc_bias(img + ENCI_BC, q);
        enc_dense<8, false>(img + ENCI_WC, n, q, cv, c);
        cur[k] = c;
        const int pi = tile_point(k);
        if (c_out != nullptr && pi < T) {
            float* co = c_out + ((size_t)b * Tmax + pi) * 32 + 4 * q;
            *reinterpret_cast<f32x4*>(co) = c.t[0];
            *reinterpret_cast<f32x4*>(co + 16) = c.t[1];
        }
    }
    __syncthreads();
    put_net(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's zero-fill stores have been acknowledged
    __syncthreads();

    // scatter_mean into the zero-filled channel-last planes: the first point of a cell walks its ring, which adds the
    // cell's features in ascending point order (the order of an index-ordered scatter_add); later points of the cell
    // meet a smaller index on their first wrap and drop out
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        if (wave + k * NW >= ntiles) continue;
        const int pi = tile_point(k);
        if (pi >= T) continue;
#pragma unroll 1
        for (int P = 0; P < 3; ++P) {
            const unsigned short* rp = ring + P * NT;
            Acc8 sum = cur[k];
            float cnt = 1.f;
            bool first = true;
            for (int j = rp[pi]; j != pi; j = rp[j]) {
                if (j < pi) { first = false; break; }
                cnt += 1.f;
                const float* s = net + j * ENC_NSTR + 4 * q;
                sum.t[0] += *reinterpret_cast<const f32x4*>(s);
                sum.t[1] += *reinterpret_cast<const f32x4*>(s + 16);
            }
            if (first) {
                float* dst = planes + (((size_t)b * 3 + P) * RES * RES + cellb[P * NT + pi]) * CH + 4 * q;
                *reinterpret_cast<f32x4*>(dst) = f32x4{sum.t[0][0] / cnt, sum.t[0][1] / cnt, sum.t[0][2] / cnt, sum.t[0][3] / cnt};
                *reinterpret_cast<f32x4*>(dst + 16) = f32x4{sum.t[1][0] / cnt, sum.t[1][1] / cnt, sum.t[1][2] / cnt, sum.t[1][3] / cnt};
            }
        }
    }
}

constexpr size_t enc_mfma_lds(int threads) { return (size_t)threads * ENC_NSTR * 4 + 6 * (size_t)threads * 2; }   // 99,840 B @640

hipError_t configure_encoder_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(encode_points_kernel<640>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds(640));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(encode_points_kernel<1024>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds(1024));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(encode_points_mfma_kernel<10>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_mfma_lds(640));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(encode_points_mfma_kernel<16>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_mfma_lds(1024));
}

int enc_image_floats() { return ENCI_FLOATS; }
// aligned copy of the point-net's weights in the layout of encode_points_mfma_kernel, from the canonical weight vector
void build_enc_image(const float* w, const EncPointOffsets& eo, float* img) {
    for (int o = 0; o < 64; ++o) {
        for (int a = 0; a < 3; ++a) img[ENCI_POS + 4 * o + a] = w[eo.pos_w + 3 * o + a];
        img[ENCI_POS + 4 * o + 3] = w[eo.pos_b + o];
    }
    for (int i = 0; i < 5; ++i) {
        float* d = img + ENCI_BLK0 + i * ENCI_BLK;
        for (int k = 0; k < 2048; ++k) { d[ENCI_W0 + k] = w[eo.fc0_w[i] + k]; d[ENCI_WS + k] = w[eo.sc_w[i] + k]; }
        for (int k = 0; k < 1024; ++k) d[ENCI_W1 + k] = w[eo.fc1_w[i] + k];
        for (int k = 0; k < 32; ++k) { d[ENCI_B0 + k] = w[eo.fc0_b[i] + k]; d[ENCI_B1 + k] = w[eo.fc1_b[i] + k]; }
    }
    for (int k = 0; k < 1024; ++k) img[ENCI_WC + k] = w[eo.fcc_w + k];
    for (int k = 0; k < 32; ++k) img[ENCI_BC + k] = w[eo.fcc_b + k];
}

// enc_img: build_enc_image's copy on the device.  -DIFD_ENC_VALU / env IFD_ENC_VALU=1: the thread-per-point kernel (validation, A/B);
// it expects zero-filled planes, the MFMA kernel fills them itself.
hipError_t launch_encode_points(const float* w, const EncPointOffsets& eo, const float* enc_img, const float* sel,
                                const int* t_per_cloud, int B, int Tmax, float* planes, float* c_out, DecConst dc, hipStream_t s) {
#ifdef IFD_ENC_VALU
    const bool valu = true;
#else
    static const bool valu = [] { const char* e = getenv("IFD_ENC_VALU"); return e != nullptr && e[0] == '1'; }();
#endif
    if (valu) {
        hipError_t e = hipMemsetAsync(planes, 0, (size_t)B * CLOUD_PLANE_FLOATS * sizeof(float), s);
        if (e != hipSuccess) return e;
        // 640 threads (10 waves, 168 VGPRs) cover the shipped pointcloud_n = 600; larger subsets use 1024 threads
        if (Tmax <= 640)
            hipLaunchKernelGGL(encode_points_kernel<640>, dim3(B), dim3(640), enc_lds(640), s, w, eo, sel, t_per_cloud,
                               Tmax, planes, c_out, dc);
        else
            hipLaunchKernelGGL(encode_points_kernel<1024>, dim3(B), dim3(1024), enc_lds(1024), s, w, eo, sel,
                               t_per_cloud, Tmax, planes, c_out, dc);
        return hipGetLastError();
    }
    if (Tmax <= 640)
        hipLaunchKernelGGL(encode_points_mfma_kernel<10>, dim3(B), dim3(640), enc_mfma_lds(640), s, enc_img, sel, t_per_cloud, Tmax,
                           planes, c_out, dc);
    else
        hipLaunchKernelGGL(encode_points_mfma_kernel<16>, dim3(B), dim3(1024), enc_mfma_lds(1024), s, enc_img, sel, t_per_cloud, Tmax,
                           planes, c_out, dc);
    return hipGetLastError();
}

}  // namespace ifd
