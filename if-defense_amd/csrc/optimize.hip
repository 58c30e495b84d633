// ConvONet-Opt hot loop for MI355X (gfx950): one workgroup owns one cloud for all Adam steps.
//
//   per step (reference: ConvONet/opt_defense.py:210-228)
//     phase A  decoder forward + input-gradient on 32-point tiles, one wave per tile:
//              bilinear gather of the 3 channel-last planes (decoder.py:50-57), the 5-block
//              ResNet MLP on v_mfma_f32_32x32x2_f32 (decoder.py:83-93, layers.py:39-48), BCE-to-
//              threshold derivative (opt_defense.py:213-216), transposed MLP, re-gather for dc/du.
//     phase B  brute-force 5-NN over the cloud's xyz held in LDS + repulsion loss gradient
//              (defense/pn_utils.py:64-83, defense/repulsion_loss.py:43-54); neighbour terms are
//              scattered with 64-bit fixed-point LDS atomics => order independent, bit reproducible.
//     phase C  fused Adam update (torch.optim.Adam single-tensor form), moments in registers.
//   Nothing but the plane taps is read from global memory inside the loop.
//
// MFMA operand mapping (32x32x2, f32): M = output channel, N = point of the tile, K = input
// channel.  The accumulator layout (lane l: point l&31, register r: channel (r&3)+8(r>>2)+4(l>>5))
// is exactly the B-operand layout of the next layer when MFMA step s consumes register s, so
// activations never leave registers and never need a transpose; weights stream from LDS as the
// A operand (one ds_read_b32 per MFMA, conflict-free thanks to the 33-float row stride).
#include "ifd_device.h"
#include "ifd_internal.h"

namespace ifd {

__device__ __forceinline__ int chan(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// out[o][n] (+)= sum_k A[o][k] * in[k][n] with A = W (forward) or W^T (backward).
template <bool TRANSPOSED>
__device__ __forceinline__ f32x16 dense32(const float* __restrict__ wl, int n, int hi, const f32x16& in,
                                          f32x16 acc) {
    const float* base = TRANSPOSED ? (wl + (4 * hi) * W_STRIDE + n) : (wl + n * W_STRIDE + 4 * hi);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int o = (s & 3) + 8 * (s >> 2);
        const float a = base[TRANSPOSED ? o * W_STRIDE : o];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, in[s], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);       // do not hoist the next layer's weight loads over this one
    return acc;
}

__device__ __forceinline__ f32x16 load_bias(const float* __restrict__ W, int layer, int hi) {
    f32x16 b;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(W + DEC_OFF_BIAS + layer * 32 + 8 * g + 4 * hi);
        b[4 * g + 0] = t.x; b[4 * g + 1] = t.y; b[4 * g + 2] = t.z; b[4 * g + 3] = t.w;
    }
    return b;
}

__device__ __forceinline__ uint32_t mask_pos(const f32x16& v) {
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m |= (v[r] > 0.f ? 1u : 0u) << r;
    return m;
}

__device__ __forceinline__ f32x16 relu16(const f32x16& v) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = fmaxf(v[r], 0.f);
    return o;
}

// common.py:250-257 then grid_sample's unnormalise (align_corners) + border clip.
__device__ __forceinline__ void pixel_coord(float xa, const DecConst& dc, float& pix, float& live) {
    float u = xa / dc.sdiv + 0.5f;
    live = 1.f;
    if (u >= 1.f) { u = dc.uclamp; live = 0.f; }
    if (u < 0.f) { u = 0.f; live = 0.f; }
    const float v = 2.0f * u - 1.0f;
    pix = ((v + 1.f) / 2.f) * (float)(RES - 1);
    pix = fminf(fmaxf(pix, 0.f), (float)(RES - 1));
}

enum { MODE_OPT = 0, MODE_SUM = 1 };

// One 32-point tile on one wave.  Every lane pair (l, l^32) works on point l&31 and holds 16 of its
// 32 channels.  Returns logit, the BCE term and d(loss)/dx (valid on every lane after the pair reduce).
template <int MODE, bool WANT_GRAD>
__device__ __forceinline__ void decoder_tile(const float* __restrict__ W, const float* __restrict__ planes,
                                             float x0, float x1, float x2, int lane, const DecConst dc,
                                             float thr, float inv_lb, float& logit_out, float& bce_out,
                                             float (&dx)[3]) {
    const int n = lane & 31, hi = lane >> 5;
    float pix[3], live[3];
    pixel_coord(x0, dc, pix[0], live[0]);
    pixel_coord(x1, dc, pix[1], live[1]);
    pixel_coord(x2, dc, pix[2], live[2]);
    int cell[3];
    float w1[3], w0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int ci = min((int)floorf(pix[a]), RES - 2);
        cell[a] = ci;
        w1[a] = pix[a] - (float)ci;            // weight of the east / south tap
        w0[a] = ((float)ci + 1.f) - pix[a];    // weight of the west / north tap
    }
    constexpr int AX0[3] = {0, 0, 1}, AX1[3] = {2, 1, 2};   // xz, xy, yz (common.py:243-248)

    // ---- forward: c = sum over planes of the bilinear sample -------------------------------
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
    for (int P = 0; P < 3; ++P) {
        const int a0 = AX0[P], a1 = AX1[P];
        const float* q = planes + ((P * RES + cell[a1]) * RES + cell[a0]) * CH + 4 * hi;
        const float wnw = w0[a0] * w0[a1], wne = w1[a0] * w0[a1], wsw = w0[a0] * w1[a1], wse = w1[a0] * w1[a1];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 tnw = *reinterpret_cast<const f32x4*>(q + 8 * g);
            const f32x4 tne = *reinterpret_cast<const f32x4*>(q + CH + 8 * g);
            const f32x4 tsw = *reinterpret_cast<const f32x4*>(q + RES * CH + 8 * g);
            const f32x4 tse = *reinterpret_cast<const f32x4*>(q + RES * CH + CH + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = tnw[j] * wnw;
                s = fmaf(tne[j], wne, s);
                s = fmaf(tsw[j], wsw, s);
                s = fmaf(tse[j], wse, s);
                c[4 * g + j] += s;
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep at most one plane's 16 tap loads in flight
    }

    // ---- forward MLP ------------------------------------------------------------------------
    f32x16 net;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WP + (8 * g + 4 * hi + j) * 4);
            net[4 * g + j] = fmaf(wp.z, x2, fmaf(wp.y, x1, fmaf(wp.x, x0, wp.w)));
        }
    uint32_t mask_a[NBLK], mask_h[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const float* Wl = W + DEC_OFF_W + 3 * i * W_LAYER;
        f32x16 a = net + load_bias(W, 3 * i, hi);
        a = dense32<false>(Wl, n, hi, c, a);                          // a_i = n_i + fc_c[i](c)
        mask_a[i] = mask_pos(a);
        f32x16 h = load_bias(W, 3 * i + 1, hi);
        h = dense32<false>(Wl + W_LAYER, n, hi, relu16(a), h);        // fc_0(relu(a))
        mask_h[i] = mask_pos(h);
        f32x16 o = a + load_bias(W, 3 * i + 2, hi);
        net = dense32<false>(Wl + 2 * W_LAYER, n, hi, relu16(h), o);  // a + fc_1(relu(h))
    }
    const uint32_t mask_n = mask_pos(net);
    float wout[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WOUT + 8 * g + 4 * hi);
        wout[4 * g + 0] = t.x; wout[4 * g + 1] = t.y; wout[4 * g + 2] = t.z; wout[4 * g + 3] = t.w;
    }
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part = fmaf(wout[r], fmaxf(net[r], 0.f), part);
    const float logit = part + __shfl_xor(part, 32) + W[DEC_OFF_BOUT];
    logit_out = logit;
    bce_out = 0.f;
    if (!WANT_GRAD) return;

    // ---- backward (parameters frozen: only the path to the input) --------------------------------
    float dl;
    if (MODE == MODE_OPT) {
        const float e = expf(-fabsf(logit));
        bce_out = fmaxf(logit, 0.f) - thr * logit + log1pf(e);
        const float sig = logit >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        dl = (sig - thr) * inv_lb;
    } else {
        dl = 1.f;
    }
    f32x16 dn, zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        zero[r] = 0.f;
        dn[r] = ((mask_n >> r) & 1u) ? dl * wout[r] : 0.f;
    }
    f32x16 dcc = zero;
#pragma unroll
    for (int i = NBLK - 1; i >= 0; --i) {
        const float* Wl = W + DEC_OFF_W + 3 * i * W_LAYER;
        f32x16 dh = dense32<true>(Wl + 2 * W_LAYER, n, hi, dn, zero);
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = ((mask_h[i] >> r) & 1u) ? dh[r] : 0.f;
        const f32x16 t = dense32<true>(Wl + W_LAYER, n, hi, dh, zero);
#pragma unroll
        for (int r = 0; r < 16; ++r) dn[r] += ((mask_a[i] >> r) & 1u) ? t[r] : 0.f;   // delta a_i
        dcc = dense32<true>(Wl, n, hi, dn, dcc);                                       // += Wc^T delta a_i
    }
    float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WP + (8 * gi + 4 * hi + j) * 4);
            const float d = dn[4 * gi + j];
            g[0] = fmaf(wp.x, d, g[0]); g[1] = fmaf(wp.y, d, g[1]); g[2] = fmaf(wp.z, d, g[2]);
        }
    // d c / d u through the bilinear taps (grid_sampler_2d backward w.r.t. the grid)
#pragma unroll
    for (int P = 0; P < 3; ++P) {
        const int a0 = AX0[P], a1 = AX1[P];
        int off = ((P * RES + cell[a1]) * RES + cell[a0]) * CH + 4 * hi;
        // opaque to the optimiser: otherwise these loads are CSE'd with the forward gather and all
        // 192 tap registers stay live (and spill) across the whole MLP.  The re-read is L1/L2 traffic.
        asm volatile("" : "+v"(off));
        const float* q = planes + off;
        float dnw = 0.f, dne = 0.f, dsw = 0.f, dse = 0.f;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const f32x4 tnw = *reinterpret_cast<const f32x4*>(q + 8 * gi);
            const f32x4 tne = *reinterpret_cast<const f32x4*>(q + CH + 8 * gi);
            const f32x4 tsw = *reinterpret_cast<const f32x4*>(q + RES * CH + 8 * gi);
            const f32x4 tse = *reinterpret_cast<const f32x4*>(q + RES * CH + CH + 8 * gi);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = dcc[4 * gi + j];
                dnw = fmaf(tnw[j], d, dnw); dne = fmaf(tne[j], d, dne);
                dsw = fmaf(tsw[j], d, dsw); dse = fmaf(tse[j], d, dse);
            }
        }
        const float gix = (dne - dnw) * w0[a1] + (dse - dsw) * w1[a1];
        const float giy = (dsw - dnw) * w0[a0] + (dse - dne) * w1[a0];
        const float sc = (0.5f * (float)(RES - 1)) * 2.f;
        g[a0] += live[a0] * ((gix * sc) / dc.sdiv);
        g[a1] += live[a1] * ((giy * sc) / dc.sdiv);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) dx[a] = g[a] + __shfl_xor(g[a], 32);
}

// ---------------------------------------------------------------------------------------------
// 5-NN + repulsion
// ---------------------------------------------------------------------------------------------
struct Top5 {
    float d0, d1, d2, d3, d4;
    int i0, i1, i2, i3, i4;
};

__device__ __forceinline__ void top5_init(Top5& t) {
    t.d0 = t.d1 = t.d2 = t.d3 = t.d4 = INFINITY;
    t.i0 = t.i1 = t.i2 = t.i3 = t.i4 = 0;
}

__device__ __forceinline__ void top5_insert(Top5& t, float d, int j) {
    if (d < t.d4) {
        const bool c3 = d < t.d3, c2 = d < t.d2, c1 = d < t.d1, c0 = d < t.d0;
        t.d4 = c3 ? t.d3 : d;                 t.i4 = c3 ? t.i3 : j;
        t.d3 = c3 ? (c2 ? t.d2 : d) : t.d3;   t.i3 = c3 ? (c2 ? t.i2 : j) : t.i3;
        t.d2 = c2 ? (c1 ? t.d1 : d) : t.d2;   t.i2 = c2 ? (c1 ? t.i1 : j) : t.i2;
        t.d1 = c1 ? (c0 ? t.d0 : d) : t.d1;   t.i1 = c1 ? (c0 ? t.i0 : j) : t.i1;
        t.d0 = c0 ? d : t.d0;                 t.i0 = c0 ? j : t.i0;
    }
}

// Brute-force scan of all K points (broadcast LDS reads) for the two points owned by this thread.
__device__ __forceinline__ void knn_scan2(const f32x4* __restrict__ X, int K, int pa, int pb, Top5& ta,
                                          Top5& tb) {
    const f32x4 xa = X[min(pa, K - 1)], xb = X[min(pb, K - 1)];
    top5_init(ta);
    top5_init(tb);
#pragma unroll 4
    for (int j = 0; j < K; ++j) {
        const f32x4 xj = X[j];
        float ax = xj.x - xa.x, ay = xj.y - xa.y, az = xj.z - xa.z;
        float bx = xj.x - xb.x, by = xj.y - xb.y, bz = xj.z - xb.z;
        float da = fmaf(az, az, fmaf(ay, ay, ax * ax));
        float db = fmaf(bz, bz, fmaf(by, by, bx * bx));
        da = (j == pa) ? INFINITY : da;
        db = (j == pb) ? INFINITY : db;
        top5_insert(ta, da, j);
        top5_insert(tb, db, j);
    }
}

struct RepConst {
    float radius, h, eps;
};

// Loss and gradient terms of one centre point (repulsion_loss.py:43-53).  The centre part is
// returned in gc (un-scaled), neighbour parts go to the fixed-point LDS accumulator.
__device__ __forceinline__ void rep_point(const f32x4* __restrict__ X, long long* __restrict__ F, int i,
                                          const Top5& t, const RepConst rc, float& loss, float (&gc)[3],
                                          bool want_grad) {
    const f32x4 xi = X[i];
    const int idx[5] = {t.i0, t.i1, t.i2, t.i3, t.i4};
    loss = 0.f;
    gc[0] = gc[1] = gc[2] = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int j = idx[k];
        const f32x4 xj = X[j];
        const float ex = xj.x - xi.x, ey = xj.y - xi.y, ez = xj.z - xi.z;
        const float d2raw = ex * ex + ey * ey + ez * ez;
        const float d2 = fmaxf(d2raw, rc.eps);
        const float d = sqrtf(d2);
        const float q = d / rc.h;
        const float w = expf(-(q * q));
        loss += (rc.radius - d) * w;
        if (want_grad) {
            // dL/dd = -w - (r-d) w 2 d / h^2 ; chain through sqrt and the clamp (zero below eps)
            const float dd = -w - (rc.radius - d) * w * (2.f * q / rc.h);
            const float coef = d2raw > rc.eps ? dd / d : 0.f;
            const float gx = coef * ex, gy = coef * ey, gz = coef * ez;
            gc[0] -= gx; gc[1] -= gy; gc[2] -= gz;
            atomicAdd(reinterpret_cast<unsigned long long*>(F + 3 * j + 0),
                      (unsigned long long)__float2ll_rn(gx * FIX_SCALE));
            atomicAdd(reinterpret_cast<unsigned long long*>(F + 3 * j + 1),
                      (unsigned long long)__float2ll_rn(gy * FIX_SCALE));
            atomicAdd(reinterpret_cast<unsigned long long*>(F + 3 * j + 2),
                      (unsigned long long)__float2ll_rn(gz * FIX_SCALE));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// block helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// deterministic block reductions over OPT_THREADS threads; `scratch` holds >= 8 floats
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += scratch[w];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float s = scratch[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s = fmaxf(s, scratch[w]);
    return s;
}

__device__ __forceinline__ void load_dec_image(float* __restrict__ W, const float* __restrict__ img) {
    for (int i = threadIdx.x * 4; i < DEC_FLOATS; i += blockDim.x * 4)
        *reinterpret_cast<f32x4*>(W + i) = *reinterpret_cast<const f32x4*>(img + i);
}

// normalize_batch_pc (opt_defense.py:76-83) on the cloud held in X; two points per thread.
__device__ __forceinline__ void normalize_in_lds(f32x4* __restrict__ X, int K, float* scratch) {
    const int pa = threadIdx.x, pb = threadIdx.x + OPT_THREADS;
    f32x4 a = pa < K ? X[pa] : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 b = pb < K ? X[pb] : f32x4{0.f, 0.f, 0.f, 0.f};
    const float cx = block_sum(a.x + b.x, scratch) / (float)K;
    const float cy = block_sum(a.y + b.y, scratch) / (float)K;
    const float cz = block_sum(a.z + b.z, scratch) / (float)K;
    a.x -= cx; a.y -= cy; a.z -= cz;
    b.x -= cx; b.y -= cy; b.z -= cz;
    const float da = pa < K ? sqrtf(a.x * a.x + a.y * a.y + a.z * a.z) : 0.f;
    const float db = pb < K ? sqrtf(b.x * b.x + b.y * b.y + b.z * b.z) : 0.f;
    const float md = block_max(fmaxf(da, db), scratch);
    if (pa < K) X[pa] = f32x4{a.x / md, a.y / md, a.z / md, 0.f};
    if (pb < K) X[pb] = f32x4{b.x / md, b.y / md, b.z / md, 0.f};
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// the persistent per-cloud optimiser
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OPT_THREADS, 2) void optimize_kernel(
    const float* __restrict__ dec_img, const float* __restrict__ planes, float* __restrict__ p,
    float* __restrict__ m_io, float* __restrict__ v_io, float* __restrict__ loss_out, int K, OptArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    f32x4* X = reinterpret_cast<f32x4*>(smem + DEC_FLOATS);
    f32x4* G = X + MAXK;
    long long* F = reinterpret_cast<long long*>(G + MAXK);
    float* scratch = reinterpret_cast<float*>(F + 3 * MAXK);   // 64 floats

    const int cloud = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pl = planes + (size_t)cloud * CLOUD_PLANE_FLOATS;
    float* pc = p + (size_t)cloud * K * 3;
    const int pa = tid, pb = tid + OPT_THREADS;
    const int ntiles = (K + 31) >> 5;

    load_dec_image(W, dec_img);
    float mm[6], vv[6];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pt = q ? pb : pa;
        if (pt < K) {
            X[pt] = f32x4{pc[3 * pt], pc[3 * pt + 1], pc[3 * pt + 2], 0.f};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const bool have = A.t0 > 0 && m_io != nullptr;
                mm[3 * q + a] = have ? m_io[((size_t)cloud * K + pt) * 3 + a] : 0.f;
                vv[3 * q + a] = have ? v_io[((size_t)cloud * K + pt) * 3 + a] : 0.f;
            }
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) mm[3 * q + a] = vv[3 * q + a] = 0.f;
        }
    }
    for (int i = tid; i < MAXK * 3; i += OPT_THREADS) F[i] = 0;
    __syncthreads();

    const DecConst dc = A.dc;
    const RepConst rc = {A.rep_radius, A.rep_h, A.rep_eps};
    const float inv_lb = 1.0f / (float)A.loss_batch;
    const float rep_scale = A.rep_weight / ((float)A.loss_batch * (float)K * 5.f);
    const bool use_rep = A.rep_weight > 0.f;
    double b1t = pow(0.9, (double)A.t0), b2t = pow(0.999, (double)A.t0);
    float rep_loss_a = 0.f, rep_loss_b = 0.f;

    for (int step = 0; step < A.steps; ++step) {
        const bool last = step == A.steps - 1;
        float gca[3] = {0.f, 0.f, 0.f}, gcb[3] = {0.f, 0.f, 0.f};
        // waves 0-3 (one per SIMD) run decoder then kNN, waves 4-7 the other way round, so the MFMA
        // pipe of each SIMD always has a decoder wave while its partner does the VALU-only kNN.
#pragma unroll 1
        for (int phase = 0; phase < 2; ++phase) {
            const bool do_dec = (phase == 0) == (wave < 4);
            if (do_dec) {
#pragma unroll 1
                for (int tile = wave; tile < ntiles; tile += OPT_THREADS / 64) {
                    const int pt = min(tile * 32 + (lane & 31), K - 1);
                    const f32x4 x = X[pt];
                    float logit, bce, dx[3];
                    decoder_tile<MODE_OPT, true>(W, pl, x.x, x.y, x.z, lane, dc, A.threshold, inv_lb, logit,
                                                 bce, dx);
                    if (lane < 32 && tile * 32 + lane < K) G[pt] = f32x4{dx[0], dx[1], dx[2], bce};
                }
            } else if (use_rep) {
                Top5 ta, tb;
                knn_scan2(X, K, pa, pb, ta, tb);
                if (pa < K) rep_point(X, F, pa, ta, rc, rep_loss_a, gca, true);
                if (pb < K) rep_point(X, F, pb, tb, rc, rep_loss_b, gcb, true);
            }
        }
        __syncthreads();
        if (last && loss_out != nullptr) {   // losses at the pre-update points of the last step
            float occ = (pa < K ? G[pa].w : 0.f) + (pb < K ? G[pb].w : 0.f);
            float rep = (pa < K ? rep_loss_a : 0.f) + (pb < K ? rep_loss_b : 0.f);
            occ = wave_sum(occ);
            rep = wave_sum(rep);
            float (*red)[OPT_THREADS / 64] = reinterpret_cast<float (*)[OPT_THREADS / 64]>(scratch);
            if (lane == 0) { red[0][wave] = occ; red[1][wave] = rep; }
            __syncthreads();
            if (tid == 0) {
                float so = 0.f, sr = 0.f;
                for (int w = 0; w < OPT_THREADS / 64; ++w) { so += red[0][w]; sr += red[1][w]; }
                loss_out[2 * cloud + 0] = so;
                loss_out[2 * cloud + 1] = sr / ((float)K * 5.f);
            }
        }
        // ---- Adam (torch/optim/adam.py _single_tensor_adam: lerp form, eps added after the bias-
        //      corrected sqrt) ---------------------------------------------------------------------
        b1t *= 0.9;
        b2t *= 0.999;
        const float step_size = (float)((double)A.lr / (1.0 - b1t));
        const float bc2 = (float)sqrt(1.0 - b2t);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pt = q ? pb : pa;
            if (pt < K) {
                const f32x4 go = G[pt];
                f32x4 x = X[pt];
                const float gocc[3] = {go.x, go.y, go.z};
                float xs[3] = {x.x, x.y, x.z};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float gn = __ll2float_rn(F[3 * pt + a]) * FIX_INV;
                    F[3 * pt + a] = 0;
                    const float gr = ((q ? gcb[a] : gca[a]) + gn) * rep_scale;
                    const float g = gocc[a] + gr;
                    float& mr = mm[3 * q + a];
                    float& vr = vv[3 * q + a];
                    mr = mr + (g - mr) * (1.f - 0.9f);
                    vr = vr * 0.999f + (1.f - 0.999f) * g * g;
                    const float denom = sqrtf(vr) / bc2 + 1e-8f;
                    xs[a] = xs[a] - step_size * (mr / denom);
                }
                X[pt] = f32x4{xs[0], xs[1], xs[2], 0.f};
            }
        }
        __syncthreads();
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
                    m_io[((size_t)cloud * K + pt) * 3 + a] = mm[3 * q + a];
                    v_io[((size_t)cloud * K + pt) * 3 + a] = vv[3 * q + a];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// stand-alone entry kernels (same device functions; used by ifd_decode / ifd_repulsion and tests)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OPT_THREADS, 2) void decode_kernel(const float* __restrict__ dec_img,
                                                                 const float* __restrict__ planes,
                                                                 const float* __restrict__ p, int K,
                                                                 float* __restrict__ logits,
                                                                 float* __restrict__ dlogit_dp, DecConst dc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;
    const int cloud = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    load_dec_image(W, dec_img);
    __syncthreads();
    const float* pl = planes + (size_t)cloud * CLOUD_PLANE_FLOATS;
    const float* pc = p + (size_t)cloud * K * 3;
    const int ntiles = (K + 31) >> 5;
    for (int tile = wave; tile < ntiles; tile += OPT_THREADS / 64) {
        const int pt = min(tile * 32 + (lane & 31), K - 1);
        const float x0 = pc[3 * pt], x1 = pc[3 * pt + 1], x2 = pc[3 * pt + 2];
        float logit, bce, dx[3] = {0.f, 0.f, 0.f};
        if (dlogit_dp != nullptr)
            decoder_tile<MODE_SUM, true>(W, pl, x0, x1, x2, lane, dc, 0.f, 1.f, logit, bce, dx);
        else
            decoder_tile<MODE_SUM, false>(W, pl, x0, x1, x2, lane, dc, 0.f, 1.f, logit, bce, dx);
        if (lane < 32 && tile * 32 + lane < K) {
            logits[(size_t)cloud * K + pt] = logit;
            if (dlogit_dp != nullptr) {
                float* o = dlogit_dp + ((size_t)cloud * K + pt) * 3;
                o[0] = dx[0]; o[1] = dx[1]; o[2] = dx[2];
            }
        }
    }
}

__global__ __launch_bounds__(OPT_THREADS, 2) void repulsion_kernel(const float* __restrict__ p, int K,
                                                                    float* __restrict__ loss,
                                                                    float* __restrict__ grad,
                                                                    int32_t* __restrict__ knn_idx, RepConst rc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* X = reinterpret_cast<f32x4*>(smem);
    long long* F = reinterpret_cast<long long*>(X + MAXK);
    float* scratch = reinterpret_cast<float*>(F + 3 * MAXK);
    const int cloud = blockIdx.x, tid = threadIdx.x;
    const float* pc = p + (size_t)cloud * K * 3;
    const int pa = tid, pb = tid + OPT_THREADS;
    if (pa < K) X[pa] = f32x4{pc[3 * pa], pc[3 * pa + 1], pc[3 * pa + 2], 0.f};
    if (pb < K) X[pb] = f32x4{pc[3 * pb], pc[3 * pb + 1], pc[3 * pb + 2], 0.f};
    for (int i = tid; i < MAXK * 3; i += OPT_THREADS) F[i] = 0;
    __syncthreads();
    Top5 ta, tb;
    knn_scan2(X, K, pa, pb, ta, tb);
    float la = 0.f, lb = 0.f, gca[3] = {0.f, 0.f, 0.f}, gcb[3] = {0.f, 0.f, 0.f};
    if (pa < K) rep_point(X, F, pa, ta, rc, la, gca, grad != nullptr);
    if (pb < K) rep_point(X, F, pb, tb, rc, lb, gcb, grad != nullptr);
    if (knn_idx != nullptr) {
        if (pa < K) {
            int32_t* o = knn_idx + ((size_t)cloud * K + pa) * 5;
            o[0] = ta.i0; o[1] = ta.i1; o[2] = ta.i2; o[3] = ta.i3; o[4] = ta.i4;
        }
        if (pb < K) {
            int32_t* o = knn_idx + ((size_t)cloud * K + pb) * 5;
            o[0] = tb.i0; o[1] = tb.i1; o[2] = tb.i2; o[3] = tb.i3; o[4] = tb.i4;
        }
    }
    const float tot = block_sum((pa < K ? la : 0.f) + (pb < K ? lb : 0.f), scratch);
    if (tid == 0) loss[cloud] = tot / ((float)K * 5.f);
    if (grad != nullptr) {
        const float sc = 1.f / ((float)K * 5.f);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pt = q ? pb : pa;
            if (pt < K)
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    grad[((size_t)cloud * K + pt) * 3 + a] =
                        ((q ? gcb[a] : gca[a]) + __ll2float_rn(F[3 * pt + a]) * FIX_INV) * sc;
        }
    }
}

__global__ __launch_bounds__(OPT_THREADS, 2) void normalize_kernel(float* __restrict__ p, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* X = reinterpret_cast<f32x4*>(smem);
    float* scratch = reinterpret_cast<float*>(X + MAXK);
    float* pc = p + (size_t)blockIdx.x * K * 3;
    const int pa = threadIdx.x, pb = threadIdx.x + OPT_THREADS;
    if (pa < K) X[pa] = f32x4{pc[3 * pa], pc[3 * pa + 1], pc[3 * pa + 2], 0.f};
    if (pb < K) X[pb] = f32x4{pc[3 * pb], pc[3 * pb + 1], pc[3 * pb + 2], 0.f};
    __syncthreads();
    normalize_in_lds(X, K, scratch);
    if (pa < K) { pc[3 * pa] = X[pa].x; pc[3 * pa + 1] = X[pa].y; pc[3 * pa + 2] = X[pa].z; }
    if (pb < K) { pc[3 * pb] = X[pb].x; pc[3 * pb + 1] = X[pb].y; pc[3 * pb + 2] = X[pb].z; }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
constexpr size_t OPT_LDS = DEC_FLOATS * 4 + MAXK * 16 * 2 + MAXK * 3 * 8 + 256;     // 123,536 B
constexpr size_t DEC_LDS = DEC_FLOATS * 4;
constexpr size_t REP_LDS = MAXK * 16 + MAXK * 3 * 8 + 64;
constexpr size_t NRM_LDS = MAXK * 16 + 64;

hipError_t configure_optimize_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(optimize_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)OPT_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEC_LDS);
    return e;
}

hipError_t launch_optimize(const float* dec_img, const float* planes, float* p, float* m, float* v,
                           float* loss, int B, int K, const OptArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(optimize_kernel, dim3(B), dim3(OPT_THREADS), OPT_LDS, s, dec_img, planes, p, m, v, loss,
                       K, a);
    return hipGetLastError();
}

hipError_t launch_decode(const float* dec_img, const float* planes, const float* p, int B, int K,
                         float* logits, float* dlogit_dp, DecConst dc, hipStream_t s) {
    hipLaunchKernelGGL(decode_kernel, dim3(B), dim3(OPT_THREADS), DEC_LDS, s, dec_img, planes, p, K, logits,
                       dlogit_dp, dc);
    return hipGetLastError();
}

hipError_t launch_repulsion(const float* p, int B, int K, float* loss, float* grad, int32_t* knn_idx,
                            float radius, float h, float eps, hipStream_t s) {
    RepConst rc = {radius, h, eps};
    hipLaunchKernelGGL(repulsion_kernel, dim3(B), dim3(OPT_THREADS), REP_LDS, s, p, K, loss, grad, knn_idx, rc);
    return hipGetLastError();
}

hipError_t launch_normalize(float* p, int B, int K, hipStream_t s) {
    hipLaunchKernelGGL(normalize_kernel, dim3(B), dim3(OPT_THREADS), NRM_LDS, s, p, K);
    return hipGetLastError();
}

}  // namespace ifd
