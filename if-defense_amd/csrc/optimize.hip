// ConvONet-Opt hot loop for MI355X (gfx950): one workgroup (8 waves, 2 per SIMD, 256 VGPRs each) owns one
// cloud for all Adam steps; one thread owns two points for the kNN / Adam phases, one wave owns a 16-point
// tile at a time for the decoder phase.
//
//   per step (reference: ConvONet/opt_defense.py:210-228)
//     phase A  decoder forward + input-gradient on 16-point tiles:
//              bilinear gather of the 3 channel-last planes (decoder.py:50-57), the 5-block ResNet MLP on
//              v_mfma_f32_16x16x4_f32 (decoder.py:83-93, layers.py:39-48), BCE-to-threshold derivative
//              (opt_defense.py:213-216), transposed MLP, dc/du from the taps still held in registers.
//              All 24 tap loads of a tile are issued as one batch (one memory round trip per tile).
//     phase B  exact 5-NN from certified neighbour lists (defense/pn_utils.py:64-83) + repulsion loss
//              gradient (defense/repulsion_loss.py:43-54); neighbour terms are scattered with 64-bit
//              fixed-point LDS atomics => order independent, bit reproducible.
//     phase C  fused Adam update (torch.optim.Adam single-tensor form), moments in registers.
//   Half of the waves run A then B, the other half B then A, so every SIMD always has MFMA work queued
//   next to the VALU-only kNN work.  Nothing but the plane taps (and the L2-resident neighbour lists) is
//   read from global memory inside the loop.
//
// MFMA operand mapping (16x16x4, f32): M = output channel, N = point of the tile, K = input channel.
//   lane l = (n = l & 15, q = l >> 4).  A-operand: A[m = n][k = q];  B-operand: B[k = q][n];  C/D: lane
//   (n, q), register r  <->  row 4q + r, column n.   With two M-tiles (mt = 0, 1) a lane therefore holds, for
//   point n, the 8 channels 16 mt + 4 q + r  (register e = 4 mt + r).  MFMA step s = 4 mt' + r' consumes
//   register s as its B operand (k-slot q <-> channel 16 mt' + 4 q + r'), i.e. the accumulator layout of
//   one layer IS the B-operand layout of the next: activations never leave registers and are never
//   transposed.  Weights stream from LDS as the A operand (one ds_read_b32 per MFMA, see ifd_device.h).
#include "ifd_device.h"
#include "ifd_internal.h"

namespace ifd {

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Acc2 {
    f32x4 t[2];     // M-tile 0 (channels 4q..4q+3) and M-tile 1 (channels 16+4q..)
};

// out[o][n] (+)= sum_c A[o][c] * in[c][n] with A = W (forward) or W^T (backward); 16 MFMAs, two
// independent accumulator chains (the 40-cycle dependent latency of 16x16x4 is covered by alternating).
// `lo` is the lane's offset into a layer (LaneOff below), made opaque once per tile so that the loop-invariant
// LDS weight loads are not hoisted out of the tile loop by LICM (they would be spilled to scratch).
struct LaneOff {
    int fwd;    // n * S + q
    int bwd;    // 4 q * S + wperm16(n)
    int q4;     // 4 q
};

template <bool TRANSPOSED>
__device__ __forceinline__ Acc2 dense32(const float* __restrict__ wl, const LaneOff& lo, const f32x8& in, Acc2 acc) {
    // forward : A = W[16 mt + n][16 mt' + 4 q + r']  at  (16 mt + n) * S + 16 mt' + 4 r' + q
    // backward: A = W[16 mt' + 4 q + r'][16 mt + n]  at  (16 mt' + 4 q + r') * S + 16 mt + wperm16(n)
    const float* base = wl + (TRANSPOSED ? lo.bwd : lo.fwd);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int mtp = s >> 2, rp = s & 3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int off = TRANSPOSED ? ((16 * mtp + rp) * W_STRIDE + 16 * mt) : (16 * mt * W_STRIDE + 16 * mtp + 4 * rp);
            acc.t[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(base[off], in[s], acc.t[mt], 0, 0, 0);
        }
    }
    // Keep the next layer's weight loads behind this layer.  (An explicit one-layer-ahead register prefetch
    // of the A operands was measured slower: +32 VGPRs of live fragments push spills into the tile loop.)
    __builtin_amdgcn_sched_barrier(0);
    return acc;
}

__device__ __forceinline__ Acc2 load_bias(const float* __restrict__ W, int layer, const LaneOff& lo) {
    Acc2 b;
    b.t[0] = *reinterpret_cast<const f32x4*>(W + DEC_OFF_BIAS + layer * 32 + lo.q4);
    b.t[1] = *reinterpret_cast<const f32x4*>(W + DEC_OFF_BIAS + layer * 32 + 16 + lo.q4);
    return b;
}

__device__ __forceinline__ f32x8 flat(const Acc2& a) {
    f32x8 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o[r] = a.t[0][r]; o[4 + r] = a.t[1][r]; }
    return o;
}

__device__ __forceinline__ uint32_t mask_pos(const f32x8& v) {
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) m |= (v[r] > 0.f ? 1u : 0u) << r;
    return m;
}

// ReLU as an integer max on the float bits: one v_max_i32 instead of hipcc's canonicalise + v_max_f32 pair
// (negative floats, -0.0 included, are negative integers; positive floats keep their bits; MFMA never yields NaN
// from finite inputs here).
__device__ __forceinline__ f32x8 relu8(const f32x8& v) {
    f32x8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = __int_as_float(max(__float_as_int(v[r]), 0));
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

// One 16-point tile on one wave.  The 4 lanes (n, q = 0..3) share point n and hold 8 of its 32 channels each.
// Returns logit, the BCE term and d(loss)/dx (valid on every lane after the quad reduce).
// TAPMODE 0: taps gathered one plane at a time and re-gathered for the backward pass
//         1: all 24 tap loads in one batch, kept in 96 VGPRs for the backward (needs ~256 VGPRs: 2 waves/SIMD)
//         2: all 24 tap loads in one batch, reduced at once to c (8) and the Jacobian d c / d x (24 VGPRs); the
//            backward pass is J^T dc - no memory access, no taps held
template <int MODE, bool WANT_GRAD, int TAPMODE>
__device__ __forceinline__ void decoder_tile(const float* __restrict__ W, const float* __restrict__ planes,
                                             float x0, float x1, float x2, int lane, const DecConst dc,
                                             float thr, float inv_lb, float& logit_out, float& bce_out,
                                             float (&dx)[3]) {
    const int n = lane & 15, q = lane >> 4;
    LaneOff lo = {n * W_STRIDE + q, 4 * q * W_STRIDE + 4 * (n & 3) + (n >> 2), 4 * q};
    asm volatile("" : "+v"(lo.fwd), "+v"(lo.bwd), "+v"(lo.q4));
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

    // ---- gather + forward: c = sum over planes of the bilinear sample --------------------------------
    constexpr bool HOLD = TAPMODE != 0;       // batch all 24 loads
    constexpr bool JAC = TAPMODE == 2;
    f32x4 tap[HOLD ? 3 : 1][4][2];            // [plane][nw, ne, sw, se][M-tile]
    f32x8 c, J[JAC ? 3 : 1];
#pragma unroll
    for (int r = 0; r < 8; ++r) c[r] = 0.f;
    if (JAC) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < 8; ++r) J[JAC ? a : 0][r] = 0.f;
    }
    if (HOLD) {
#pragma unroll
        for (int P = 0; P < 3; ++P) {
            const int a0 = AX0[P], a1 = AX1[P];
            const float* qp = planes + ((P * RES + cell[a1]) * RES + cell[a0]) * CH + 4 * q;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                tap[HOLD ? P : 0][0][mt] = *reinterpret_cast<const f32x4*>(qp + 16 * mt);
                tap[HOLD ? P : 0][1][mt] = *reinterpret_cast<const f32x4*>(qp + CH + 16 * mt);
                tap[HOLD ? P : 0][2][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + 16 * mt);
                tap[HOLD ? P : 0][3][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + CH + 16 * mt);
            }
        }
    }
    const float jsc = ((0.5f * (float)(RES - 1)) * 2.f) / dc.sdiv;
#pragma unroll
    for (int P = 0; P < 3; ++P) {
        const int a0 = AX0[P], a1 = AX1[P];
        const int tp = HOLD ? P : 0;
        if (!HOLD) {
            const float* qp = planes + ((P * RES + cell[a1]) * RES + cell[a0]) * CH + 4 * q;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                tap[tp][0][mt] = *reinterpret_cast<const f32x4*>(qp + 16 * mt);
                tap[tp][1][mt] = *reinterpret_cast<const f32x4*>(qp + CH + 16 * mt);
                tap[tp][2][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + 16 * mt);
                tap[tp][3][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + CH + 16 * mt);
            }
        }
        const float wnw = w0[a0] * w0[a1], wne = w1[a0] * w0[a1], wsw = w0[a0] * w1[a1], wse = w1[a0] * w1[a1];
        const float s0 = live[a0] * jsc, s1 = live[a1] * jsc;
        const float k0n = s0 * w0[a1], k0s = s0 * w1[a1];       // d/du0: (ne - nw) w0[a1] + (se - sw) w1[a1]
        const float k1w = s1 * w0[a0], k1e = s1 * w1[a0];       // d/du1: (sw - nw) w0[a0] + (se - ne) w1[a0]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float nw = tap[tp][0][mt][j], ne = tap[tp][1][mt][j], sw = tap[tp][2][mt][j], se = tap[tp][3][mt][j];
                float s = nw * wnw;
                s = fmaf(ne, wne, s);
                s = fmaf(sw, wsw, s);
                s = fmaf(se, wse, s);
                c[4 * mt + j] += s;
                if (JAC) {
                    J[JAC ? a0 : 0][4 * mt + j] += fmaf(se - sw, k0s, (ne - nw) * k0n);
                    J[JAC ? a1 : 0][4 * mt + j] += fmaf(se - ne, k1e, (sw - nw) * k1w);
                }
            }
        if (!HOLD) __builtin_amdgcn_sched_barrier(0);   // one plane's 8 tap loads in flight at a time
    }
    if (JAC) __builtin_amdgcn_sched_barrier(0);          // the taps die here

    // ---- forward MLP ------------------------------------------------------------------------------
    Acc2 net;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WP + (16 * mt + j) * 4 + lo.q4 * 4);
            net.t[mt][j] = fmaf(wp.z, x2, fmaf(wp.y, x1, fmaf(wp.x, x0, wp.w)));
        }
    uint32_t mask_a[NBLK], mask_h[NBLK];
    const float* Wd = W + DEC_OFF_W;
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const float* Wl = Wd + 3 * i * W_LAYER;
        Acc2 a = load_bias(W, 3 * i, lo);
        a.t[0] += net.t[0];
        a.t[1] += net.t[1];
        a = dense32<false>(Wl, lo, c, a);                                       // a_i = n_i + fc_c[i](c)
        const f32x8 af = flat(a);
        mask_a[i] = mask_pos(af);
        const Acc2 h = dense32<false>(Wl + W_LAYER, lo, relu8(af), load_bias(W, 3 * i + 1, lo));   // fc_0(relu(a))
        const f32x8 hf = flat(h);
        mask_h[i] = mask_pos(hf);
        Acc2 o = load_bias(W, 3 * i + 2, lo);
        o.t[0] += a.t[0];
        o.t[1] += a.t[1];
        net = dense32<false>(Wl + 2 * W_LAYER, lo, relu8(hf), o);               // a + fc_1(relu(h))
    }
    const f32x8 nf = flat(net);
    const uint32_t mask_n = mask_pos(nf);
    f32x8 wout;
    {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WOUT + lo.q4);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WOUT + 16 + lo.q4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { wout[r] = t0[r]; wout[4 + r] = t1[r]; }
    }
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) part = fmaf(wout[r], fmaxf(nf[r], 0.f), part);
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    const float logit = part + W[DEC_OFF_BOUT];
    logit_out = logit;
    bce_out = 0.f;
    if (!WANT_GRAD) return;

    // ---- backward (parameters frozen: only the path to the input) ----------------------------------
    float dl;
    if (MODE == MODE_OPT) {
        const float e = expf(-fabsf(logit));
        bce_out = fmaxf(logit, 0.f) - thr * logit + log1pf(e);
        const float sig = logit >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        dl = (sig - thr) * inv_lb;
    } else {
        dl = 1.f;
    }
    f32x8 dn;
#pragma unroll
    for (int r = 0; r < 8; ++r) dn[r] = ((mask_n >> r) & 1u) ? dl * wout[r] : 0.f;
    Acc2 zero;
    zero.t[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    zero.t[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    Acc2 dcc = zero;
#pragma unroll
    for (int i = NBLK - 1; i >= 0; --i) {
        const float* Wl = Wd + 3 * i * W_LAYER;
        f32x8 dh = flat(dense32<true>(Wl + 2 * W_LAYER, lo, dn, zero));
#pragma unroll
        for (int r = 0; r < 8; ++r) dh[r] = ((mask_h[i] >> r) & 1u) ? dh[r] : 0.f;
        const f32x8 t = flat(dense32<true>(Wl + W_LAYER, lo, dh, zero));
#pragma unroll
        for (int r = 0; r < 8; ++r) dn[r] += ((mask_a[i] >> r) & 1u) ? t[r] : 0.f;   // delta a_i
        dcc = dense32<true>(Wl, lo, dn, dcc);                                          // += Wc^T delta a_i
    }
    const f32x8 dcf = flat(dcc);
    float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WP + (16 * mt + j) * 4 + lo.q4 * 4);
            const float d = dn[4 * mt + j];
            g[0] = fmaf(wp.x, d, g[0]); g[1] = fmaf(wp.y, d, g[1]); g[2] = fmaf(wp.z, d, g[2]);
        }
    if (JAC) {     // d loss / d x through the sampled features: J^T dc (no memory access)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float acc = g[a];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc = fmaf(J[JAC ? a : 0][r], dcf[r], acc);
            g[a] = acc;
        }
    }
    // d c / d u through the bilinear taps (grid_sampler_2d backward w.r.t. the grid)
#pragma unroll
    for (int P = 0; P < (JAC ? 0 : 3); ++P) {
        const int a0 = AX0[P], a1 = AX1[P];
        const int tp = HOLD ? P : 0;
        if (!HOLD) {
            int off = ((P * RES + cell[a1]) * RES + cell[a0]) * CH + 4 * q;
            // opaque to the optimiser: otherwise these loads are CSE'd with the forward gather and the taps are
            // kept live (and spilled) across the whole MLP.  The re-read is L1/L2 traffic.
            asm volatile("" : "+v"(off));
            const float* qp = planes + off;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                tap[tp][0][mt] = *reinterpret_cast<const f32x4*>(qp + 16 * mt);
                tap[tp][1][mt] = *reinterpret_cast<const f32x4*>(qp + CH + 16 * mt);
                tap[tp][2][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + 16 * mt);
                tap[tp][3][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + CH + 16 * mt);
            }
        }
        float dnw = 0.f, dne = 0.f, dsw = 0.f, dse = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = dcf[4 * mt + j];
                dnw = fmaf(tap[tp][0][mt][j], d, dnw); dne = fmaf(tap[tp][1][mt][j], d, dne);
                dsw = fmaf(tap[tp][2][mt][j], d, dsw); dse = fmaf(tap[tp][3][mt][j], d, dse);
            }
        const float gix = (dne - dnw) * w0[a1] + (dse - dsw) * w1[a1];
        const float giy = (dsw - dnw) * w0[a0] + (dse - dne) * w1[a0];
        const float sc = (0.5f * (float)(RES - 1)) * 2.f;
        g[a0] += live[a0] * ((gix * sc) / dc.sdiv);
        g[a1] += live[a1] * ((giy * sc) / dc.sdiv);
        if (!HOLD) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float v = g[a];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        dx[a] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Two 16-point sub-tiles per wave, processed in lock-step (the optimiser's phase A).
//   * 4 independent accumulator chains per layer (2 sub-tiles x 2 M-tiles): no dependent-MFMA stalls, and the
//     VALU epilogue of one sub-tile (bias, ReLU, mask) overlaps the other sub-tile's MFMAs;
//   * every A operand (weight) is fetched from LDS once and feeds both sub-tiles: half the LDS reads per MFMA;
//   * ReLU masks are packed 8 bits per layer and made opaque (asm) so the compiler keeps them as 1 VGPR each.
// The taps are gathered one plane at a time and re-gathered for the backward pass (L1/L2 hits): holding them
// for two sub-tiles would need 192 VGPRs.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mask_pos_packed(const f32x8& v) {
    uint32_t b[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) b[r] = v[r] > 0.f ? (1u << r) : 0u;
    uint32_t m = ((b[0] | b[1]) | (b[2] | b[3])) | ((b[4] | b[5]) | (b[6] | b[7]));
    asm volatile("" : "+v"(m));      // keep it a bit-mask: otherwise hipcc re-expands it into 8 float selectors
    return m;
}

struct SubGeo {                      // per-point sampling geometry of one sub-tile lane
    float x[3], w0[3], w1[3], live[3];
    int cell[3];
};

__device__ __forceinline__ void sub_geometry(SubGeo& g, float x0, float x1, float x2, const DecConst& dc) {
    g.x[0] = x0; g.x[1] = x1; g.x[2] = x2;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float pix;
        pixel_coord(g.x[a], dc, pix, g.live[a]);
        const int ci = min((int)floorf(pix), RES - 2);
        g.cell[a] = ci;
        g.w1[a] = pix - (float)ci;
        g.w0[a] = ((float)ci + 1.f) - pix;
    }
}

// two sub-tiles, shared A operands: 32 MFMAs per layer, 16 LDS reads
template <bool TRANSPOSED>
__device__ __forceinline__ void dense32x2(const float* __restrict__ wl, const LaneOff& lo, const f32x8& in0,
                                          const f32x8& in1, Acc2& acc0, Acc2& acc1) {
    const float* base = wl + (TRANSPOSED ? lo.bwd : lo.fwd);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int mtp = s >> 2, rp = s & 3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int off = TRANSPOSED ? ((16 * mtp + rp) * W_STRIDE + 16 * mt) : (16 * mt * W_STRIDE + 16 * mtp + 4 * rp);
            const float a = base[off];
            acc0.t[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, in0[s], acc0.t[mt], 0, 0, 0);
            acc1.t[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, in1[s], acc1.t[mt], 0, 0, 0);
        }
    }
}

template <int MODE>
__device__ __forceinline__ void decoder_tile2(const float* __restrict__ W, const float* __restrict__ planes,
                                              const f32x4 xa, const f32x4 xb, int lane, const DecConst dc, float thr,
                                              float inv_lb, float (&bce)[2], float (&dx)[2][3]) {
    const int n = lane & 15, q = lane >> 4;
    LaneOff lo = {n * W_STRIDE + q, 4 * q * W_STRIDE + 4 * (n & 3) + (n >> 2), 4 * q};
    asm volatile("" : "+v"(lo.fwd), "+v"(lo.bwd), "+v"(lo.q4));
    constexpr int AX0[3] = {0, 0, 1}, AX1[3] = {2, 1, 2};   // xz, xy, yz (common.py:243-248)
    SubGeo geo[2];
    sub_geometry(geo[0], xa.x, xa.y, xa.z, dc);
    sub_geometry(geo[1], xb.x, xb.y, xb.z, dc);

    // ---- gather + forward bilinear sample, one plane (2 x 8 loads) at a time ---------------------------------
    f32x8 c[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 8; ++r) c[t][r] = 0.f;
#pragma unroll
    for (int P = 0; P < 3; ++P) {
        const int a0 = AX0[P], a1 = AX1[P];
        f32x4 tap[2][4][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* qp = planes + ((P * RES + geo[t].cell[a1]) * RES + geo[t].cell[a0]) * CH + 4 * q;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                tap[t][0][mt] = *reinterpret_cast<const f32x4*>(qp + 16 * mt);
                tap[t][1][mt] = *reinterpret_cast<const f32x4*>(qp + CH + 16 * mt);
                tap[t][2][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + 16 * mt);
                tap[t][3][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + CH + 16 * mt);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const SubGeo& g = geo[t];
            const float wnw = g.w0[a0] * g.w0[a1], wne = g.w1[a0] * g.w0[a1], wsw = g.w0[a0] * g.w1[a1],
                        wse = g.w1[a0] * g.w1[a1];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s = tap[t][0][mt][j] * wnw;
                    s = fmaf(tap[t][1][mt][j], wne, s);
                    s = fmaf(tap[t][2][mt][j], wsw, s);
                    s = fmaf(tap[t][3][mt][j], wse, s);
                    c[t][4 * mt + j] += s;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- forward MLP ----------------------------------------------------------------------------------------
    Acc2 net[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WP + (16 * mt + j) * 4 + lo.q4 * 4);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                net[t].t[mt][j] = fmaf(wp.z, geo[t].x[2], fmaf(wp.y, geo[t].x[1], fmaf(wp.x, geo[t].x[0], wp.w)));
        }
    uint32_t mask_a[2][NBLK], mask_h[2][NBLK];
    const float* Wd = W + DEC_OFF_W;
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const float* Wl = Wd + 3 * i * W_LAYER;
        Acc2 a[2], h[2], o[2];
        {
            const Acc2 b = load_bias(W, 3 * i, lo);
#pragma unroll
            for (int t = 0; t < 2; ++t) { a[t].t[0] = b.t[0] + net[t].t[0]; a[t].t[1] = b.t[1] + net[t].t[1]; }
        }
        dense32x2<false>(Wl, lo, c[0], c[1], a[0], a[1]);                       // a_i = n_i + fc_c[i](c)
        f32x8 ra[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x8 af = flat(a[t]);
            mask_a[t][i] = mask_pos_packed(af);
            ra[t] = relu8(af);
        }
        {
            const Acc2 b = load_bias(W, 3 * i + 1, lo);
            h[0] = b; h[1] = b;
        }
        dense32x2<false>(Wl + W_LAYER, lo, ra[0], ra[1], h[0], h[1]);          // fc_0(relu(a))
        f32x8 rh[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x8 hf = flat(h[t]);
            mask_h[t][i] = mask_pos_packed(hf);
            rh[t] = relu8(hf);
        }
        {
            const Acc2 b = load_bias(W, 3 * i + 2, lo);
#pragma unroll
            for (int t = 0; t < 2; ++t) { o[t].t[0] = b.t[0] + a[t].t[0]; o[t].t[1] = b.t[1] + a[t].t[1]; }
        }
        dense32x2<false>(Wl + 2 * W_LAYER, lo, rh[0], rh[1], o[0], o[1]);      // a + fc_1(relu(h))
        net[0] = o[0];
        net[1] = o[1];
        __builtin_amdgcn_sched_barrier(0);      // one block at a time: bounds the weight-load hoisting
    }
    f32x8 wout;
    {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WOUT + lo.q4);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WOUT + 16 + lo.q4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { wout[r] = t0[r]; wout[4 + r] = t1[r]; }
    }
    f32x8 dn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x8 nf = flat(net[t]);
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) part = fmaf(wout[r], fmaxf(nf[r], 0.f), part);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const float logit = part + W[DEC_OFF_BOUT];
        float dl;
        if (MODE == MODE_OPT) {
            const float e = expf(-fabsf(logit));
            bce[t] = fmaxf(logit, 0.f) - thr * logit + log1pf(e);
            const float sig = logit >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            dl = (sig - thr) * inv_lb;
        } else {
            bce[t] = logit;                   // MODE_SUM: report the logit, gradient of sum(logits)
            dl = 1.f;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) dn[t][r] = nf[r] > 0.f ? dl * wout[r] : 0.f;
    }

    // ---- backward (parameters frozen: only the path to the input) -----------------------------------------------
    Acc2 dcc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { dcc[t].t[0] = f32x4{0.f, 0.f, 0.f, 0.f}; dcc[t].t[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int i = NBLK - 1; i >= 0; --i) {
        const float* Wl = Wd + 3 * i * W_LAYER;
        Acc2 z[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { z[t].t[0] = f32x4{0.f, 0.f, 0.f, 0.f}; z[t].t[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        dense32x2<true>(Wl + 2 * W_LAYER, lo, dn[0], dn[1], z[0], z[1]);       // fc_1^T dn
        f32x8 dh[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            dh[t] = flat(z[t]);
#pragma unroll
            for (int r = 0; r < 8; ++r) dh[t][r] = ((mask_h[t][i] >> r) & 1u) ? dh[t][r] : 0.f;
            z[t].t[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            z[t].t[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        dense32x2<true>(Wl + W_LAYER, lo, dh[0], dh[1], z[0], z[1]);           // fc_0^T dh
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x8 tt = flat(z[t]);
#pragma unroll
            for (int r = 0; r < 8; ++r) dn[t][r] += ((mask_a[t][i] >> r) & 1u) ? tt[r] : 0.f;   // delta a_i
        }
        dense32x2<true>(Wl, lo, dn[0], dn[1], dcc[0], dcc[1]);                 // dc += fc_c^T da
        __builtin_amdgcn_sched_barrier(0);
    }
    float g[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) g[t][0] = g[t][1] = g[t][2] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WP + (16 * mt + j) * 4 + lo.q4 * 4);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float d = dn[t][4 * mt + j];
                g[t][0] = fmaf(wp.x, d, g[t][0]); g[t][1] = fmaf(wp.y, d, g[t][1]); g[t][2] = fmaf(wp.z, d, g[t][2]);
            }
        }
    // d c / d u through the bilinear taps (grid_sampler_2d backward w.r.t. the grid), taps re-gathered
    f32x8 dcf[2] = {flat(dcc[0]), flat(dcc[1])};
#pragma unroll
    for (int P = 0; P < 3; ++P) {
        const int a0 = AX0[P], a1 = AX1[P];
        f32x4 tap[2][4][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int off = ((P * RES + geo[t].cell[a1]) * RES + geo[t].cell[a0]) * CH + 4 * q;
            asm volatile("" : "+v"(off));      // opaque: do not CSE with (and keep alive since) the forward gather
            const float* qp = planes + off;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                tap[t][0][mt] = *reinterpret_cast<const f32x4*>(qp + 16 * mt);
                tap[t][1][mt] = *reinterpret_cast<const f32x4*>(qp + CH + 16 * mt);
                tap[t][2][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + 16 * mt);
                tap[t][3][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + CH + 16 * mt);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const SubGeo& gg = geo[t];
            float dnw = 0.f, dne = 0.f, dsw = 0.f, dse = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = dcf[t][4 * mt + j];
                    dnw = fmaf(tap[t][0][mt][j], d, dnw); dne = fmaf(tap[t][1][mt][j], d, dne);
                    dsw = fmaf(tap[t][2][mt][j], d, dsw); dse = fmaf(tap[t][3][mt][j], d, dse);
                }
            const float gix = (dne - dnw) * gg.w0[a1] + (dse - dsw) * gg.w1[a1];
            const float giy = (dsw - dnw) * gg.w0[a0] + (dse - dne) * gg.w1[a0];
            const float sc = (0.5f * (float)(RES - 1)) * 2.f;
            g[t][a0] += gg.live[a0] * ((gix * sc) / dc.sdiv);
            g[t][a1] += gg.live[a1] * ((giy * sc) / dc.sdiv);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = g[t][a];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            dx[t][a] = v;
        }
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined version of the two-sub-tile schedule (SCHED = 2).
//   region k.1 :  16 MFMAs of sub-tile 0, layer k   ||  VALU epilogue (bias/ReLU/mask) of sub-tile 1, layer k-1
//   region k.2 :  16 MFMAs of sub-tile 1, layer k   ||  VALU epilogue of sub-tile 0, layer k  ||  LDS prefetch of
//                 layer k+1's 16 A operands + bias into registers
// so the matrix pipe never waits for an epilogue or an LDS round trip of its own wave.  An in-order wave only
// overlaps what is adjacent in its instruction stream, hence the explicit sched_group_barrier interleave
// (1 MFMA : n VALU : m DS-read) inside every region and a sched_barrier between regions.
// ---------------------------------------------------------------------------------------------
struct WFrag {
    float a[16];    // a[2 * s + mt]
};

template <bool TRANSPOSED>
__device__ __forceinline__ WFrag load_wfrag(const float* __restrict__ wl, const LaneOff& lo) {
    const float* base = wl + (TRANSPOSED ? lo.bwd : lo.fwd);
    WFrag f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int mtp = s >> 2, rp = s & 3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int off = TRANSPOSED ? ((16 * mtp + rp) * W_STRIDE + 16 * mt) : (16 * mt * W_STRIDE + 16 * mtp + 4 * rp);
            f.a[2 * s + mt] = base[off];
        }
    }
    return f;
}

__device__ __forceinline__ void mfma16(const WFrag& f, const f32x8& in, Acc2& acc) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            acc.t[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[2 * s + mt], in[s], acc.t[mt], 0, 0, 0);
}

// interleave pattern of one region: 16 x { 1 MFMA, NV VALU, ND DS reads }, then close the region
template <int NV, int ND>
__device__ __forceinline__ void region_end() {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        if (ND > 0) __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ Acc2 acc_add(const Acc2& a, const Acc2& b) {
    Acc2 r;
    r.t[0] = a.t[0] + b.t[0];
    r.t[1] = a.t[1] + b.t[1];
    return r;
}
__device__ __forceinline__ Acc2 acc_zero() {
    Acc2 r;
    r.t[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    r.t[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    return r;
}
__device__ __forceinline__ f32x8 masked(const Acc2& z, uint32_t m) {
    f32x8 v = flat(z);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = ((m >> r) & 1u) ? v[r] : 0.f;
    return v;
}

template <int MODE>
__device__ __forceinline__ void decoder_tile3(const float* __restrict__ W, const float* __restrict__ planes,
                                              const f32x4 xa, const f32x4 xb, int lane, const DecConst dc, float thr,
                                              float inv_lb, float (&bce)[2], float (&dx)[2][3]) {
    const int n = lane & 15, q = lane >> 4;
    LaneOff lo = {n * W_STRIDE + q, 4 * q * W_STRIDE + 4 * (n & 3) + (n >> 2), 4 * q};
    asm volatile("" : "+v"(lo.fwd), "+v"(lo.bwd), "+v"(lo.q4));
    constexpr int AX0[3] = {0, 0, 1}, AX1[3] = {2, 1, 2};   // xz, xy, yz (common.py:243-248)
    SubGeo geo[2];
    sub_geometry(geo[0], xa.x, xa.y, xa.z, dc);
    sub_geometry(geo[1], xb.x, xb.y, xb.z, dc);
    const float* Wd = W + DEC_OFF_W;

    // weights of the first layer ride along with the gather
    WFrag A = load_wfrag<false>(Wd, lo);
    Acc2 B = load_bias(W, 0, lo);

    // ---- gather + forward bilinear sample, one plane (2 x 8 loads) at a time ---------------------------------
    f32x8 c[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 8; ++r) c[t][r] = 0.f;
#pragma unroll
    for (int P = 0; P < 3; ++P) {
        const int a0 = AX0[P], a1 = AX1[P];
        f32x4 tap[2][4][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* qp = planes + ((P * RES + geo[t].cell[a1]) * RES + geo[t].cell[a0]) * CH + 4 * q;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                tap[t][0][mt] = *reinterpret_cast<const f32x4*>(qp + 16 * mt);
                tap[t][1][mt] = *reinterpret_cast<const f32x4*>(qp + CH + 16 * mt);
                tap[t][2][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + 16 * mt);
                tap[t][3][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + CH + 16 * mt);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const SubGeo& g = geo[t];
            const float wnw = g.w0[a0] * g.w0[a1], wne = g.w1[a0] * g.w0[a1], wsw = g.w0[a0] * g.w1[a1],
                        wse = g.w1[a0] * g.w1[a1];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s = tap[t][0][mt][j] * wnw;
                    s = fmaf(tap[t][1][mt][j], wne, s);
                    s = fmaf(tap[t][2][mt][j], wsw, s);
                    s = fmaf(tap[t][3][mt][j], wse, s);
                    c[t][4 * mt + j] += s;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    Acc2 net[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WP + (16 * mt + j) * 4 + lo.q4 * 4);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                net[t].t[mt][j] = fmaf(wp.z, geo[t].x[2], fmaf(wp.y, geo[t].x[1], fmaf(wp.x, geo[t].x[0], wp.w)));
        }
    __builtin_amdgcn_sched_barrier(0);

    // ---- forward MLP, software pipelined ---------------------------------------------------------------------
    uint32_t mask_a[2][NBLK], mask_h[2][NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const float* Wl = Wd + 3 * i * W_LAYER;
        // fc_c: a = n + fc_c(c)
        Acc2 a0 = acc_add(B, net[0]), a1 = acc_add(B, net[1]);
        mfma16(A, c[0], a0);                                                  // R1
        region_end<0, 0>();
        const WFrag A0 = load_wfrag<false>(Wl + W_LAYER, lo);                 // R2: prefetch fc_0
        const Acc2 B0 = load_bias(W, 3 * i + 1, lo);
        mfma16(A, c[1], a1);
        const f32x8 af0 = flat(a0);
        mask_a[0][i] = mask_pos_packed(af0);
        const f32x8 ra0 = relu8(af0);
        region_end<2, 1>();
        Acc2 h0 = B0;                                                          // R3
        mfma16(A0, ra0, h0);
        const f32x8 af1 = flat(a1);
        mask_a[1][i] = mask_pos_packed(af1);
        const f32x8 ra1 = relu8(af1);
        region_end<2, 0>();
        const WFrag A1 = load_wfrag<false>(Wl + 2 * W_LAYER, lo);             // R4: prefetch fc_1
        const Acc2 B1 = load_bias(W, 3 * i + 2, lo);
        Acc2 h1 = B0;
        mfma16(A0, ra1, h1);
        const f32x8 hf0 = flat(h0);
        mask_h[0][i] = mask_pos_packed(hf0);
        const f32x8 rh0 = relu8(hf0);
        region_end<2, 1>();
        Acc2 o0 = acc_add(B1, a0);                                             // R5
        mfma16(A1, rh0, o0);
        const f32x8 hf1 = flat(h1);
        mask_h[1][i] = mask_pos_packed(hf1);
        const f32x8 rh1 = relu8(hf1);
        region_end<2, 0>();
        if (i + 1 < NBLK) {                                                    // R6: prefetch next fc_c / first fc_1^T
            A = load_wfrag<false>(Wl + 3 * W_LAYER, lo);
            B = load_bias(W, 3 * i + 3, lo);
        } else {
            A = load_wfrag<true>(Wl + 2 * W_LAYER, lo);
        }
        Acc2 o1 = acc_add(B1, a1);
        mfma16(A1, rh1, o1);
        net[0] = o0;
        net[1] = o1;
        region_end<0, 1>();
    }
    f32x8 wout;
    {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WOUT + lo.q4);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WOUT + 16 + lo.q4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { wout[r] = t0[r]; wout[4 + r] = t1[r]; }
    }
    f32x8 dn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x8 nf = flat(net[t]);
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) part = fmaf(wout[r], fmaxf(nf[r], 0.f), part);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const float logit = part + W[DEC_OFF_BOUT];
        float dl;
        if (MODE == MODE_OPT) {
            const float e = expf(-fabsf(logit));
            bce[t] = fmaxf(logit, 0.f) - thr * logit + log1pf(e);
            const float sig = logit >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            dl = (sig - thr) * inv_lb;
        } else {
            bce[t] = logit;
            dl = 1.f;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) dn[t][r] = nf[r] > 0.f ? dl * wout[r] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- backward, software pipelined (A holds fc_1[4]^T) ----------------------------------------------------------
    Acc2 dcc[2] = {acc_zero(), acc_zero()};
#pragma unroll
    for (int i = NBLK - 1; i >= 0; --i) {
        const float* Wl = Wd + 3 * i * W_LAYER;
        Acc2 z0 = acc_zero();                                                  // R1: fc_1^T dn (sub-tile 0)
        mfma16(A, dn[0], z0);
        region_end<0, 0>();
        const WFrag A0 = load_wfrag<true>(Wl + W_LAYER, lo);                  // R2: prefetch fc_0^T
        Acc2 z1 = acc_zero();
        mfma16(A, dn[1], z1);
        const f32x8 dh0 = masked(z0, mask_h[0][i]);
        region_end<1, 1>();
        Acc2 y0 = acc_zero();                                                  // R3: fc_0^T dh (sub-tile 0)
        mfma16(A0, dh0, y0);
        const f32x8 dh1 = masked(z1, mask_h[1][i]);
        region_end<1, 0>();
        const WFrag Ac = load_wfrag<true>(Wl, lo);                            // R4: prefetch fc_c^T
        Acc2 y1 = acc_zero();
        mfma16(A0, dh1, y1);
        {
            const f32x8 t = masked(y0, mask_a[0][i]);
#pragma unroll
            for (int r = 0; r < 8; ++r) dn[0][r] += t[r];                      // delta a_i
        }
        region_end<1, 1>();
        mfma16(Ac, dn[0], dcc[0]);                                             // R5: dc += fc_c^T da (sub-tile 0)
        {
            const f32x8 t = masked(y1, mask_a[1][i]);
#pragma unroll
            for (int r = 0; r < 8; ++r) dn[1][r] += t[r];
        }
        region_end<1, 0>();
        if (i > 0) A = load_wfrag<true>(Wl - W_LAYER, lo);                    // R6: prefetch fc_1[i-1]^T
        mfma16(Ac, dn[1], dcc[1]);
        region_end<0, 1>();
    }
    float g[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) g[t][0] = g[t][1] = g[t][2] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WP + (16 * mt + j) * 4 + lo.q4 * 4);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float d = dn[t][4 * mt + j];
                g[t][0] = fmaf(wp.x, d, g[t][0]); g[t][1] = fmaf(wp.y, d, g[t][1]); g[t][2] = fmaf(wp.z, d, g[t][2]);
            }
        }
    // d c / d u through the bilinear taps (grid_sampler_2d backward w.r.t. the grid), taps re-gathered
    f32x8 dcf[2] = {flat(dcc[0]), flat(dcc[1])};
#pragma unroll
    for (int P = 0; P < 3; ++P) {
        const int a0 = AX0[P], a1 = AX1[P];
        f32x4 tap[2][4][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int off = ((P * RES + geo[t].cell[a1]) * RES + geo[t].cell[a0]) * CH + 4 * q;
            asm volatile("" : "+v"(off));      // opaque: do not CSE with (and keep alive since) the forward gather
            const float* qp = planes + off;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                tap[t][0][mt] = *reinterpret_cast<const f32x4*>(qp + 16 * mt);
                tap[t][1][mt] = *reinterpret_cast<const f32x4*>(qp + CH + 16 * mt);
                tap[t][2][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + 16 * mt);
                tap[t][3][mt] = *reinterpret_cast<const f32x4*>(qp + RES * CH + CH + 16 * mt);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const SubGeo& gg = geo[t];
            float dnw = 0.f, dne = 0.f, dsw = 0.f, dse = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = dcf[t][4 * mt + j];
                    dnw = fmaf(tap[t][0][mt][j], d, dnw); dne = fmaf(tap[t][1][mt][j], d, dne);
                    dsw = fmaf(tap[t][2][mt][j], d, dsw); dse = fmaf(tap[t][3][mt][j], d, dse);
                }
            const float gix = (dne - dnw) * gg.w0[a1] + (dse - dsw) * gg.w1[a1];
            const float giy = (dsw - dnw) * gg.w0[a0] + (dse - dne) * gg.w1[a0];
            const float sc = (0.5f * (float)(RES - 1)) * 2.f;
            g[t][a0] += gg.live[a0] * ((gix * sc) / dc.sdiv);
            g[t][a1] += gg.live[a1] * ((giy * sc) / dc.sdiv);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = g[t][a];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            dx[t][a] = v;
        }
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

// Branch-free sorted insertion (bubble the new key down with selects).  NOTE: the obvious nested-ternary
// form is turned into exec-mask control flow by hipcc (4x the code, I-cache thrash) - keep this shape.
__device__ __forceinline__ void top5_insert_nb(Top5& t, float d, int j) {
    bool c; float lo; int li;
    c = d < t.d0; lo = c ? d : t.d0; li = c ? j : t.i0; d = c ? t.d0 : d; j = c ? t.i0 : j; t.d0 = lo; t.i0 = li;
    c = d < t.d1; lo = c ? d : t.d1; li = c ? j : t.i1; d = c ? t.d1 : d; j = c ? t.i1 : j; t.d1 = lo; t.i1 = li;
    c = d < t.d2; lo = c ? d : t.d2; li = c ? j : t.i2; d = c ? t.d2 : d; j = c ? t.i2 : j; t.d2 = lo; t.i2 = li;
    c = d < t.d3; lo = c ? d : t.d3; li = c ? j : t.i3; d = c ? t.d3 : d; j = c ? t.i3 : j; t.d3 = lo; t.i3 = li;
    c = d < t.d4; t.d4 = c ? d : t.d4; t.i4 = c ? j : t.i4;
}

__device__ __forceinline__ void top5_insert(Top5& t, float d, int j) {
    if (d < t.d4) top5_insert_nb(t, d, j);      // rare after the first few dozen candidates
}

// Exact brute-force scan of all K points (broadcast LDS reads) for the two points of this thread
// (self excluded by index).
__device__ __forceinline__ void knn_scan2(const f32x4* __restrict__ X, int K, int pa, int pb, Top5& ta, Top5& tb) {
    const f32x4 xa = X[min(pa, K - 1)], xb = X[min(pb, K - 1)];
    top5_init(ta);
    top5_init(tb);
#pragma unroll 4
    for (int j = 0; j < K; ++j) {
        const f32x4 xj = X[j];
        const float ax = xj.x - xa.x, ay = xj.y - xa.y, az = xj.z - xa.z;
        const float bx = xj.x - xb.x, by = xj.y - xb.y, bz = xj.z - xb.z;
        float da = fmaf(az, az, fmaf(ay, ay, ax * ax));
        float db = fmaf(bz, bz, fmaf(by, by, bx * bx));
        da = (j == pa) ? INFINITY : da;
        db = (j == pb) ? INFINITY : db;
        top5_insert(ta, da, j);
        top5_insert(tb, db, j);
    }
}

// ---------------------------------------------------------------------------------------------
// Certified neighbour lists: exact 5-NN at O(LIST_M) per point per step.
//
//   build (rare, all waves of the cloud in the same step):  for each point i store every j with
//       |x_j - x_i| < rho_i, rho_i^2 = alpha2_i * (an upper bound of i's squared 5-NN distance); alpha2_i
//       adapts so that the ball holds <= LIST_M points.  x0_i = x_i at build time.
//   step:  the 5 nearest list members are the true 5-NN iff  r5 < rho_i - |x_i - x0_i| - Dmax, with
//       Dmax = max_j |x_j - x0_j|  (a point outside the list was >= rho_i away at build time).
//       soft margin violated -> request a synchronous rebuild for the NEXT step (nobody stalls alone);
//       certificate violated -> this wave runs the exact brute-force scan for this step.
//   Either way every step uses the exact 5-NN set; ties follow ascending j like the scan.
// ---------------------------------------------------------------------------------------------
constexpr int LIST_F = 16;               // "front": every point within rho_f at build time (evaluated every step)
constexpr int LIST_B = 32;               // "back" : the ring rho_f <= d < rho_b (evaluated only when the front fails)
constexpr int LIST_M = LIST_F + LIST_B;  // uint16 entries per point; lists live in global memory (L2-resident)

// 5 nearest of points ia / ib among entries [E0, E1) of their lists, continuing the running top-5 in ta / tb.
// The two independent insertion chains are interleaved for ILP.  Entries >= cnt are ignored.
template <int E0, int E1>
__device__ __forceinline__ void list_top5_2(const f32x4* __restrict__ X, const uint16_t* La, const uint16_t* Lb,
                                            int cnt_a, int cnt_b, int ia, int ib, Top5& ta, Top5& tb) {
    constexpr int NC = (E1 - E0) / 8;
    const f32x4 xa = X[ia], xb = X[ib];
    u32x4 wa[NC], wb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        wa[c] = reinterpret_cast<const u32x4*>(La)[E0 / 8 + c];
        wb[c] = reinterpret_cast<const u32x4*>(Lb)[E0 / 8 + c];
    }
    // One entry of each list per block.  The sched_barrier keeps every compare next to the selects that consume
    // it: left alone, the scheduler runs the distance chains ahead and parks dozens of lane masks in spilled SGPRs
    // (v_writelane / v_readlane + s_nop per compare - 2x the instructions).  The next entry's coordinates are
    // fetched one block ahead to cover the LDS latency.
    int ja = (int)(wa[0][0] & 0xffffu), jb = (int)(wb[0][0] & 0xffffu);
    bool va = E0 < cnt_a, vb = E0 < cnt_b;
    f32x4 na = X[va ? ja : ia], nb = X[vb ? jb : ib];
#pragma unroll
    for (int e = E0; e < E1; ++e) {
        const f32x4 pa_ = na, pb_ = nb;
        const int cja = ja, cjb = jb;
        const bool cva = va, cvb = vb;
        if (e + 1 < E1) {
            const int r = e + 1 - E0;
            const unsigned int pka = wa[r >> 3][(r & 7) >> 1], pkb = wb[r >> 3][(r & 7) >> 1];
            ja = (r & 1) ? (int)(pka >> 16) : (int)(pka & 0xffffu);
            jb = (r & 1) ? (int)(pkb >> 16) : (int)(pkb & 0xffffu);
            va = e + 1 < cnt_a;
            vb = e + 1 < cnt_b;
            na = X[va ? ja : ia];
            nb = X[vb ? jb : ib];
        }
        const float ax = pa_.x - xa.x, ay = pa_.y - xa.y, az = pa_.z - xa.z;
        const float bx = pb_.x - xb.x, by = pb_.y - xb.y, bz = pb_.z - xb.z;
        float da = fmaf(az, az, fmaf(ay, ay, ax * ax));
        float db = fmaf(bz, bz, fmaf(by, by, bx * bx));
        da = cva ? da : INFINITY;
        db = cvb ? db : INFINITY;
        top5_insert_nb(ta, da, cja);
        top5_insert_nb(tb, db, cjb);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- fast evaluation: 32-bit keys = (distance bits with the low 10 mantissa bits replaced by the index) --------
// Positive floats order like their bit patterns, so a running sorted top-6 of keys needs one v_min_u32 and five
// v_med3_u32 per entry - no compare masks, no index selects (16 instead of ~33 VALU ops per entry).  The 5 smallest
// keys are EXACTLY the 5 nearest entries whenever key 5 and key 6 differ above the index bits (every other entry
// is then strictly farther than all five); otherwise (relative distance gap < 2^-13 at the 5/6 boundary, rare) the
// caller falls back to the exact insertion path.  The order inside the five is irrelevant: the gradient sums are
// fixed-point (rep_point).  Unused list slots hold index MAXK, the far-away dummy point X[MAXK].
constexpr unsigned int KEY_IDX_MASK = 1023u;
struct Keys6 {
    unsigned int k0, k1, k2, k3, k4, k5;
};
__device__ __forceinline__ void keys6_init(Keys6& q) { q.k0 = q.k1 = q.k2 = q.k3 = q.k4 = q.k5 = 0xffffffffu; }
__device__ __forceinline__ unsigned int umed3(unsigned int a, unsigned int b, unsigned int c) {
    unsigned int r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void keys6_insert(Keys6& q, unsigned int x) {
    q.k5 = umed3(q.k4, q.k5, x);      // new k_i = median(old k_{i-1}, old k_i, x)
    q.k4 = umed3(q.k3, q.k4, x);
    q.k3 = umed3(q.k2, q.k3, x);
    q.k2 = umed3(q.k1, q.k2, x);
    q.k1 = umed3(q.k0, q.k1, x);
    q.k0 = min(q.k0, x);
}
// wa / wb: the LIST_M / 8 packed index words of the two lists (loaded by the caller in ONE global round trip);
// chunks [C0, C1) of 8 entries are evaluated.  Per chunk all 16 coordinate reads are issued before the first use:
// the phase is latency-bound (a lone VALU wave next to an MFMA wave), so LDS round trips are batched, not chained.
template <int C0, int C1>
__device__ __forceinline__ void list_keys6_2(const f32x4* __restrict__ X, const u32x4 (&wa)[LIST_M / 8],
                                             const u32x4 (&wb)[LIST_M / 8], int ia, int ib, Keys6& qa, Keys6& qb) {
    const f32x4 xa = X[ia], xb = X[ib];
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        unsigned int ja[8], jb[8];
        f32x4 pa_[8], pb_[8];
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
            const unsigned int pka = wa[c][e8 >> 1], pkb = wb[c][e8 >> 1];
            ja[e8] = (e8 & 1) ? (pka >> 16) : (pka & 0xffffu);
            jb[e8] = (e8 & 1) ? (pkb >> 16) : (pkb & 0xffffu);
        }
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) { pa_[e8] = X[ja[e8]]; pb_[e8] = X[jb[e8]]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
            const float ax = pa_[e8].x - xa.x, ay = pa_[e8].y - xa.y, az = pa_[e8].z - xa.z;
            const float bx = pb_[e8].x - xb.x, by = pb_[e8].y - xb.y, bz = pb_[e8].z - xb.z;
            const float da = fmaf(az, az, fmaf(ay, ay, ax * ax));
            const float db = fmaf(bz, bz, fmaf(by, by, bx * bx));
            keys6_insert(qa, (__float_as_uint(da) & ~KEY_IDX_MASK) | ja[e8]);
            keys6_insert(qb, (__float_as_uint(db) & ~KEY_IDX_MASK) | jb[e8]);
        }
    }
}
__device__ __forceinline__ void load_list_words(const uint16_t* La, const uint16_t* Lb, u32x4 (&wa)[LIST_M / 8],
                                                u32x4 (&wb)[LIST_M / 8]) {
#pragma unroll
    for (int c = 0; c < LIST_M / 8; ++c) {
        wa[c] = reinterpret_cast<const u32x4*>(La)[c];
        wb[c] = reinterpret_cast<const u32x4*>(Lb)[c];
    }
}
// upper bound of the squared distance of key k (INF for the init value / NaN patterns)
__device__ __forceinline__ float key_d_upper(unsigned int k) {
    return k >= 0x7f800000u ? INFINITY : __uint_as_float(k | KEY_IDX_MASK);
}
__device__ __forceinline__ bool keys6_ambiguous(const Keys6& q) { return ((q.k4 ^ q.k5) & ~KEY_IDX_MASK) == 0u; }
__device__ __forceinline__ void keys6_to_top5(const Keys6& q, Top5& t) {
    t.i0 = (int)(q.k0 & KEY_IDX_MASK); t.i1 = (int)(q.k1 & KEY_IDX_MASK); t.i2 = (int)(q.k2 & KEY_IDX_MASK);
    t.i3 = (int)(q.k3 & KEY_IDX_MASK); t.i4 = (int)(q.k4 & KEY_IDX_MASK);
    t.d0 = t.d1 = t.d2 = t.d3 = 0.f;
    t.d4 = key_d_upper(q.k4);
}

// Per-point list state kept by the owning lane.
struct KnnPt {
    int cnt_f, cnt_b;   // valid entries of the front / back segment; cnt_b = -1: no valid list (ball too crowded)
    float rho_f;        // the front holds EVERY point that was within rho_f at build time (0: front not complete)
    float rho_b;        // front + back hold every point that was within rho_b at build time
    float al_f, al_b;   // alpha^2 of the two radii: rho^2 = al * (upper bound of the squared 5-NN distance)
    f32x4 x0;           // position at build time
    float dbase;        // Dmax at this point's build time (0 for whole-cloud rebuilds, which start a new epoch)
    float r5p;          // last step's 5-NN distance (upper bound)
    bool frag;          // certificate too short-lived to be worth a whole-cloud rebuild: refreshed individually
    bool pend;          // individual refresh requested for the next step
};

// One target of the wave-cooperative ("transposed") list build: the lanes hold the K candidate points in registers
// (16 each); one distance per candidate, hits are compacted with ballot / mbcnt into the target's list - no divergent
// branches, exact counts.  tf / tb: squared front / back radii (wave-uniform).  Unused slots get the dummy index.
__device__ __forceinline__ void knn_build_one(const float (&cx)[16], const float (&cy)[16], const float (&cz)[16],
                                              const f32x4 xi, int i, int lane, float tf, float tb,
                                              uint16_t* __restrict__ lst, int& nf_out, int& nb_out) {
    int nf = 0, nb = 0;                                                            // wave-uniform running counts
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = lane + 64 * k;
        const float ex = cx[k] - xi.x, ey = cy[k] - xi.y, ez = cz[k] - xi.z;
        const float d = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
        const bool in_b = d < tb && j != i;
        const bool in_f = d < tf && in_b;
        const unsigned long long mf = __ballot(in_f);
        const int pos_f = nf + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mf >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mf, 0u));
        const bool to_f = in_f && pos_f < LIST_F;
        if (to_f) lst[pos_f] = (uint16_t)j;
        nf += __popcll(mf);
        const bool to_b = in_b && !to_f;                  // ring members + front hits that did not fit
        const unsigned long long mb = __ballot(to_b);
        const int pos_b = nb + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u));
        if (to_b && pos_b < LIST_B) lst[LIST_F + pos_b] = (uint16_t)j;
        nb += __popcll(mb);
    }
    // unused slots point at the dummy X[MAXK] (the key evaluation does not look at counts)
    if (lane < LIST_M && lane >= (lane < LIST_F ? nf : LIST_F + nb)) lst[lane] = (uint16_t)MAXK;
    nf_out = nf;
    nb_out = nb;
}

__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// (Re)build the lists of the flagged points of this wave (need_a / need_b per lane; d4a / d4b: upper bounds of their
// squared 5-NN distances at the current positions).  Targets are taken one at a time off the ballot mask, so the
// cost is ~1 us for loading the candidates plus ~0.5 us per target - whole-cloud rebuilds (all flagged) and the
// individual refreshes of short-lived ("fragile") certificates share this code.  A ball that overflows its list is
// shrunk in proportion to the overshoot (hit count ~ r^2 on a surface) and rebuilt on the spot.
__device__ __forceinline__ void knn_refresh(const f32x4* __restrict__ X, int K, int wave, int lane,
                                            uint16_t* __restrict__ lists, bool need_a, bool need_b, float d4a,
                                            float d4b, KnnPt& ka, KnnPt& kb, float dbase, float mv,
                                            unsigned int& n_targets, unsigned int& n_repass) {
    float cx[16], cy[16], cz[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = lane + 64 * k;
        const f32x4 p = X[min(j, K - 1)];
        const bool v = j < K;
        cx[k] = v ? p.x : 1e18f; cy[k] = v ? p.y : 1e18f; cz[k] = v ? p.z : 1e18f;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        KnnPt& kp = half ? kb : ka;
        const float d4 = half ? d4b : d4a;
        const float my_tf = d4 * kp.al_f, my_tb = d4 * kp.al_b;
        unsigned long long mask = __ballot(half ? need_b : need_a);
#pragma unroll 1
        while (mask != 0ull) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1ull;
            const int i = (half ? OPT_THREADS : 0) + wave * 64 + l;
            float tf = readlane_f(my_tf, l), tb = readlane_f(my_tb, l);
            const f32x4 xi = X[i];                                                   // wave-uniform address
            uint16_t* lst = lists + (size_t)i * LIST_M;
            int nf, nb;
            ++n_targets;
#pragma unroll 1
            for (int rep = 0;; ++rep) {
                knn_build_one(cx, cy, cz, xi, i, lane, tf, tb, lst, nf, nb);
                if ((nf <= LIST_F && nb <= LIST_B) || rep == 3) break;
                ++n_repass;
                const float n_in = (float)(min(nf, LIST_F) + nb);                   // points inside the back radius
                float nf_new = (float)nf;
                if (nf > LIST_F) { tf *= (0.7f * LIST_F) / (float)nf; nf_new = 0.7f * LIST_F; }
                if (n_in - nf_new > 0.85f * LIST_B) tb *= (nf_new + 0.75f * LIST_B) / n_in;
                tb = fmaxf(tb, tf);
            }
            if (lane == l) {
                kp.cnt_f = min(nf, LIST_F);
                kp.rho_f = nf <= LIST_F ? sqrtf(tf) : 0.f;          // front complete only if everything fitted
                kp.cnt_b = nb <= LIST_B ? nb : -1;
                kp.rho_b = sqrtf(tb);
                // carry the (possibly shrunk) radii forward as multiples of the 5-NN bound; grow slowly when sparse
                if (d4 > 0.f) { kp.al_f = tf / d4; kp.al_b = tb / d4; }
                if (nf < LIST_F / 2) kp.al_f *= 1.15f;
                if (nf + nb < LIST_M / 2) kp.al_b *= 1.15f;
                kp.al_f = fminf(fmaxf(kp.al_f, 1.1f), 6.f);
                kp.al_b = fminf(fmaxf(kp.al_b, 1.2f), 30.f);
                kp.al_f = fminf(kp.al_f, kp.al_b);
                kp.x0 = xi;
                kp.dbase = dbase;
                // expected lifetime of the certificate ~ (rho - r5 - 6 mv) / (~2 mv per step)
#ifndef IFD_FRAG_MULT
#define IFD_FRAG_MULT 16.f
#endif
                kp.frag = kp.rho_b - sqrtf(d4) < IFD_FRAG_MULT * mv;
            }
        }
    }
}

struct RepConst {
    float radius, h, eps;
};

// Loss and gradient terms of one centre point (repulsion_loss.py:43-53).  The centre part is
// returned in gc (un-scaled), neighbour parts go to the fixed-point LDS accumulator.
__device__ __forceinline__ void rep_point(const f32x4* __restrict__ X, long long* __restrict__ F, int i,
                                          const Top5& t, const RepConst rc, float& loss, long long (&gc)[3],
                                          bool want_grad) {
    const f32x4 xi = X[i];
    const int idx[5] = {t.i0, t.i1, t.i2, t.i3, t.i4};
    loss = 0.f;
    gc[0] = gc[1] = gc[2] = 0;
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
            // fixed point on both ends: the sums do not depend on the order of the neighbours or of the atomics,
            // and the centre receives exactly minus what its neighbours receive
            const long long fx = __float2ll_rn(gx * FIX_SCALE), fy = __float2ll_rn(gy * FIX_SCALE),
                            fz = __float2ll_rn(gz * FIX_SCALE);
            gc[0] -= fx; gc[1] -= fy; gc[2] -= fz;
            atomicAdd(reinterpret_cast<unsigned long long*>(F + 3 * j + 0), (unsigned long long)fx);
            atomicAdd(reinterpret_cast<unsigned long long*>(F + 3 * j + 1), (unsigned long long)fy);
            atomicAdd(reinterpret_cast<unsigned long long*>(F + 3 * j + 2), (unsigned long long)fz);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// block helpers
// ---------------------------------------------------------------------------------------------
constexpr int OWN_WAVES = OPT_THREADS / 64;   // waves whose threads own points (kNN / Adam duty): threads [0, 512)
constexpr int MAX_WAVES = 16;

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
// deterministic block reductions; `scratch` holds >= NWAVES floats
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
    const bool owner = threadIdx.x < OPT_THREADS;
    const int pa = owner ? (int)threadIdx.x : MAXK, pb = owner ? (int)threadIdx.x + OPT_THREADS : MAXK;
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
// NW waves per workgroup (8: 2 per SIMD / 256 VGPRs, 12: 3 per SIMD / 168 VGPRs); threads [0,512) own two points
// each (kNN + Adam duty), every wave pulls 16-point decoder tiles from an LDS counter.
template <int NW, int SCHED>
__global__ __launch_bounds__(NW * 64, NW / 4) void optimize_kernel(
    const float* __restrict__ dec_img, const float* __restrict__ planes, float* __restrict__ p,
    float* __restrict__ m_io, float* __restrict__ v_io, float* __restrict__ loss_out,
    const int32_t* __restrict__ loss_batch_per_cloud, uint16_t* knn_lists,
    unsigned long long* __restrict__ counters, int K, OptArgs A) {
    constexpr bool HOLD = SCHED == 0;                    // single 16-point tiles (else 32-point super-tiles)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W = smem;                                                 // decoder parameter image
    f32x4* G = reinterpret_cast<f32x4*>(smem + DEC_FLOATS);          // occupancy gradient (+ BCE term in .w)
    f32x4* X = G + MAXK;                                             // current points; X[MAXK] = far-away dummy
    long long* F = reinterpret_cast<long long*>(X + MAXK + 1);       // fixed-point neighbour-gradient scatter
    float* scratch = reinterpret_cast<float*>(F + 3 * MAXK);         // 128 floats

    const int cloud = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pl = planes + (A.shared_planes ? (size_t)0 : (size_t)cloud * CLOUD_PLANE_FLOATS);
    float* pc = p + (size_t)cloud * K * 3;
    const bool owner = tid < OPT_THREADS;
    const int pa = owner ? tid : MAXK, pb = owner ? tid + OPT_THREADS : MAXK;   // the two points this thread owns
    const int ntiles = HOLD ? (K + 15) >> 4 : (K + 31) >> 5;          // HOLD: 16-point tiles, else 32-point super-tiles

    const unsigned long long t_begin = __builtin_readcyclecounter();     // shader clock (s_memtime)
    load_dec_image(W, dec_img);
    float mm[6], vv[6];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pt = q ? pb : pa;
#pragma unroll
        for (int a = 0; a < 3; ++a) mm[3 * q + a] = vv[3 * q + a] = 0.f;
        if (pt < K) {
            X[pt] = f32x4{pc[3 * pt], pc[3 * pt + 1], pc[3 * pt + 2], 0.f};
            if (A.t0 > 0 && m_io != nullptr) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    mm[3 * q + a] = m_io[((size_t)cloud * K + pt) * 3 + a];
                    vv[3 * q + a] = v_io[((size_t)cloud * K + pt) * 3 + a];
                }
            }
        }
    }
    for (int i = tid; i < MAXK * 3; i += NW * 64) F[i] = 0;
    if (tid == 0) X[MAXK] = f32x4{1e18f, 1e18f, 1e18f, 0.f};

    const DecConst dc = A.dc;
    const RepConst rc = {A.rep_radius, A.rep_h, A.rep_eps};
    const int loss_batch = loss_batch_per_cloud ? loss_batch_per_cloud[cloud] : A.loss_batch;
    const float inv_lb = 1.0f / (float)loss_batch;
    const float rep_scale = A.rep_weight / ((float)loss_batch * (float)K * 5.f);
    const bool use_rep = A.rep_weight > 0.f;
    double b1t = pow(0.9, (double)A.t0), b2t = pow(0.999, (double)A.t0);
    float rep_loss_a = 0.f, rep_loss_b = 0.f;
    // certified neighbour lists of the two owned points
    uint16_t* La = knn_lists + ((size_t)cloud * MAXK + (pa & (MAXK - 1))) * LIST_M;   // global (L2-resident)
    uint16_t* Lb = knn_lists + ((size_t)cloud * MAXK + (pb & (MAXK - 1))) * LIST_M;
    KnnPt ka = {0, -1, 0.f, 0.f, 2.0f, 7.0f, f32x4{0.f, 0.f, 0.f, 0.f}, 0.f, 0.f, false, false};   // ~10 front / ~35 total hits on a flat patch
    KnnPt kb = ka;
    uint16_t* cloud_lists = knn_lists + (size_t)cloud * MAXK * LIST_M;
    float* dmaxbuf = scratch + 32;                                   // [2][MAX_WAVES] per-wave max |x - x0| (next step)
    float* movebuf = scratch + 64;                                   // [2][MAX_WAVES] per-wave max single-step move
    volatile int* rebuild_flag = reinterpret_cast<volatile int*>(scratch + 28);   // [2] by step parity
    int* tile_ctr = reinterpret_cast<int*>(scratch + 30);            // next decoder tile of this step
    unsigned int n_rebuild = 0, n_brute = 0, n_pass = 0, n_tier2 = 0, n_exact = 0, n_refresh = 0, n_targets = 0;
#ifdef IFD_PROF      // cycle accounting of cloud 0 (diagnostic builds only; overwrites the counters)
    unsigned long long pc_build = 0, pc_eval = 0, pc_rep = 0, pc_tiles = 0, pc_wait = 0, pc_adam = 0, pc_t = 0;
#define PROF_T0() pc_t = __builtin_readcyclecounter()
#define PROF_ACC(v) do { const unsigned long long n_ = __builtin_readcyclecounter(); v += n_ - pc_t; pc_t = n_; } while (0)
#else
#define PROF_T0()
#define PROF_ACC(v)
#endif
    if (tid < 2) rebuild_flag[tid] = 0;
    if (tid == 0) *tile_ctr = 0;
    if (tid < 2 * MAX_WAVES) { dmaxbuf[tid] = 0.f; movebuf[tid] = 2.f * A.lr; }
    __syncthreads();

    for (int step = 0; step < A.steps; ++step) {
        const bool last = step == A.steps - 1;
        long long gca[3] = {0, 0, 0}, gcb[3] = {0, 0, 0};
        // Work of one step: ntiles decoder tiles (pulled from an LDS counter by every wave) + the kNN/repulsion of
        // the 8 owner waves.  Waves 4-7 do their kNN first, waves 0-3 after a few tiles, so that each SIMD always
        // has MFMA work queued next to the VALU-only kNN; waves >= 8 (3-per-SIMD configuration) only pull tiles.
        const int quota = wave < 4 ? (ntiles + 2 * NW - 1) / (2 * NW) : 0;
#pragma unroll 1
        for (int phase = 0; phase < 2; ++phase) {
            if (phase == 1 && wave < OWN_WAVES && use_rep) {
                Top5 ta, tb;
                top5_init(ta);
                top5_init(tb);
                const int ia = min(pa, K - 1), ib = min(pb, K - 1);
                if (A.knn_scan_every_step) {
                    knn_scan2(X, K, pa, pb, ta, tb);
                } else {
                    const bool force = step == 0 || rebuild_flag[step & 1] != 0;      // block-uniform
                    float dmax = 0.f, mv = 0.f;
#pragma unroll
                    for (int w = 0; w < OWN_WAVES; ++w) {
                        dmax = fmaxf(dmax, dmaxbuf[(step & 1) * MAX_WAVES + w]);
                        mv = fmaxf(mv, movebuf[(step & 1) * MAX_WAVES + w]);
                    }
                    // the certificate must survive one more step: r5 grows <= 2 mv, both displacements <= mv
                    const float soft_slack = 6.f * mv;
                    PROF_T0();
                    const bool need_a = pa < K && (force || ka.pend), need_b = pb < K && (force || kb.pend);
                    if (__any(need_a || need_b)) {
                        float d4a, d4b;        // upper bounds of the squared 5-NN distances at the current positions
                        if (force) {
                            // ---- synchronous whole-cloud rebuild (every owner wave, this step): new epoch ---------
                            ++n_rebuild;
                            // any 5 members of the current lists (a truncated ring still holds valid points), else scan
                            bool have = step != 0;
                            if (have) {
                                Keys6 qa, qb;
                                keys6_init(qa);
                                keys6_init(qb);
                                u32x4 wa[LIST_M / 8], wb[LIST_M / 8];
                                load_list_words(La, Lb, wa, wb);
                                list_keys6_2<0, LIST_M / 8>(X, wa, wb, ia, ib, qa, qb);
                                ta.d4 = key_d_upper(qa.k4);
                                tb.d4 = key_d_upper(qb.k4);
                                have = ta.d4 < 1e30f && tb.d4 < 1e30f;       // the dummy point is ~3e36 away
                            }
                            if (!__all(have)) {
                                ++n_pass;
                                knn_scan2(X, K, pa, pb, ta, tb);
                            }
                            d4a = ta.d4;
                            d4b = tb.d4;
                            dmax = 0.f;
                        } else {
                            // ---- individual refresh of fragile certificates: r5 grows by at most 2 mv per step ----
                            ++n_refresh;
                            const float ra = ka.r5p + 2.f * mv, rb = kb.r5p + 2.f * mv;
                            d4a = ra * ra;
                            d4b = rb * rb;
                        }
                        knn_refresh(X, K, wave, lane, cloud_lists, need_a, need_b, d4a, d4b, ka, kb, dmax, mv,
                                    n_targets, n_pass);
                    }
                    ka.pend = kb.pend = false;
                    PROF_ACC(pc_build);
                    // ---- tier 1: the front ball -----------------------------------------------------------------
                    const f32x4 xa = X[ia], xb = X[ib];
                    const float da0 = sqrtf((xa.x - ka.x0.x) * (xa.x - ka.x0.x) + (xa.y - ka.x0.y) * (xa.y - ka.x0.y) +
                                            (xa.z - ka.x0.z) * (xa.z - ka.x0.z));
                    const float db0 = sqrtf((xb.x - kb.x0.x) * (xb.x - kb.x0.x) + (xb.y - kb.x0.y) * (xb.y - kb.x0.y) +
                                            (xb.z - kb.x0.z) * (xb.z - kb.x0.z));
                    // displacement budget already spent: own move since the build + everybody else's (since the epoch
                    // reference: now, and at this point's build time).  A list built THIS step is exact as it stands.
                    const float spent_a = da0 + dmax + ka.dbase, spent_b = db0 + dmax + kb.dbase;
                    const float hs_a = need_a ? 0.f : spent_a, hs_b = need_b ? 0.f : spent_b;
                    bool exact = last;       // the reported loss sums the five terms in ascending-distance order
                    bool soft_a = true, soft_b = true;
                    if (!exact) {
                        // ---- fast path: key networks (see list_keys6_2) ------------------------------------------
                        Keys6 qa, qb;
                        keys6_init(qa);
                        keys6_init(qb);
                        u32x4 wa[LIST_M / 8], wb[LIST_M / 8];
                        load_list_words(La, Lb, wa, wb);       // front and ring together: one L2 round trip
                        list_keys6_2<0, LIST_F / 8>(X, wa, wb, ia, ib, qa, qb);
                        float r5a = sqrtf(key_d_upper(qa.k4)), r5b = sqrtf(key_d_upper(qb.k4));
                        // every point outside a ball of build radius rho is now farther than rho - (spent budget)
                        const bool ok1 = (pa >= K || r5a < (ka.rho_f - hs_a) * 0.99999f - 1e-7f) &&
                                         (pb >= K || r5b < (kb.rho_f - hs_b) * 0.99999f - 1e-7f);
                        bool scanned = false;
                        if (!__all(ok1)) {
                            ++n_tier2;
                            list_keys6_2<LIST_F / 8, LIST_M / 8>(X, wa, wb, ia, ib, qa, qb);
                            r5a = sqrtf(key_d_upper(qa.k4));
                            r5b = sqrtf(key_d_upper(qb.k4));
                            const bool hard = (pa >= K || (ka.cnt_b >= 0 && r5a < (ka.rho_b - hs_a) * 0.99999f - 1e-7f)) &&
                                              (pb >= K || (kb.cnt_b >= 0 && r5b < (kb.rho_b - hs_b) * 0.99999f - 1e-7f));
                            if (!__all(hard)) {      // certificate failed: exact scan for this wave, this step
                                ++n_brute;
                                knn_scan2(X, K, pa, pb, ta, tb);
                                scanned = true;
                            }
                            // will it still hold next step?  (crowded balls, cnt_b < 0, are served by the scan anyway)
                            soft_a = pa >= K || ka.cnt_b < 0 || r5a < (ka.rho_b - spent_a) * 0.99999f - 1e-7f - soft_slack;
                            soft_b = pb >= K || kb.cnt_b < 0 || r5b < (kb.rho_b - spent_b) * 0.99999f - 1e-7f - soft_slack;
                        }
                        if (!scanned) {
                            const bool amb = (pa < K && keys6_ambiguous(qa)) || (pb < K && keys6_ambiguous(qb));
                            if (__any(amb)) {
                                exact = true;
                            } else {
                                keys6_to_top5(qa, ta);
                                keys6_to_top5(qb, tb);
                            }
                        }
                    }
                    if (exact) {
                        // ---- exact path: sorted insertion with indices (last step, near-ties) --------------------
                        ++n_exact;
                        soft_a = soft_b = true;
                        top5_init(ta);
                        top5_init(tb);
                        list_top5_2<0, LIST_F>(X, La, Lb, ka.cnt_f, kb.cnt_f, ia, ib, ta, tb);
                        const bool ok1 = (pa >= K || sqrtf(ta.d4) < (ka.rho_f - hs_a) * 0.99999f - 1e-7f) &&
                                         (pb >= K || sqrtf(tb.d4) < (kb.rho_f - hs_b) * 0.99999f - 1e-7f);
                        if (!__all(ok1)) {
                            list_top5_2<LIST_F, LIST_M>(X, La, Lb, LIST_F + ka.cnt_b, LIST_F + kb.cnt_b, ia, ib, ta, tb);
                            const float r5a = sqrtf(ta.d4), r5b = sqrtf(tb.d4);
                            const bool hard = (pa >= K || (ka.cnt_b >= 0 && r5a < (ka.rho_b - hs_a) * 0.99999f - 1e-7f)) &&
                                              (pb >= K || (kb.cnt_b >= 0 && r5b < (kb.rho_b - hs_b) * 0.99999f - 1e-7f));
                            if (!__all(hard)) {
                                ++n_brute;
                                knn_scan2(X, K, pa, pb, ta, tb);
                            }
                            soft_a = pa >= K || ka.cnt_b < 0 || r5a < (ka.rho_b - spent_a) * 0.99999f - 1e-7f - soft_slack;
                            soft_b = pb >= K || kb.cnt_b < 0 || r5b < (kb.rho_b - spent_b) * 0.99999f - 1e-7f - soft_slack;
                        }
                    }
                    ka.r5p = sqrtf(ta.d4);
                    kb.r5p = sqrtf(tb.d4);
                    // a certificate about to expire: fragile ones are refreshed individually next step, the others
                    // mean the epoch is old -> whole-cloud rebuild next step
                    ka.pend = !soft_a && ka.frag;
                    kb.pend = !soft_b && kb.frag;
                    const bool soft_ok = (soft_a || ka.frag) && (soft_b || kb.frag);
                    if (__any(!soft_ok) && lane == 0) rebuild_flag[(step + 1) & 1] = 1;
                    PROF_ACC(pc_eval);
                }
                if (pa < K) rep_point(X, F, pa, ta, rc, rep_loss_a, gca, true);
                if (pb < K) rep_point(X, F, pb, tb, rc, rep_loss_b, gcb, true);
                PROF_ACC(pc_rep);
            }
            PROF_T0();
            // decoder tiles: phase 0 = up to `quota` tiles before the kNN, phase 1 = until the step's tiles run out
#pragma unroll 1
            for (int n = 0; phase == 1 || n < quota; ++n) {
                int tile = 0;
                if (lane == 0) tile = atomicAdd(tile_ctr, 1);
                tile = __shfl(tile, 0);
                if (tile >= ntiles) break;
                if (HOLD) {          // one 16-point tile, taps held in registers
                    const int tp = min(tile * 16 + (lane & 15), K - 1);
                    const f32x4 x = X[tp];
                    float logit, bce, dx[3];
                    decoder_tile<MODE_OPT, true, 1>(W, pl, x.x, x.y, x.z, lane, dc, A.threshold, inv_lb, logit, bce, dx);
                    if (lane < 16 && tile * 16 + lane < K) G[tp] = f32x4{dx[0], dx[1], dx[2], bce};
                } else {             // two 16-point sub-tiles in lock-step
                    const int ia = tile * 32 + (lane & 15), ib = ia + 16;
                    const int tpa = min(ia, K - 1), tpb = min(ib, K - 1);
                    float bce[2], dx[2][3];
                    if (SCHED == 2)
                        decoder_tile3<MODE_OPT>(W, pl, X[tpa], X[tpb], lane, dc, A.threshold, inv_lb, bce, dx);
                    else
                        decoder_tile2<MODE_OPT>(W, pl, X[tpa], X[tpb], lane, dc, A.threshold, inv_lb, bce, dx);
                    if (lane < 16) {
                        if (ia < K) G[tpa] = f32x4{dx[0][0], dx[0][1], dx[0][2], bce[0]};
                        if (ib < K) G[tpb] = f32x4{dx[1][0], dx[1][1], dx[1][2], bce[1]};
                    }
                }
            }
            PROF_ACC(pc_tiles);
        }
        __syncthreads();
        PROF_ACC(pc_wait);
        if (last && loss_out != nullptr) {   // losses at the pre-update points of the last step
            float occ = (pa < K ? G[pa].w : 0.f) + (pb < K ? G[pb].w : 0.f);
            float rep = (pa < K ? rep_loss_a : 0.f) + (pb < K ? rep_loss_b : 0.f);
            occ = wave_sum(occ);
            rep = wave_sum(rep);
            if (lane == 0) { scratch[wave] = occ; scratch[MAX_WAVES + wave] = rep; }
            __syncthreads();
            if (tid == 0) {
                float so = 0.f, sr = 0.f;
                for (int w = 0; w < NW; ++w) { so += scratch[w]; sr += scratch[MAX_WAVES + w]; }
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
        float dmax2 = 0.f, mv2 = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pt = q ? pb : pa;
            if (pt < K) {
                const f32x4 go = G[pt];
                const f32x4 x = X[pt];
                const float gocc[3] = {go.x, go.y, go.z};
                float xs[3] = {x.x, x.y, x.z};
                float msq = 0.f;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float gn = __ll2float_rn(F[3 * pt + a] + (q ? gcb[a] : gca[a])) * FIX_INV;
                    F[3 * pt + a] = 0;
                    const float gr = gn * rep_scale;
                    const float g = gocc[a] + gr;
                    float& mr = mm[3 * q + a];
                    float& vr = vv[3 * q + a];
                    mr = mr + (g - mr) * (1.f - 0.9f);
                    vr = vr * 0.999f + (1.f - 0.999f) * g * g;
                    const float denom = sqrtf(vr) / bc2 + 1e-8f;
                    const float upd = step_size * (mr / denom);
                    xs[a] = xs[a] - upd;
                    msq = fmaf(upd, upd, msq);
                }
                const f32x4 x0 = q ? kb.x0 : ka.x0;
                const float dsq = (xs[0] - x0.x) * (xs[0] - x0.x) + (xs[1] - x0.y) * (xs[1] - x0.y) +
                                  (xs[2] - x0.z) * (xs[2] - x0.z);
                dmax2 = fmaxf(dmax2, sqrtf(dsq) + (q ? kb.dbase : ka.dbase));
                mv2 = fmaxf(mv2, msq);
                X[pt] = f32x4{xs[0], xs[1], xs[2], 0.f};
            }
        }
        dmax2 = wave_max(dmax2);
        mv2 = wave_max(mv2);
        if (lane == 0) {
            dmaxbuf[((step + 1) & 1) * MAX_WAVES + wave] = dmax2 * 1.00001f + 1e-7f;
            movebuf[((step + 1) & 1) * MAX_WAVES + wave] = sqrtf(mv2);
        }
        if (tid == 0) { rebuild_flag[step & 1] = 0; *tile_ctr = 0; }   // both consumed before the mid-step barrier
        __syncthreads();
        PROF_ACC(pc_adam);
    }

    if (counters != nullptr && lane == 0) {
        atomicAdd(counters + 0, (unsigned long long)n_rebuild);   // wave-level list rebuilds
        atomicAdd(counters + 1, (unsigned long long)n_brute);     // wave-level certificate failures (exact scans)
        atomicAdd(counters + 2, (unsigned long long)n_pass);      // extra rebuild work: exact scans for radii + overflow re-passes
        atomicAdd(counters + 4, (unsigned long long)n_tier2);     // wave-steps that had to evaluate the back ring
        atomicAdd(counters + 5, (unsigned long long)n_exact);     // wave-steps on the exact insertion path (last step, near-ties)
        atomicAdd(counters + 6, (unsigned long long)n_refresh);   // wave-steps with individual list refreshes
        atomicAdd(counters + 7, (unsigned long long)n_targets);   // lists built (whole-cloud rebuilds + individual refreshes)
        if (tid == 0 && cloud == 0) counters[3] = __builtin_readcyclecounter() - t_begin;   // shader cycles of cloud 0
#ifdef IFD_PROF
        if (tid == 0) {
            const unsigned long long cyc = __builtin_readcyclecounter() - t_begin;
            atomicMax(counters + 14, cyc);
            atomicMax(counters + 13, (unsigned long long)n_rebuild << 32);
            atomicAdd(counters + 15, cyc);
        }
        if (cloud == 0) {
            atomicAdd(counters + 8, pc_build); atomicAdd(counters + 9, pc_eval); atomicAdd(counters + 10, pc_rep);
            atomicAdd(counters + 11, pc_tiles); atomicAdd(counters + 12, pc_wait); atomicAdd(counters + 13, pc_adam);
        }
#endif
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
    const int ntiles = (K + 15) >> 4;
    for (int tile = wave; tile < ntiles; tile += (int)(blockDim.x >> 6)) {
        const int tp = min(tile * 16 + (lane & 15), K - 1);
        const float x0 = pc[3 * tp], x1 = pc[3 * tp + 1], x2 = pc[3 * tp + 2];
        float logit, bce, dx[3] = {0.f, 0.f, 0.f};
        if (dlogit_dp != nullptr)
            decoder_tile<MODE_SUM, true, 2>(W, pl, x0, x1, x2, lane, dc, 0.f, 1.f, logit, bce, dx);
        else
            decoder_tile<MODE_SUM, false, 1>(W, pl, x0, x1, x2, lane, dc, 0.f, 1.f, logit, bce, dx);
        if (lane < 16 && tile * 16 + lane < K) {
            logits[(size_t)cloud * K + tp] = logit;
            if (dlogit_dp != nullptr) {
                float* o = dlogit_dp + ((size_t)cloud * K + tp) * 3;
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
    float la = 0.f, lb = 0.f;
    long long gca[3] = {0, 0, 0}, gcb[3] = {0, 0, 0};
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
                        __ll2float_rn(F[3 * pt + a] + (q ? gcb[a] : gca[a])) * FIX_INV * sc;
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
constexpr size_t OPT_LDS = DEC_FLOATS * 4 + MAXK * 16 * 2 + 16 + MAXK * 3 * 8 + 128 * 4;   // 125,728 B
constexpr size_t DEC_LDS = DEC_FLOATS * 4;
constexpr size_t REP_LDS = MAXK * 16 + MAXK * 3 * 8 + 64;
constexpr size_t NRM_LDS = MAXK * 16 + 64;
static_assert(OPT_LDS <= 160 * 1024, "LDS budget");

size_t knn_list_bytes(int B) { return (size_t)B * MAXK * LIST_M * sizeof(uint16_t); }

hipError_t configure_optimize_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(optimize_kernel<8, 0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)OPT_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(optimize_kernel<8, 1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)OPT_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(optimize_kernel<8, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)OPT_LDS);
    if (e != hipSuccess) return e;

    e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEC_LDS);
    return e;
}

hipError_t launch_optimize(const float* dec_img, const float* planes, float* p, float* m, float* v,
                           float* loss, const int32_t* loss_batch_per_cloud, uint16_t* knn_lists,
                           unsigned long long* counters, int B, int K, const OptArgs& a, hipStream_t s) {
    if (a.variant == 2)
        hipLaunchKernelGGL((optimize_kernel<8, 2>), dim3(B), dim3(512), OPT_LDS, s, dec_img, planes, p, m, v, loss,
                           loss_batch_per_cloud, knn_lists, counters, K, a);
    else if (a.variant == 1)
        hipLaunchKernelGGL((optimize_kernel<8, 1>), dim3(B), dim3(512), OPT_LDS, s, dec_img, planes, p, m, v, loss,
                           loss_batch_per_cloud, knn_lists, counters, K, a);
    else
        hipLaunchKernelGGL((optimize_kernel<8, 0>), dim3(B), dim3(512), OPT_LDS, s, dec_img, planes, p, m, v, loss,
                           loss_batch_per_cloud, knn_lists, counters, K, a);
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
