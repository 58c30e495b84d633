// Device code of the persistent per-cloud optimiser (optimize_kernel and its decoder tiles): shared by optimize.hip
// (f32 MFMA tiles) and optimize_bf.hip (the split-precision tiles on the bf16 matrix core).
#pragma once
// ConvONet-Opt hot loop for MI355X (gfx950): one workgroup (8 waves, 2 per SIMD, 256 VGPRs each) owns one
// cloud for all Adam steps; one thread owns two points for the kNN / Adam phases, one wave owns a 32-point
// tile (two software-pipelined 16-point sub-tiles) at a time for the decoder phase.
//
//   per step (reference: ConvONet/opt_defense.py:210-228)
//     kNN      exact 5-NN from certified neighbour lists (defense/pn_utils.py:64-83; knn_device.h) + repulsion
//              loss gradient (defense/repulsion_loss.py:43-54); neighbour AND centre terms are accumulated in 64-bit
//              fixed point (LDS atomics) => order independent, bit reproducible.  All waves run this phase first,
//              together: a VALU-only wave next to an MFMA wave on a SIMD gains nothing (no MFMA/VALU overlap on
//              gfx950) and is starved.
//     tiles    decoder forward + input-gradient, tiles pulled from an LDS counter: bilinear gather of the 3
//              channel-last planes (decoder.py:50-57; all three planes in flight - the taps come from HBM / Infinity
//              Cache), the 5-block ResNet MLP on v_mfma_f32_16x16x4_f32 (decoder.py:83-93, layers.py:39-48),
//              BCE-to-threshold derivative (opt_defense.py:213-216), transposed MLP, dc/du from re-gathered taps.
//     Adam     fused update (torch.optim.Adam single-tensor form), moments in registers.
//   Nothing but the plane taps (and the L2-resident neighbour lists) is read from global memory inside the loop.
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
#include "knn_device.h"

namespace ifd {
[[maybe_unused]] constexpr int IFD_TRACE_BASE = 16;      // first trace slot in the device counter buffer (= IFD_N_COUNTERS)
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef float f32x8 __attribute__((ext_vector_type(8)));

// [pcsamp:standalone_tile]
struct Acc2 {
    f32x4 t[2];     // M-tile 0 (channels 4q..4q+3) and M-tile 1 (channels 16+4q..)
};

// out[o][n] (+)= sum_c A[o][c] * in[c][n] with A = W (forward) or W^T (backward); 16 MFMAs, two
// independent accumulator chains (the 40-cycle dependent latency of 16x16x4 is covered by alternating).
// `lo` is the lane's offset into a layer (LaneOff below), made opaque once per tile so that the loop-invariant
// LDS weight loads are not hoisted out of the tile loop by LICM (they would be spilled to scratch).
struct LaneOff {
    int fwd;    // n * S + wperm(4 q)
    int bwd;    // 4 q * S + wperm(n)
    int q4;     // 4 q
};

template <bool TRANSPOSED>
__device__ __forceinline__ Acc2 dense32(const float* __restrict__ wl, const LaneOff& lo, const f32x8& in, Acc2 acc) {
    // forward : A = W[16 mt + n][16 mt' + 4 q + r']  at  (16 mt + n) * S + wperm(16 mt' + r') + wperm(4 q)
    // backward: A = W[16 mt' + 4 q + r'][16 mt + n]  at  (16 mt' + 4 q + r') * S + wperm(16 mt) + wperm(n)
    // (wperm's bit fields are independent: wperm(a + b) = wperm(a) + wperm(b) for the index parts above)
    const float* base = wl + (TRANSPOSED ? lo.bwd : lo.fwd);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int mtp = s >> 2, rp = s & 3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int off = TRANSPOSED ? ((16 * mtp + rp) * W_STRIDE + wperm(16 * mt)) : (16 * mt * W_STRIDE + wperm(16 * mtp + rp));
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
// [pcsamp:tile.relu]
__device__ __forceinline__ f32x8 relu8(const f32x8& v) {
    f32x8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = __int_as_float(max(__float_as_int(v[r]), 0));
    return o;
}

// [pcsamp:standalone_tile]
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
    LaneOff lo = {n * W_STRIDE + wperm(4 * q), 4 * q * W_STRIDE + wperm(n), 4 * q};
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
    part = add_lane_xor32(add_lane_xor16(part));
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
        v = add_lane_xor32(add_lane_xor16(v));
        dx[a] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Two 16-point sub-tiles per wave (the optimiser's decoder tile; helpers shared by decoder_tile3 below).
//   * 4 independent accumulator chains per layer (2 sub-tiles x 2 M-tiles): no dependent-MFMA stalls, and the
//     VALU epilogue of one sub-tile (bias, ReLU, mask) overlaps the other sub-tile's MFMAs;
//   * every A operand (weight) is fetched from LDS once and feeds both sub-tiles: half the LDS reads per MFMA;
//   * ReLU masks are packed 8 bits per layer (one v_alignbit per value going in, v_bfe_i32 + v_bfi per value coming
//     out) and made opaque (asm) so the compiler keeps them as 1 VGPR each.
// The taps are gathered one plane at a time and re-gathered for the backward pass (L1/L2 hits): holding them
// for two sub-tiles would need 192 VGPRs.
// ---------------------------------------------------------------------------------------------
// ReLU masks of the optimiser tile.  One v_alignbyte per value shifts the top byte of the pre-activation (sign bit
// first) into a mask word, four values per word (value r -> byte 3 - r of word r / 4); the words are inverted once, and
// the backward pass applies a mask with ONE v_and_b32_sdwa per value (the sign-extended byte: all ones where the sign
// bit was clear).  "Alive" therefore means sign bit clear: v > 0 or v == +0.0 - torch's threshold_backward uses v > 0, so
// the two differ only for a pre-activation that is exactly +0.0 (forward values are identical; the one structural case,
// a zero bias under an all-dead input, sits under a dead outer mask).  Integer / bit-field / SDWA instructions cost ~6
// SIMD cycles each next to the f32 MFMAs (scripts/valu_rates.hip): round 1's exact-at-+0 form (v_max, v_add -1,
// v_alignbit | v_bfe, v_bfi) was 29 cycles per value, bit masks through v_alignbit | v_bfe + v_and 24, this one 20.
// [pcsamp:tile.mask_pack]
struct Mask8 {
    uint32_t w[2];
};
// v: pre-activations straight out of the MFMAs; rv: relu(v) as the compiler computed it.  The inline asm lists rv as an
// (unused) input so that it is ordered behind the compiler's own first read of v - the compiler pads MFMA -> VALU
// read-after-write hazards for its own instructions, not for inline asm.
template <int BYTE>
__device__ __forceinline__ void put_sign_byte(uint32_t& x, float v, float rv, int c31) {
    if (BYTE == 0) asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(x) : "v"(c31), "v"(v), "v"(rv));
    if (BYTE == 1) asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(x) : "v"(c31), "v"(v), "v"(rv));
    if (BYTE == 2) asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(x) : "v"(c31), "v"(v), "v"(rv));
    if (BYTE == 3) asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(x) : "v"(c31), "v"(v), "v"(rv));
}
__device__ __forceinline__ Mask8 mask_alive_packed(const f32x8& v, const f32x8& rv, int c31) {
    Mask8 m;
#ifdef IFD_EXACT_REP          // the exact-arithmetic build (libifd_exact.so): torch's rule, threshold_backward passes where v > 0 -
                              // a pre-activation of exactly +0.0 is DEAD (the default build below keeps the sign bit: +0.0 alive)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t x = 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) x |= (v[4 * h + r] > 0.f ? 0xffu : 0u) << (8 * (3 - r));
        asm volatile("" : "+v"(x));
        m.w[h] = x;
    }
#elif defined(IFD_MASK_SDWA)          // round 2: one SDWA sign-byte shift per value (+ one s_nop between partial writes of a word) and a v_not
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t x;                                  // byte 3 - r of word h: 0xff where value 4 h + r has its sign bit set
        asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"
            : "=v"(x) : "v"(c31), "v"(v[4 * h + 0]), "v"(rv[4 * h + 0]));
        put_sign_byte<2>(x, v[4 * h + 1], rv[4 * h + 1], c31);
        put_sign_byte<1>(x, v[4 * h + 2], rv[4 * h + 2], c31);
        put_sign_byte<0>(x, v[4 * h + 3], rv[4 * h + 3], c31);
        x = ~x;
        asm volatile("" : "+v"(x));
        m.w[h] = x;
    }
#else
    // v_perm_b32 can replicate the sign bit of either source DWORD into a result byte (selector 11: bit 31 of src0, 9: bit 31
    // of src1; 12: 0x00), so one instruction extracts the sign bytes of TWO values; the two half-filled words have disjoint
    // bytes, and v_xnor = ~(a ^ b) = ~(a | b) merges and inverts them in one go: 3 instructions per 4 values instead of 4
    // SDWA shifts + their 3 hazard nops + a v_not (34 against ~65 SIMD cycles per 8 values, scripts/valu_rates.hip).
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t hi = __builtin_amdgcn_perm(__float_as_uint(v[4 * h + 0]), __float_as_uint(v[4 * h + 1]), 0x0b090c0cu);
        const uint32_t lo = __builtin_amdgcn_perm(__float_as_uint(v[4 * h + 2]), __float_as_uint(v[4 * h + 3]), 0x0c0c0b09u);
        uint32_t x = ~(hi ^ lo);
        asm volatile("" : "+v"(x));
        m.w[h] = x;
    }
#endif
    return m;
}

// relu(v) * 2^-RELU_K for the eight values of a sub-tile and layer: four v_pk_mul_f32 ... clamp (ifd_device.h "packed ReLU").
// The instruction is inline asm (clang folds the clamp into scalar multiplies only), and the compiler pads MFMA -> VALU
// read-after-write hazards for its own instructions, not for inline asm: each statement therefore takes the mask word of its
// M-tile as an (unused) input - the v_perm that made that word is the compiler's own read of the same MFMA result quad.
__device__ __forceinline__ f32x8 relu8s(const f32x8& v, const Mask8& m, unsigned long long sc2) {
    f32x8 o;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#ifdef IFD_RELU_NOASM
        o[2 * h] = fminf(fmaxf(v[2 * h] * RELU_DN, 0.f), 1.f);
        o[2 * h + 1] = fminf(fmaxf(v[2 * h + 1] * RELU_DN, 0.f), 1.f);
#else
        const f32x2 in = {v[2 * h], v[2 * h + 1]};
        f32x2 out;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] clamp" : "=v"(out) : "v"(in), "s"(sc2), "v"(m.w[h >> 1]));
        o[2 * h] = out.x;
        o[2 * h + 1] = out.y;
#endif
    }
    return o;
}
// (the pair of relu and mask of one sub-tile and layer, either way)
__device__ __forceinline__ void relu_and_mask(const f32x8& v, int c31, unsigned long long sc2, f32x8& r, Mask8& m) {
    if (RELU_K != 0) {
        m = mask_alive_packed(v, v, c31);
        r = relu8s(v, m, sc2);
    } else {
        r = relu8(v);
        m = mask_alive_packed(v, r, c31);
    }
}
// d += t * 2^-2K (RELU_K != 0: the transposed fc_1 / fc_0 products carry 2^2K) or d += t: four packed instructions either way
__device__ __forceinline__ void add8_scaled(f32x8& d, const f32x8& t) {
    if (RELU_K == 0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const f32x2 v = f32x2{d[2 * h], d[2 * h + 1]} + f32x2{t[2 * h], t[2 * h + 1]};
            d[2 * h] = v.x; d[2 * h + 1] = v.y;
        }
    } else {
        // (inline asm: left to itself the compiler turns the constant operand into eight scalar v_fma_f32; the inputs come from
        // vector instructions - the mask application - so there is no MFMA hazard to pad here)
        constexpr unsigned int cb = __builtin_bit_cast(unsigned int, RELU_DN2);
        const unsigned long long c2 = ((unsigned long long)cb << 32) | cb;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
#ifdef IFD_RELU_NOASM
            d[2 * h] = __builtin_fmaf(t[2 * h], RELU_DN2, d[2 * h]);
            d[2 * h + 1] = __builtin_fmaf(t[2 * h + 1], RELU_DN2, d[2 * h + 1]);
#else
            f32x2 v = {d[2 * h], d[2 * h + 1]};
            const f32x2 tt = {t[2 * h], t[2 * h + 1]};
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(v) : "v"(tt), "s"(c2));
            d[2 * h] = v.x; d[2 * h + 1] = v.y;
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The two-sub-tile schedule, software-pipelined (32 MFMAs per layer, 16 LDS reads - every A operand feeds both).
//   region k.1 :  16 MFMAs of sub-tile 0, layer k   ||  VALU epilogue (bias/ReLU/mask) of sub-tile 1, layer k-1
//   region k.2 :  16 MFMAs of sub-tile 1, layer k   ||  VALU epilogue of sub-tile 0, layer k  ||  LDS prefetch of
//                 layer k+1's 16 A operands + bias into registers
// so the matrix pipe never waits for an epilogue or an LDS round trip of its own wave.  An in-order wave only
// overlaps what is adjacent in its instruction stream, hence the explicit sched_group_barrier interleave
// (1 MFMA : n VALU : m DS-read) inside every region and a sched_barrier between regions.
// ---------------------------------------------------------------------------------------------
// [pcsamp:tile.wfrag_lds]
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
            const int off = TRANSPOSED ? ((16 * mtp + rp) * W_STRIDE + wperm(16 * mt)) : (16 * mt * W_STRIDE + wperm(16 * mtp + rp));
            f.a[2 * s + mt] = base[off];
        }
    }
    return f;
}

// [pcsamp:tile.mfma]
__device__ __forceinline__ void mfma16(const WFrag& f, const f32x8& in, Acc2& acc) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            acc.t[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[2 * s + mt], in[s], acc.t[mt], 0, 0, 0);
}

// interleave pattern of one region: 4 x { 4 MFMAs, 4 NV VALU, 4 ND DS reads }, then close the region.  (An f32 MFMA and a
// vector instruction exclude each other on the SIMD and every switch between the two kinds costs issue cycles, so they
// alternate in groups, not one by one.)
// -DIFD_EXTRA_VALU=<n>: n dummy vector instructions per software-pipeline region (measurement only: what ONE more vector
// instruction next to the MFMA stream costs in the real kernel - scripts/ab_bench.sh)
#ifndef IFD_EXTRA_VALU
#define IFD_EXTRA_VALU 0
#endif
template <int NV, int ND>
__device__ __forceinline__ void region_end() {
    if (IFD_EXTRA_VALU > 0) {
        int dummy = 0;
#pragma unroll
        for (int k = 0; k < IFD_EXTRA_VALU; ++k) asm volatile("v_add_u32 %0, 1, %0" : "+v"(dummy));
    }
    constexpr int GROUP = 4;         // 2: 418 k, 4 / 8: 412 k, 16: 413 k cycles per step (one by one: 428 k)
#pragma unroll
    for (int i = 0; i < 16 / GROUP; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, GROUP, 0);
        if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, GROUP * NV, 0);
        if (ND > 0) __builtin_amdgcn_sched_group_barrier(0x100, GROUP * ND, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// [pcsamp:tile.acc_add]
// four v_pk_add_f32 instead of eight v_add_f32 (the accumulator quads are register pairs)
__device__ __forceinline__ f32x4 add4_pk(const f32x4& a, const f32x4& b) {
    const f32x2 lo = f32x2{a.x, a.y} + f32x2{b.x, b.y};
    const f32x2 hi = f32x2{a.z, a.w} + f32x2{b.z, b.w};
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ Acc2 acc_add(const Acc2& a, const Acc2& b) {
    Acc2 r;
    r.t[0] = add4_pk(a.t[0], b.t[0]);
    r.t[1] = add4_pk(a.t[1], b.t[1]);
    return r;
}
__device__ __forceinline__ void add8_pk(f32x8& d, const f32x8& t) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const f32x2 v = f32x2{d[2 * h], d[2 * h + 1]} + f32x2{t[2 * h], t[2 * h + 1]};
        d[2 * h] = v.x; d[2 * h + 1] = v.y;
    }
}

__device__ __forceinline__ Acc2 acc_zero() {
    Acc2 r;
    r.t[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    r.t[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    return r;
}
// [pcsamp:tile.mask_apply]
// z & (sign-extended mask byte): written so that hipcc's SDWA peephole folds the byte extraction into the v_and
// (v_and_b32_sdwa ... src0_sel:BYTE_k with sext) - one instruction per value, and the compiler keeps track of the
// MFMA -> VALU hazard of z itself.
template <int BYTE>
__device__ __forceinline__ float keep_alive(float z, uint32_t word) {
    const int keep = (int)(int8_t)(word >> (8 * BYTE));
    return __uint_as_float(__float_as_uint(z) & (uint32_t)keep);
}
__device__ __forceinline__ f32x8 masked(const f32x8& z, const Mask8& m) {
    f32x8 v;
    v[0] = keep_alive<3>(z[0], m.w[0]); v[1] = keep_alive<2>(z[1], m.w[0]);
    v[2] = keep_alive<1>(z[2], m.w[0]); v[3] = keep_alive<0>(z[3], m.w[0]);
    v[4] = keep_alive<3>(z[4], m.w[1]); v[5] = keep_alive<2>(z[5], m.w[1]);
    v[6] = keep_alive<1>(z[6], m.w[1]); v[7] = keep_alive<0>(z[7], m.w[1]);
    return v;
}
__device__ __forceinline__ f32x8 masked(const Acc2& z, const Mask8& m) { return masked(flat(z), m); }

// [pcsamp:adam.pix]
// ---- sampling coordinates, once per point and step ---------------------------------------------------------------
// The owner thread of a point turns its coordinates into the three pixel coordinates of grid_sample (pixel_coord above:
// normalize_coordinate's divide / clamp, align_corners un-normalisation, border clip) when it writes the point - at
// start-up and in the Adam phase - and keeps them in LDS (PIX).  A coordinate that normalize_coordinate clamped carries
// a minus sign (its plane gradient is zero; -0.0 for the lower clamp).  In a decoder tile four lanes share a point, so
// there the same arithmetic costs a whole wave instruction per 16 points; in the Adam phase one per 64.
__device__ __forceinline__ f32x4 pix_encode(float x0, float x1, float x2, const DecConst& dc) {
    const float xs[3] = {x0, x1, x2};
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float pix, live;
        pixel_coord(xs[a], dc, pix, live);
        o[a] = live != 0.f ? pix : __uint_as_float(__float_as_uint(pix) | 0x80000000u);
    }
    return o;
}

// [pcsamp:tile.geometry]
struct SubGeo {                      // per-point sampling geometry of one sub-tile lane
    float w0[3], w1[3], lk[3];       // bilinear weights per axis; d pix / d x (0 for a clamped coordinate)
    int cell[3];
};

__device__ __forceinline__ void sub_geometry(SubGeo& g, const f32x4 pp, float ksc) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ap = fabsf(pp[a]);
        const int ci = min((int)ap, RES - 2);       // pix >= 0: truncation is floor
        const float cf = (float)ci;
        g.cell[a] = ci;
        g.w1[a] = ap - cf;                           // weight of the east / south tap
        g.w0[a] = (cf + 1.f) - ap;                   // weight of the west / north tap
        g.lk[a] = (int)__float_as_uint(pp[a]) < 0 ? 0.f : ksc;
    }
}

// [pcsamp:tile.setup]
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// -DIFD_TRACE -DIFD_TRACE2=<n>: time stamps INSIDE the n-th decoder tile a wave runs in the traced step (scripts/tile_trace.py):
// one stamp per software-pipeline region of the MLP and per section around it, 128 slots per wave behind the status words
// of the counter buffer.  A stamp is an s_memtime + s_waitcnt lgkmcnt(0) + one store by lane 0 - it drains the wave's LDS
// prefetches at the region boundary, so a traced tile runs a few per cent slower than an untraced one.
#ifdef IFD_TRACE2
#define T2(slot) do { if (tr != nullptr) { const unsigned long long t2_ = __builtin_readcyclecounter(); if (lane == 0) tr[slot] = t2_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define T2(slot)
#endif

template <int MODE>
__device__ __forceinline__ void decoder_tile3(const float* __restrict__ W, __amdgpu_buffer_rsrc_t planes,
                                              const f32x4 ppa, const f32x4 ppb, float xqa, float xqb, int lane,
                                              const DecConst dc, float thr, float inv_lb, bool want_loss,
                                              float (&bce)[2], float (&dx)[2][3], [[maybe_unused]] unsigned long long* tr = nullptr) {
    constexpr int TV = 1;
    T2(0);                                       // tile entered
    const int n = lane & 15, q = lane >> 4;
    LaneOff lo = {n * W_STRIDE + wperm(4 * q), 4 * q * W_STRIDE + wperm(n), 4 * q};
    asm volatile("" : "+v"(lo.fwd), "+v"(lo.bwd), "+v"(lo.q4));
    constexpr int AX0[3] = {0, 0, 1}, AX1[3] = {2, 1, 2};   // xz, xy, yz (common.py:243-248)
    // d pix / d x = (RES - 1) / 2 * 2 / sdiv as one constant (the reference multiplies and divides in sequence; the
    // difference is a rounding in the last place of a gradient term)
    const float ksc = ((0.5f * (float)(RES - 1)) * 2.f) / dc.sdiv;
    SubGeo geo[2];
    sub_geometry(geo[0], ppa, ksc);
    sub_geometry(geo[1], ppb, ksc);
    const float* Wd = W + DEC_OFF_W;
    // bias rows through their own base register: the image is > 64 KB, past the reach of a DS immediate offset
    int boff = DEC_OFF_BIAS + lo.q4;             // (opaque as an integer: an opaque POINTER loses its LDS address space)
    asm volatile("" : "+v"(boff));
    const float* Bq = W + boff;
    auto bias = [&](int layer) {
        Acc2 b;
        b.t[0] = *reinterpret_cast<const f32x4*>(Bq + layer * 32);
        b.t[1] = *reinterpret_cast<const f32x4*>(Bq + layer * 32 + 16);
        return b;
    };

    // ---- gather + forward bilinear sample ----------------------------------------------------------------------
    // The taps come from HBM / Infinity Cache (L2 hit rate 37 %): one round trip per plane would cost ~1.5 us each, so
    // the loads of all planes are in flight at once (64 tap registers per plane).  Buffer loads: the cloud's plane
    // base sits in the SGPR resource, the lane supplies a 32-bit byte offset - no 64-bit address arithmetic.
    f32x8 c[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 8; ++r) c[t][r] = 0.f;
    const int q16 = 16 * q;
// [pcsamp:tile.tap_loads]
    auto load_taps = [&](int P, f32x4 (&tap)[2][4][2], bool opaque) {
        const int a0 = AX0[P], a1 = AX1[P];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int off = (geo[t].cell[a1] * RES + geo[t].cell[a0]) * (CH * 4) + (P * PLANE_FLOATS * 4 + q16);
            if (opaque) asm volatile("" : "+v"(off));      // backward re-gather: do not CSE with (and keep alive since) the forward one
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const u32x4v t0 = __builtin_amdgcn_raw_buffer_load_b128(planes, off + 64 * mt, 0, 0);
                const u32x4v t1 = __builtin_amdgcn_raw_buffer_load_b128(planes, off + CH * 4 + 64 * mt, 0, 0);
                const u32x4v t2 = __builtin_amdgcn_raw_buffer_load_b128(planes, off + RES * CH * 4 + 64 * mt, 0, 0);
                const u32x4v t3 = __builtin_amdgcn_raw_buffer_load_b128(planes, off + RES * CH * 4 + CH * 4 + 64 * mt, 0, 0);
                tap[t][0][mt] = __builtin_bit_cast(f32x4, t0);
                tap[t][1][mt] = __builtin_bit_cast(f32x4, t1);
                tap[t][2][mt] = __builtin_bit_cast(f32x4, t2);
                tap[t][3][mt] = __builtin_bit_cast(f32x4, t3);
            }
        }
    };
// [pcsamp:tile.sample_fwd]
    auto sample_fwd = [&](int P, const f32x4 (&tap)[2][4][2]) {
        const int a0 = AX0[P], a1 = AX1[P];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const SubGeo& g = geo[t];
            const float wnw = g.w0[a0] * g.w0[a1], wne = g.w1[a0] * g.w0[a1], wsw = g.w0[a0] * g.w1[a1],
                        wse = g.w1[a0] * g.w1[a1];
#ifdef IFD_SAMPLE_SCALAR
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {      // (one fma chain over all 12 taps measured +2.6 %)
                    float s = tap[t][0][mt][j] * wnw;
                    s = fmaf(tap[t][1][mt][j], wne, s);
                    s = fmaf(tap[t][2][mt][j], wsw, s);
                    s = fmaf(tap[t][3][mt][j], wse, s);
                    c[t][4 * mt + j] += s;
                }
#else
            // packed along the channels like the backward sampling: the taps arrive as aligned register pairs, the weight is
            // broadcast by op_sel - 5 v_pk instructions per channel pair instead of 10 scalar ones, same operation order
            const f32x2 wnw2 = {wnw, wnw}, wne2 = {wne, wne}, wsw2 = {wsw, wsw}, wse2 = {wse, wse};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 s2 = f32x2{tap[t][0][mt][2 * h], tap[t][0][mt][2 * h + 1]} * wnw2;
                    s2 = __builtin_elementwise_fma(f32x2{tap[t][1][mt][2 * h], tap[t][1][mt][2 * h + 1]}, wne2, s2);
                    s2 = __builtin_elementwise_fma(f32x2{tap[t][2][mt][2 * h], tap[t][2][mt][2 * h + 1]}, wsw2, s2);
                    s2 = __builtin_elementwise_fma(f32x2{tap[t][3][mt][2 * h], tap[t][3][mt][2 * h + 1]}, wse2, s2);
                    const f32x2 cc = P == 0 ? s2 : f32x2{c[t][4 * mt + 2 * h], c[t][4 * mt + 2 * h + 1]} + s2;   // (c starts at 0)
                    c[t][4 * mt + 2 * h] = cc.x;
                    c[t][4 * mt + 2 * h + 1] = cc.y;
                }
#endif
        }
    };
    {
// [pcsamp:tile.gather_seq]
        f32x4 tap0[2][4][2], tap1[2][4][2], tap2[2][4][2];
        load_taps(0, tap0, false);
        load_taps(1, tap1, false);
        load_taps(2, tap2, false);               // all three planes in flight (192 registers; nothing else is live yet)
        __builtin_amdgcn_sched_barrier(0);
        T2(1);                                   // 48 tap loads issued
#ifdef IFD_TRACE2
        if (tr != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        T2(2);                                   // taps here
#endif
        sample_fwd(0, tap0);
        __builtin_amdgcn_sched_barrier(0);
        sample_fwd(1, tap1);
        __builtin_amdgcn_sched_barrier(0);
        sample_fwd(2, tap2);
        __builtin_amdgcn_sched_barrier(0);
        T2(3);                                   // forward sampling done
    }

// [pcsamp:tile.fc_p]
    // Wave priority: 1 while the wave streams MFMAs, 0 in its VALU / memory sections (and in the kNN / Adam phases).  An
    // f32 MFMA and a vector instruction cannot overlap on a SIMD and every switch between the two costs issue cycles
    // (scripts/mfma_valu_inwave.hip); with the MFMA wave preferred its stream runs back to back and the partner's vector
    // work fills the gaps where it stalls, instead of the two alternating instruction by instruction (412 k -> 388 k
    // cycles per step; the inverse assignment measured 425 k).
    asm volatile("s_setprio 1");
    // ---- fc_p on the matrix pipe: n_0 = [Wp | bp + bc_0] [x; 1]  (K = 4: one MFMA per M-tile and sub-tile) ----------
    // The lane's B operand is component q of its point (X.w = 1 carries the bias), its A operand row 16 mt + n, column q of
    // the [32][4] fc_p block.  (The biases of fc_c[i] are folded into the bias of the layer before: api.cpp build_dec_image.)
    Acc2 net[2];
    {
        const float ap0 = W[DEC_OFF_WP + n * 4 + q], ap1 = W[DEC_OFF_WP + (16 + n) * 4 + q];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float xq = t ? xqb : xqa;
            net[t].t[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap0, xq, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            net[t].t[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap1, xq, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
    }
    WFrag A = load_wfrag<false>(Wd, lo);         // weights of the first layer (after the gather: its registers are free now)
    __builtin_amdgcn_sched_barrier(0);
    T2(4);                                       // fc_p issued, first weights requested

// [pcsamp:tile.fwd_mlp]
    // ---- forward MLP, software pipelined ---------------------------------------------------------------------
    Mask8 mask_a[2][NBLK], mask_h[2][NBLK];
    f32x8 wout;
    float bout = 0.f, ilb = 0.f;
    int c31 = 31;                                  // shift count of the sign-byte extraction (an SDWA operand must be a register)
    asm volatile("" : "+v"(c31));
    // 2^-RELU_K in both halves of an SGPR pair (the packed ReLU's second operand)
    const unsigned long long sc2 = ((unsigned long long)__builtin_bit_cast(unsigned int, RELU_DN) << 32) | __builtin_bit_cast(unsigned int, RELU_DN);
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const float* Wl = Wd + 3 * i * W_LAYER;
        // fc_c: a = n + fc_c(c)   (bias already inside n)
        Acc2 a0 = net[0], a1 = net[1];
        mfma16(A, c[0], a0);                                                  // R1
        region_end<0, 0>();
        T2(5 + 6 * i + 0);
        const WFrag A0 = load_wfrag<false>(Wl + W_LAYER, lo);                 // R2: prefetch fc_0
        const Acc2 B0 = bias(3 * i + 1);
        mfma16(A, c[1], a1);
        const f32x8 af0 = flat(a0);
        f32x8 ra0;
        relu_and_mask(af0, c31, sc2, ra0, mask_a[0][i]);
        region_end<TV, 1>();
        T2(5 + 6 * i + 1);
        Acc2 h0 = B0;                                                          // R3
        mfma16(A0, ra0, h0);
        const f32x8 af1 = flat(a1);
        f32x8 ra1;
        relu_and_mask(af1, c31, sc2, ra1, mask_a[1][i]);
        region_end<TV, 0>();
        T2(5 + 6 * i + 2);
        const WFrag A1 = load_wfrag<false>(Wl + 2 * W_LAYER, lo);             // R4: prefetch fc_1
        const Acc2 B1 = bias(3 * i + 2);
        Acc2 h1 = B0;
        mfma16(A0, ra1, h1);
        const f32x8 hf0 = flat(h0);
        f32x8 rh0;
        relu_and_mask(hf0, c31, sc2, rh0, mask_h[0][i]);
        region_end<TV, 1>();
        T2(5 + 6 * i + 3);
        Acc2 o0 = acc_add(B1, a0);                                             // R5
        mfma16(A1, rh0, o0);
        const f32x8 hf1 = flat(h1);
        f32x8 rh1;
        relu_and_mask(hf1, c31, sc2, rh1, mask_h[1][i]);
        region_end<TV, 0>();
        T2(5 + 6 * i + 4);
        if (i + 1 < NBLK) {                                                    // R6: prefetch next fc_c / first fc_1^T
            A = load_wfrag<false>(Wl + 3 * W_LAYER, lo);
        } else {
            A = load_wfrag<true>(Wl + 2 * W_LAYER, lo);
            // fc_out's weights and bias ride in with the last region's LDS reads: read after it they would be an LDS round
            // trip at the head of the forward -> backward chain (logit, sigmoid, seed), where this wave has no MFMA to issue
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WOUT + lo.q4);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(W + DEC_OFF_WOUT + 16 + lo.q4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { wout[r] = t0[r]; wout[4 + r] = t1[r]; }
            const f32x2 bo = *reinterpret_cast<const f32x2*>(W + DEC_OFF_BOUT);        // {fc_out's bias, 1 / B of the cloud (optimize_kernel)}
            bout = bo.x;
            ilb = bo.y;
        }
        Acc2 o1 = acc_add(B1, a1);
        mfma16(A1, rh1, o1);
        net[0] = o0;
        net[1] = o1;
        region_end<0, 1>();
        T2(5 + 6 * i + 5);
    }
// [pcsamp:tile.logit]
    f32x8 dn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x8 nf = flat(net[t]);
        f32x8 rn;
        Mask8 mask_n;
        relu_and_mask(nf, c31, sc2, rn, mask_n);
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) part = fmaf(wout[r], rn[r], part);
        part = add_lane_xor32(add_lane_xor16(part));
        const float logit = part + bout;
        float dl;
        if (MODE == MODE_OPT) {
            // BCE-with-logits against the threshold (opt_defense.py:213-216): d/dlogit = (sigmoid - thr) / B.  The
            // loss value itself is only reported for the last step.
#ifdef IFD_FAST_SIGMOID_EXP
            const float e = __expf(-fabsf(logit));
#else
            const float e = expf(-fabsf(logit));       // (v_exp_f32 directly: 379.2 against 380.0 k cycles per step - not taken)
#endif
            const float rc = __builtin_amdgcn_rcpf(1.f + e);
            const float sig = logit >= 0.f ? rc : e * rc;
            dl = (sig - thr) * ilb;
            bce[t] = want_loss ? fmaxf(logit, 0.f) - thr * logit + log1pf(e) : 0.f;
        } else {
            bce[t] = logit;
            dl = 1.f;
        }
        {
            f32x8 dw;
            const float dls = RELU_K != 0 ? dl * RELU_DN : dl;      // (fc_out sits in the image times 2^K: same products)
#pragma unroll
            for (int r = 0; r < 8; ++r) dw[r] = dls * wout[r];
            dn[t] = masked(dw, mask_n);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    T2(35);                                      // logits, loss derivative, seed of the backward pass

// [pcsamp:tile.bwd_mlp]
    // ---- backward, software pipelined (A holds fc_1[4]^T) ----------------------------------------------------------
    Acc2 dcc[2] = {acc_zero(), acc_zero()};
#pragma unroll
    for (int i = NBLK - 1; i >= 0; --i) {
        const float* Wl = Wd + 3 * i * W_LAYER;
        Acc2 z0 = acc_zero();                                                  // R1: fc_1^T dn (sub-tile 0)
        mfma16(A, dn[0], z0);
        region_end<0, 0>();
        T2(36 + 6 * (NBLK - 1 - i) + 0);
        const WFrag A0 = load_wfrag<true>(Wl + W_LAYER, lo);                  // R2: prefetch fc_0^T
        Acc2 z1 = acc_zero();
        mfma16(A, dn[1], z1);
        const f32x8 dh0 = masked(z0, mask_h[0][i]);
        region_end<1, 1>();
        T2(36 + 6 * (NBLK - 1 - i) + 1);
        Acc2 y0 = acc_zero();                                                  // R3: fc_0^T dh (sub-tile 0)
        mfma16(A0, dh0, y0);
        const f32x8 dh1 = masked(z1, mask_h[1][i]);
        region_end<1, 0>();
        T2(36 + 6 * (NBLK - 1 - i) + 2);
        const WFrag Ac = load_wfrag<true>(Wl, lo);                            // R4: prefetch fc_c^T
        Acc2 y1 = acc_zero();
        mfma16(A0, dh1, y1);
        {
            const f32x8 t = masked(y0, mask_a[0][i]);
            add8_scaled(dn[0], t);                                             // delta a_i
        }
        region_end<1, 1>();
        T2(36 + 6 * (NBLK - 1 - i) + 3);
        mfma16(Ac, dn[0], dcc[0]);                                             // R5: dc += fc_c^T da (sub-tile 0)
        {
            const f32x8 t = masked(y1, mask_a[1][i]);
            add8_scaled(dn[1], t);
        }
        region_end<1, 0>();
        T2(36 + 6 * (NBLK - 1 - i) + 4);
        if (i > 0) A = load_wfrag<true>(Wl - W_LAYER, lo);                    // R6: prefetch fc_1[i-1]^T
        mfma16(Ac, dn[1], dcc[1]);
        region_end<0, 1>();
        T2(36 + 6 * (NBLK - 1 - i) + 5);
    }
    asm volatile("s_setprio 0");
// [pcsamp:tile.fc_p_bwd]
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
    // packed along the channels (even / odd partial sums, added at the end): register pairs as loaded, no shuffling
// [pcsamp:tile.sample_bwd]
    auto sample_bwd = [&](int P, const f32x4 (&tap)[2][4][2]) {
        const int a0 = AX0[P], a1 = AX1[P];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const SubGeo& gg = geo[t];
            f32x2 pnw = {0.f, 0.f}, pne = {0.f, 0.f}, psw = {0.f, 0.f}, pse = {0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2 d = {dcf[t][4 * mt + 2 * h], dcf[t][4 * mt + 2 * h + 1]};
                    pnw = __builtin_elementwise_fma(f32x2{tap[t][0][mt][2 * h], tap[t][0][mt][2 * h + 1]}, d, pnw);
                    pne = __builtin_elementwise_fma(f32x2{tap[t][1][mt][2 * h], tap[t][1][mt][2 * h + 1]}, d, pne);
                    psw = __builtin_elementwise_fma(f32x2{tap[t][2][mt][2 * h], tap[t][2][mt][2 * h + 1]}, d, psw);
                    pse = __builtin_elementwise_fma(f32x2{tap[t][3][mt][2 * h], tap[t][3][mt][2 * h + 1]}, d, pse);
                }
            // the bilinear-weight derivative is linear: apply it to the even / odd halves, then one horizontal add each
            // (as inline asm: the SLP vectoriser would otherwise re-pair the halves across accumulators through v_mov's)
            const f32x2 w0y = {gg.w0[a1], gg.w0[a1]}, w1y = {gg.w1[a1], gg.w1[a1]};
            const f32x2 w0x = {gg.w0[a0], gg.w0[a0]}, w1x = {gg.w1[a0], gg.w1[a0]};
            const f32x2 pgx = (pne - pnw) * w0y + (pse - psw) * w1y;
            const f32x2 pgy = (psw - pnw) * w0x + (pse - pne) * w1x;
            float gix, giy;
            asm("v_add_f32 %0, %1, %2" : "=v"(gix) : "v"(pgx.x), "v"(pgx.y));
            asm("v_add_f32 %0, %1, %2" : "=v"(giy) : "v"(pgy.x), "v"(pgy.y));
            g[t][a0] = fmaf(gg.lk[a0], gix, g[t][a0]);
            g[t][a1] = fmaf(gg.lk[a1], giy, g[t][a1]);
        }
    };
    {
// [pcsamp:tile.bwd_gather_seq]
        // two planes in flight (three spill here, and starting the re-gather under the last MLP block does too:
        // both measured slower)
        f32x4 btap0[2][4][2], tap1[2][4][2], tap2[2][4][2];
        T2(66);                                  // fc_p backward done
        load_taps(0, btap0, true);
        load_taps(1, tap1, true);
        __builtin_amdgcn_sched_barrier(0);
        T2(67);                                  // re-gather of two planes issued
#ifdef IFD_TRACE2
        if (tr != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        T2(68);                                  // ... here
#endif
        sample_bwd(0, btap0);
        __builtin_amdgcn_sched_barrier(0);
        load_taps(2, tap2, true);
        __builtin_amdgcn_sched_barrier(0);
        sample_bwd(1, tap1);
        __builtin_amdgcn_sched_barrier(0);
        sample_bwd(2, tap2);
        __builtin_amdgcn_sched_barrier(0);
        T2(69);                                  // backward sampling done
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            dx[t][a] = add_lane_xor32(add_lane_xor16(g[t][a]));
        }
    T2(70);                                      // tile left
    asm volatile("s_setprio 0");
}

#include "tile_bf.h"

// [pcsamp:kernel.prologue]
__device__ __forceinline__ void load_dec_image(float* __restrict__ W, const float* __restrict__ img, int n_floats = DEC_FLOATS) {
    for (int i = threadIdx.x * 4; i < n_floats; i += blockDim.x * 4)
        *reinterpret_cast<f32x4*>(W + i) = *reinterpret_cast<const f32x4*>(img + i);
}

// ---------------------------------------------------------------------------------------------
// the persistent per-cloud optimiser
// ---------------------------------------------------------------------------------------------
// NW waves per workgroup (8: 2 per SIMD / 256 VGPRs, 12: 3 per SIMD / 168 VGPRs); threads [0,512) own two points
// each (kNN + Adam duty), every wave pulls 16-point decoder tiles from an LDS counter.
//
// S > 1: "split" clouds - S workgroups (S CUs) work on ONE cloud, so that a launch with fewer clouds than CUs (the last
// partial round of a file, a small shard of a file spread over many GPUs) still fills the chip.  Member m of a cloud owns
// 1024 / S points (CoopWs in knn_device.h): the threads [0, 512 / S) keep their neighbour lists, repulsion terms and Adam
// state, and the member's waves run those points' decoder tiles.  Per step the members exchange: the neighbour terms of
// points they do not own (integer atomics into the owner's global accumulators during the kNN phase, collected by the owner
// in its Adam phase - arrival counter bar_knn, waited for under the decoder tiles), and the new positions + certificate
// maxima (arrival counter bar_step, the one exposed cross-CU wait of a step).  Every sum that crosses members is an integer
// sum or a maximum, the last step's loss reduction and the final normalisation are done on the whole cloud in the
// one-workgroup order: the results are bit-identical to the S = 1 kernel.
//
// PREC > 0: the decoder tiles on the bf16 matrix core with f32-exact operand splits (tile_bf.h; 1 = six terms, 2 = three).  The
// parameter image is then the 94,736-byte piece image (ifd_device.h) and the Adam moments sleep in global memory (mv_ws, 24 KB per
// workgroup, L2-resident) instead of LDS: everything else - neighbour lists, repulsion, Adam, split clouds - is the same code.
template <int NW, int S, int PREC = 0>
__global__ __launch_bounds__(NW * 64, NW / 4) void optimize_kernel(
    const float* __restrict__ dec_img, const float* __restrict__ planes, float* __restrict__ p,
    float* __restrict__ m_io, float* __restrict__ v_io, float* __restrict__ loss_out,
    const int32_t* __restrict__ loss_batch_per_cloud, uint16_t* knn_lists,
    unsigned long long* __restrict__ counters, const float* __restrict__ adam_tab, int K, OptArgs A,
    CoopWs* __restrict__ coop, int n_clouds, f32x4* __restrict__ mv_ws = nullptr) {
    // Owner threads.  One workgroup per cloud: 512 threads x two points (pa = t, pb = t + 512).  Split clouds: ONE point per
    // thread, so that the neighbour / Adam duty of the member's 1024 / S points is spread over as many waves as possible
    // (S = 2: all eight, S = 4: four) - with two points per thread only 8 / S waves had that duty, each as long as in the
    // unsplit kernel, and the phase did not shrink with S.  (The second chain of the interleaved kNN code then runs on a dummy
    // point, lane by lane.  Compiling it OUT - half the key networks - measured 53.4 / 36.3 ms against 47.7 / 30.3 at S = 2 /
    // 4, also with the one point's entries spread over two networks for the same instruction-level parallelism.)
    constexpr int PTS = MAXK / S;                     // points of this member
    constexpr int OT = S == 1 ? OPT_THREADS : (PTS < OPT_THREADS ? PTS : OPT_THREADS);      // owner threads
    constexpr int OW = OT / 64;                       // ... and owner waves
    static_assert(S * OW <= MAX_WAVES && S * OW <= MAX_COOP_WAVES, "one maxima slot per owner wave of the cloud");
    K = K < MAXK ? K : MAXK;                          // (the launcher guarantees it; spelled out so that pb < K folds to false)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int IMG_FLOATS = PREC == 0 ? DEC_FLOATS : BF_IMG_BYTES / 4;
    float* W = smem;                                                 // decoder parameter image
    f32x4* G = reinterpret_cast<f32x4*>(smem + IMG_FLOATS);          // occupancy gradient (+ BCE term in .w)
    f32x4* X = G + MAXK;                                             // current points (x, y, z, 1); X[MAXK] = far-away dummy
    f32x4* PIX = X + MAXK + 1;                                       // their sampling coordinates (pix_encode)
    const RepAcc F = {reinterpret_cast<long long*>(PIX + MAXK),      // fixed-point repulsion-gradient scatter (knn_device.h)
                      reinterpret_cast<int*>(reinterpret_cast<long long*>(PIX + MAXK) + MAXK)};
    float* scratch = reinterpret_cast<float*>(F.z + MAXK);           // 128 floats
    // [3][OPT_THREADS] Adam moments of the owner threads: LDS, or (PREC > 0: the piece image takes their place) this workgroup's
    // block of the launch's moment workspace
    f32x4* MV = PREC == 0 ? reinterpret_cast<f32x4*>(scratch + 128) : mv_ws + (size_t)blockIdx.x * (3 * OPT_THREADS);

    // split clouds: the S members of a cloud sit 8 workgroups apart - consecutive workgroups go to the 8 XCDs in turn, so
    // the members share an L2 (an optimisation only: every exchange is agent-scope)
    const int cloud = S == 1 ? (int)blockIdx.x : (int)(((blockIdx.x >> 3) / S) * 8 + (blockIdx.x & 7));
    const int member = S == 1 ? 0 : (int)((blockIdx.x >> 3) % S);
    if (S > 1 && cloud >= n_clouds) return;
    CoopWs* const cws = S > 1 ? coop + cloud : nullptr;
    const CoopView cv = {cws, member, 0};
    unsigned long long* const status = counters;      // sticky status words at STATUS_OVERFLOW / STATUS_TIMEOUT (ifd_internal.h)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pl = planes + (size_t)cloud * (A.planes_shared ? 0 : CLOUD_PLANE_FLOATS);
    float* pc = p + (size_t)cloud * K * 3;
    const bool owner = tid < OT;
    // the point(s) of owner thread t: member m of a split cloud holds {m PTS/2 + u, 512 + m PTS/2 + u : u < PTS/2} (coop_owns)
    auto point_a = [&](int t) {
        return S == 1 ? t : (t < PTS / 2 ? member * (PTS / 2) + t : OPT_THREADS + member * (PTS / 2) + (t - PTS / 2));
    };
    const int pa = owner ? point_a(tid) : MAXK, pb = (S == 1 && owner) ? tid + OPT_THREADS : MAXK;
    constexpr int NT = 32 / S, NTH = 16 / S;        // 32-point decoder tiles of this workgroup: NTH in each half of the cloud
    auto tile_base = [&](int t) { return (t / NTH) * OPT_THREADS + member * (PTS / 2) + (t % NTH) * 32; };

    const int wave_u = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);          // scalar: survives the tile phase for free
    auto hw_tid = [&]() __attribute__((always_inline)) {
        unsigned int ones;                                   // (volatile: the mbcnt pair is otherwise loop-invariant - hoisted and spilled)
        asm volatile("s_mov_b32 %0, -1" : "=s"(ones));
        int t = wave_u * 64 + (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        asm volatile("" : "+v"(t));
        return t;
    };
    const unsigned long long t_begin = __builtin_readcyclecounter();     // shader clock (s_memtime)
    load_dec_image(W, dec_img, IMG_FLOATS);
    __syncthreads();                                     // (the image's padding words are written: one of them is re-used below)
    // Per-thread state of the kNN / Adam phases: the neighbour-list certificates are parked in scratch while the decoder
    // tiles run (knn_device.h "Parking"), the Adam moments in LDS (MV) - the tile phase owns the whole register file.
    f32x4 park[PARK_SLOTS];
    {
        AdamState ast;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pt = q ? pb : pa;
#pragma unroll
            for (int a = 0; a < 3; ++a) ast.mm[3 * q + a] = ast.vv[3 * q + a] = 0.f;
            if (pt < K) {
                if (S == 1) X[pt] = f32x4{pc[3 * pt], pc[3 * pt + 1], pc[3 * pt + 2], 1.f};
                PIX[pt] = pix_encode(pc[3 * pt], pc[3 * pt + 1], pc[3 * pt + 2], A.dc);
                if (A.t0 > 0 && m_io != nullptr) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        ast.mm[3 * q + a] = m_io[((size_t)cloud * K + pt) * 3 + a];
                        ast.vv[3 * q + a] = v_io[((size_t)cloud * K + pt) * 3 + a];
                    }
                }
            }
        }
        const KnnPt k0 = {0, -1, 0.f, 0.f, 2.0f, 7.0f, f32x4{0.f, 0.f, 0.f, 0.f}, 0.f, 0.f, false, false};   // ~10 front / ~35 total hits on a flat patch
        const int z = opaque_zero();
        if (owner) store_adam(MV, tid, ast);
        if (S > 1)       // every member holds the whole cloud
            for (int i = tid; i < K; i += NW * 64) X[i] = f32x4{pc[3 * i], pc[3 * i + 1], pc[3 * i + 2], 1.f};
        park_knnpt(park, z, PARK_KNN, k0);
        park_knnpt(park, z, PARK_KNN + 4, k0);
    }
    for (int i = tid; i < MAXK; i += NW * 64) { F.xy[i] = 0; F.z[i] = 0; }
    if (tid == 0) X[MAXK] = f32x4{1e18f, 1e18f, 1e18f, 1.f};

    const DecConst dc = A.dc;
    const __amdgpu_buffer_rsrc_t plr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pl), 0, CLOUD_PLANE_FLOATS * 4, 0x00020000);
    const RepConst rc = {A.rep_radius, A.rep_h, A.rep_eps};
    const int loss_batch = loss_batch_per_cloud ? loss_batch_per_cloud[cloud] : A.loss_batch;
    const float inv_lb = 1.0f / (float)loss_batch;
    // The decoder tiles take 1 / B from a spare word of the parameter image in LDS, next to fc_out's bias (one ds_read_b64 for
    // both): as a kernel-lifetime register value it was spilled across the 256-register tiles and reloaded from scratch at the
    // forward -> backward turn-around of EVERY tile, in front of the loss derivative the whole backward pass waits for.
    if (tid == 0) W[(PREC == 0 ? DEC_OFF_BOUT : BF_OFF_BOUT / 4) + 1] = inv_lb;
    // (uniform: kept in a scalar register - as a vector value the compiler broadcast it into a register PAIR for the packed fma of
    // the Adam phase, kept the pair alive across the whole step loop and spilled it: a scratch round trip in the middle of every
    // Adam phase, `profiles/r04_*`)
    const float rep_scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(A.rep_weight / ((float)loss_batch * (float)K * 5.f))));
    const bool use_rep = A.rep_weight > 0.f;
    uint16_t* cloud_lists = knn_lists + (size_t)cloud * MAXK * LIST_M;   // certified neighbour lists, global (L2-resident)
    float* dmaxbuf = scratch + 32;                                   // [2][MAX_WAVES] per-wave max |x - x0| (next step)
    float* movebuf = scratch + 64;                                   // [2][MAX_WAVES] per-wave max single-step move
    volatile int* rebuild_flag = reinterpret_cast<volatile int*>(scratch + 28);   // [2] by step parity
    int* tile_ctr = reinterpret_cast<int*>(scratch + 30);            // next decoder tile of this step
    int* knn_done = reinterpret_cast<int*>(scratch + 31);            // split clouds: owner waves of this member past their kNN phase
    int* coop_abort = reinterpret_cast<int*>(scratch + 27);          // split clouds: a cross-CU wait of this member gave up (coop_wait)
    unsigned int* lcnt = reinterpret_cast<unsigned int*>(scratch + 96);           // [CN_COUNT] event counters
    const KnnShared ksh = {dmaxbuf, movebuf, rebuild_flag};

    if (tid < 2) rebuild_flag[tid] = 0;
    if (tid == 0) { *tile_ctr = 0; *knn_done = 0; *coop_abort = 0; }
    if (tid < CN_COUNT) lcnt[tid] = 0u;
    if (tid < 2 * MAX_WAVES) {
        dmaxbuf[tid] = 0.f;
        movebuf[tid] = (tid & (MAX_WAVES - 1)) < S * OW ? 2.f * A.lr : 0.f;      // (slot = owner wave of the CLOUD: member * OW + wave)
    }
    __syncthreads();
#ifdef IFD_PROF
    unsigned long long* lprof = reinterpret_cast<unsigned long long*>(MV + 3 * OPT_THREADS);   // [NW][PC_COUNT] behind MV
    if (tid < NW * PC_COUNT) lprof[tid] = 0ull;
    __syncthreads();
#endif

#ifndef IFD_CARRY_KNNPT
#define IFD_CARRY_KNNPT 1             // S = 1: the certificates stay in registers from the Adam phase into the next step's kNN phase
#endif
    constexpr bool CARRY = S == 1 && IFD_CARRY_KNNPT != 0;
    KnnPt ka_c = {0, -1, 0.f, 0.f, 2.0f, 7.0f, f32x4{0.f, 0.f, 0.f, 0.f}, 0.f, 0.f, false, false}, kb_c = ka_c;
// [pcsamp:step.head]
    for (int step = 0; step < A.steps; ++step) {
        const bool last = step == A.steps - 1;
        const bool want_loss = last && loss_out != nullptr;
        // Everything a phase derives from the thread index (LDS addresses of candidates, list pointers, tile lanes) is
        // re-derived from an opaque copy once per step: left alone, LICM hoists those cheap invariants out of the
        // 501-step loop and they end up in scratch, reloaded one by one inside the phases.
        // (Round 4: the copy is made from hardware state - the wave's index sits in a scalar register, the lane index is two
        // mbcnt instructions - and made AGAIN behind the tile phase: a value derived from threadIdx before the tiles and used in
        // the Adam phase is live across 256-register tiles, i.e. spilled, and its reload was a scratch round trip right behind the
        // mid-step barrier, on every wave's critical path.)
        int tid_s = hw_tid();
        int lane = tid_s & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid_s >> 6);
        bool owner = tid_s < OT;
        int pa = owner ? point_a(tid_s) : MAXK, pb = (S == 1 && owner) ? tid_s + OPT_THREADS : MAXK;
        const CoopView cv = {cws, member, (step + 1) & 1};              // this step publishes into the buffers of parity (step + 1) & 1
        const bool arrives = S == 1 || member != A.test_drop_member;    // (test hook: a member that never arrives, coop_wait's bound)
#if defined(IFD_PROF)
        KnnCounters cn{lcnt, lane, lprof + wave * PC_COUNT};
#elif defined(IFD_TRACE)
        // time stamps of one step of cloud 0: [wave][32] behind the public counters; slot PC_KNN0 = step start, PC_BUILD /
        // PC_EVAL / PC_REP = ends of those kNN sub-phases, PC_TILE0 + n = end of the wave's n-th tile, PC_TILES = tile loop
        // left, PC_WAIT = barrier passed, PC_ADAM = Adam done
        KnnCounters cn{lcnt, lane, (cloud == 0 && member == 0 && step == A.steps / 2 && counters != nullptr) ? counters + IFD_TRACE_BASE + wave * 32 : nullptr};
        PROF_ACC(PC_KNN0);
        int trace_tile = 0;
#else
        KnnCounters cn{lcnt, lane};
#endif
        // Split clouds: eight (S = 2) or four (S = 4) of a member's waves have neighbour duty, and a kNN wave beside a
        // priority-1 tile wave is starved for a whole tile (measured: 100 k cycles for its 25 k of work), so the kNN phase runs
        // FIRST at wave priority 2 (IFD_SPLIT_KNN_PRIO).  Measured at S = 2 / 4 (53 clouds x 501 steps, one S = 1 round:
        // 83.1 ms): 47.7 / 30.3 ms; without the priority 48.1 / 32.3; tiles first, kNN at the end of the step 50.6 / 34.5.
        // (S = 1, where every wave has neighbour duty: letting the upper four waves run one tile before their kNN phase, so
        // that on every SIMD one wave searches while the other feeds the MFMA pipe, measured 912 ms against 807.6 on the
        // bench launch at priority 2, 921 at equal priority, 973 with two tiles first - f32 MFMA and VALU share the issue
        // cycles, a lone tile wave runs at about half the paired rate, and the phases side by side only lengthen both.)
        // ---- decoder tiles, pulled from an LDS counter until the step's tiles run out -----------------------------------
// [pcsamp:step.tile_loop]
        auto run_tiles = [&]() __attribute__((always_inline)) {
#pragma unroll 1
            for (;;) {
                int tile = 0;
                if (lane == 0) tile = atomicAdd(tile_ctr, 1);
                tile = __builtin_amdgcn_readfirstlane(tile);            // (not __shfl: that is an LDS crossbar round trip)
                if (tile >= (S == 1 ? (K + 31) >> 5 : NT)) break;
                const int tb0 = S == 1 ? tile * 32 : tile_base(tile);
                if (S > 1 && tb0 >= K) continue;                         // (a ragged cloud: this member's tile lies beyond its end)
                const int ia = tb0 + (lane & 15), ib = ia + 16;          // two 16-point sub-tiles, software-pipelined
                const int tpa = min(ia, K - 1), tpb = min(ib, K - 1);
                float bce[2], dx[2][3];
                const float* Xf = reinterpret_cast<const float*>(X) + (lane >> 4);      // component q of the point: fc_p's B operand
#ifdef IFD_TRACE2
                unsigned long long* tr2 = (cn.pc != nullptr && trace_tile == IFD_TRACE2) ? counters + TRACE2_BASE + wave * 128 : nullptr;
                if (tr2 != nullptr && lane == 0) tr2[71] = (unsigned long long)tile;
                if (PREC == 0)
                    decoder_tile3<MODE_OPT>(W, plr, PIX[tpa], PIX[tpb], Xf[4 * tpa], Xf[4 * tpb], lane, dc,
                                            A.threshold, inv_lb, want_loss, bce, dx, tr2);
                else
                    decoder_tile3_bf<MODE_OPT, PREC>(W, plr, PIX[tpa], PIX[tpb], Xf[4 * tpa], Xf[4 * tpb], lane, dc,
                                                     A.threshold, want_loss, bce, dx, scratch + 128 + 64 * wave, tr2);
#else
                if (PREC == 0)
                    decoder_tile3<MODE_OPT>(W, plr, PIX[tpa], PIX[tpb], Xf[4 * tpa], Xf[4 * tpb], lane, dc,
                                            A.threshold, inv_lb, want_loss, bce, dx);
                else
                    decoder_tile3_bf<MODE_OPT, PREC>(W, plr, PIX[tpa], PIX[tpb], Xf[4 * tpa], Xf[4 * tpb], lane, dc,
                                                     A.threshold, want_loss, bce, dx, scratch + 128 + 64 * wave);
#endif
                if (lane < 16) {
                    if (ia < K) G[tpa] = f32x4{dx[0][0], dx[0][1], dx[0][2], bce[0]};
                    if (ib < K) G[tpb] = f32x4{dx[1][0], dx[1][1], dx[1][2], bce[1]};
                }
#ifdef IFD_TRACE
                PROF_ACC(PC_TILE0 + trace_tile);
                ++trace_tile;
#endif
            }
        };
// [pcsamp:step.knn_call]
#ifndef IFD_SPLIT_KNN_PRIO
#define IFD_SPLIT_KNN_PRIO 2          // 0: tiles first (kNN at the end of the step); 1 / 2 / 3: kNN first at that wave priority
#endif
        if (S > 1 && IFD_SPLIT_KNN_PRIO == 0) { PROF_T0(); run_tiles(); }
        // ---- kNN + repulsion of the points this wave owns (all waves at the same time: a VALU-only wave next to an
        //      MFMA-heavy tile wave on a SIMD is starved, f32 MFMA and VALU issue do not overlap on gfx950) -------------
        if (wave < OW && use_rep) {
            if (S > 1 && IFD_SPLIT_KNN_PRIO == 1) asm volatile("s_setprio 1");
            if (S > 1 && IFD_SPLIT_KNN_PRIO == 2) asm volatile("s_setprio 2");
            if (S > 1 && IFD_SPLIT_KNN_PRIO == 3) asm volatile("s_setprio 3");
            float rep_loss_a, rep_loss_b;
            // (one workgroup per cloud: the Adam phase of the last step left the certificates in registers - no scratch round
            // trip in front of the step's first dependent work)
            KnnPt ka = ka_c, kb = kb_c;
            if (!CARRY) {
                const int z = opaque_zero();
                unpark_knnpt(park, z, PARK_KNN, ka);
                unpark_knnpt(park, z, PARK_KNN + 4, kb);
            }
            TRACE_STAMP(24, "s_waitcnt vmcnt(0)");            // parked state back from scratch
            uint16_t* La = cloud_lists + (size_t)(pa & (MAXK - 1)) * LIST_M;
            uint16_t* Lb = cloud_lists + (size_t)(pb & (MAXK - 1)) * LIST_M;
            knn_phase<S>(X, F, K, pa, pb, wave, lane, step, last, A.knn_scan_every_step, La, Lb,
                         cloud_lists, ka, kb, ksh, rc, rep_loss_a, rep_loss_b, cn, cv);
            const int z2 = opaque_zero();
            park_knnpt(park, z2, PARK_KNN, ka);
            park_knnpt(park, z2, PARK_KNN + 4, kb);
            if (want_loss) {       // the last step's repulsion terms wait in the unused fourth word of the points' sampling coordinates
                if (pa < K) reinterpret_cast<float*>(PIX + pa)[3] = rep_loss_a;
                if (pb < K) reinterpret_cast<float*>(PIX + pb)[3] = rep_loss_b;
            }
            if (S > 1) {       // the member's last owner wave sends what its points owe to points of the other members
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // this wave's LDS atomics are done
                int done = 0;
                if (lane == 0) done = atomicAdd(knn_done, 1);
                if (__builtin_amdgcn_readfirstlane(done) == OW - 1) {
                    coop_flush_remote<S>(F, cv, K, lane);
                    coop_publish();
                    if (lane == 0) { *knn_done = 0; if (arrives) coop_arrive(&cv.ws->bar_knn); }
                }
                if (IFD_SPLIT_KNN_PRIO != 0) asm volatile("s_setprio 0");
            }
        }
        if (S == 1 || IFD_SPLIT_KNN_PRIO != 0) { PROF_T0(); run_tiles(); }
        if (PREC != 0) asm volatile("s_setprio 0");          // (the split-precision tile leaves its edge priority set)
        PROF_ACC(pc_tiles);
// [pcsamp:step.adam]
        // ---- Adam: its state comes back from scratch under the barrier wait ------------------------------------------------
        tid_s = hw_tid();                                        // nothing thread-derived crosses the tile phase (see the step's head)
        lane = tid_s & 63;
        owner = tid_s < OT;
        pa = owner ? point_a(tid_s) : MAXK;
        pb = (S == 1 && owner) ? tid_s + OPT_THREADS : MAXK;
        AdamState ast;
        KnnPt ka, kb;
        load_adam(MV, min(tid_s, OPT_THREADS - 1), ast);
        const int z3 = opaque_zero();
        unpark_knnpt(park, z3, PARK_KNN, ka);
        unpark_knnpt(park, z3, PARK_KNN + 4, kb);
        // (Rounds 2-3 parked the last step's repulsion terms in scratch next to the certificates and read the slot back here on
        // every step: two of its four words were never used, the register allocator re-used them while the load was in flight,
        // and the write-after-write wait it had to insert drained ALL the parked state in front of the mid-step barrier - on
        // every wave, every step.  They live in PIX[].w now - LDS, written and read on the last step only.)
        f32x2 rl = {0.f, 0.f};
        if (want_loss) rl = f32x2{pa < K ? PIX[pa].w : 0.f, pb < K ? PIX[pb].w : 0.f};
        const float step_size = adam_tab[2 * step], bc2 = adam_tab[2 * step + 1];
        // split clouds: every owner wave of the cloud has sent its neighbour terms (the wait rode under the decoder tiles)
        if (S > 1 && use_rep && wave == 0 &&
            !coop_wait(&cv.ws->bar_knn, (unsigned int)(S * (step + 1)), status, A.coop_timeout_ticks) && lane == 0)
            *coop_abort = 1;
        __syncthreads();
        PROF_ACC(pc_wait);
        if (S > 1 && want_loss) {   // the last step's per-point loss terms, reduced over the whole cloud after the loop
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int pt = q ? pb : pa;
                if (pt < K) { coop_st(&cv.ws->L[pt][0], G[pt].w); coop_st(&cv.ws->L[pt][1], q ? rl.y : rl.x); }
            }
        }
        if (S == 1 && want_loss) {   // losses at the pre-update points of the last step
            float occ = (pa < K ? G[pa].w : 0.f) + (pb < K ? G[pb].w : 0.f);
            float rep = (pa < K ? rl.x : 0.f) + (pb < K ? rl.y : 0.f);
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
        float xnew[2][3], mv2;
        adam_phase<S>(X, G, F, K, pa, pb, step_size, bc2, rep_scale, ast, xnew, mv2, status, cv);
        TRACE_STAMP(28, "s_waitcnt lgkmcnt(0)");              // Adam update done, X written
#pragma unroll
        for (int q = 0; q < 2; ++q) {        // sampling coordinates of the moved points, for the next step's decoder tiles
            const int pt = q ? pb : pa;
            if (pt < K) PIX[pt] = pix_encode(xnew[q][0], xnew[q][1], xnew[q][2], dc);
        }
        TRACE_STAMP(29, "s_waitcnt lgkmcnt(0)");              // PIX written
        if (owner) store_adam(MV, tid_s, ast);
        adam_displacement(K, pa, pb, member * OW + wave, lane, step, xnew, mv2, ka, kb, ksh, wave < OW);     // needs the parked state: last
// [pcsamp:step.end_barrier]
        if (CARRY) { ka_c = ka; kb_c = kb; }
        if (S == 1) {
            if (tid == 0) { rebuild_flag[step & 1] = 0; *tile_ctr = 0; }   // both consumed before the mid-step barrier
            __syncthreads();
        } else {
            // ---- the step's exchange between the members of a split cloud ------------------------------------------------
            const int np = (step + 1) & 1;
            if (wave < OW && lane == 0) {      // this owner wave's certificate maxima of the next step (adam_displacement)
                coop_st(&cv.ws->scal[np][member * OW + wave][0], dmaxbuf[np * MAX_WAVES + member * OW + wave]);
                coop_st(&cv.ws->scal[np][member * OW + wave][1], movebuf[np * MAX_WAVES + member * OW + wave]);
            }
            if (tid == 0) __hip_atomic_store(&cv.ws->flag[np][member], (int)rebuild_flag[np], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            coop_publish();                    // positions (adam_phase), maxima, flag: complete before this member arrives
            __syncthreads();
            if (wave == 0) {
                if (lane == 0 && arrives) coop_arrive(&cv.ws->bar_step);
                if (!coop_wait(&cv.ws->bar_step, (unsigned int)(S * (step + 1)), status, A.coop_timeout_ticks) && lane == 0)
                    *coop_abort = 1;
            }
            if (tid == 0) { rebuild_flag[step & 1] = 0; *tile_ctr = 0; }
            __syncthreads();
            if (*coop_abort != 0) break;       // a member never came (block-uniform): leave with what there is, the status word is set
            for (int i = tid_s; i < K; i += NW * 64)       // the other members' points
                if (!coop_owns<S>(i, member)) {
                    const float* xg = reinterpret_cast<const float*>(cv.ws->X[np] + i);
                    X[i] = f32x4{coop_ld(xg), coop_ld(xg + 1), coop_ld(xg + 2), 1.f};
                }
            if (tid_s < S * OW && tid_s / OW != member) {     // ... and their owner waves' maxima (slot = owner wave of the cloud)
                dmaxbuf[np * MAX_WAVES + tid_s] = coop_ld(&cv.ws->scal[np][tid_s][0]);
                movebuf[np * MAX_WAVES + tid_s] = coop_ld(&cv.ws->scal[np][tid_s][1]);
            }
            if (tid_s >= 64 && tid_s < 64 + S && tid_s - 64 != member) {     // the other members' expiring certificates
                const int nbad = __hip_atomic_load(&cv.ws->flag[np][tid_s - 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (nbad != 0) atomicAdd(const_cast<int*>(rebuild_flag) + np, nbad);
            }
            __syncthreads();
        }
        PROF_ACC(pc_adam);
    }

// [pcsamp:kernel.epilogue]
    if (counters != nullptr) {
        if (tid < CN_COUNT) {
            constexpr int SLOT[CN_COUNT] = {0, 1, 2, 4, 5, 6, 7};     // rebuilds, exact scans, extra passes, ring evaluations,
            atomicAdd(counters + SLOT[tid], (unsigned long long)lcnt[tid]);   // exact-path evaluations, refreshes, lists built
        }
        if (tid == 0 && cloud == 0 && member == 0) counters[3] = __builtin_readcyclecounter() - t_begin;   // shader cycles of cloud 0
#ifdef IFD_PROF
        if (tid == 0) {
            const unsigned long long cyc = __builtin_readcyclecounter() - t_begin;
            atomicMax(counters + 14, cyc);
            atomicMax(counters + 13, (unsigned long long)lcnt[CN_REBUILD] << 32);
            atomicAdd(counters + 15, cyc);
        }
        if (cloud == 0 && tid < NW * PC_COUNT && (tid % PC_COUNT) < 6) atomicAdd(counters + 8 + tid % PC_COUNT, lprof[tid]);
#endif
    }
    if (S > 1 && loss_out != nullptr && A.steps > 0 && member == 0) {
        // the last step's losses: every point's terms, summed exactly like the one-workgroup kernel sums them (thread t adds
        // points t and t + 512, a butterfly over the wave, the eight waves in order)
        const int qa = tid < OPT_THREADS ? tid : MAXK, qb = tid < OPT_THREADS ? tid + OPT_THREADS : MAXK;
        float occ = (qa < K ? coop_ld(&cv.ws->L[qa][0]) : 0.f) + (qb < K ? coop_ld(&cv.ws->L[qb][0]) : 0.f);
        float rep = (qa < K ? coop_ld(&cv.ws->L[qa][1]) : 0.f) + (qb < K ? coop_ld(&cv.ws->L[qb][1]) : 0.f);
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
        __syncthreads();
    }
    if (A.normalize) normalize_in_lds(X, K, scratch);
    AdamState ast;
    load_adam(MV, min(tid, OPT_THREADS - 1), ast);
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

// ---------------------------------------------------------------------------------------------
// Clouds of more than MAXK optimised points (--sample_npoint 1025 ... LARGE_MAXK; ConvONet/opt_defense.py:27,149-179): the
// occupancy half of the launch-per-step path (optimize.hip, "large" section) on the PERSISTENT kernel's decoder tile -
// decoder_tile3 / decoder_tile3_bf, two software-pipelined 16-point sub-tiles per wave, eight waves of 256 registers per CU -
// so that a point of a large cloud costs what a point of a 1024-point cloud costs (rounds 2-5 ran the stand-alone one-sub-tile
// tile here: 2.5 x per point at K = 2048), in every ifd_opt_params.precision.  Workgroup (cloud, part) takes the cloud's
// 32-point tiles part * 8 + wave, + 8 * parts, ...; writes G[cloud][point] = {d loss / d xyz, BCE term}.
// ---------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(OPT_THREADS, 1) void large_occupancy3_kernel(
    const float* __restrict__ dec_img, const float* __restrict__ planes, const float* __restrict__ p, int K,
    const int32_t* __restrict__ loss_batch_per_cloud, int loss_batch, float thr, int want_loss_i, f32x4* __restrict__ G, DecConst dc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int IMG_FLOATS = PREC == 0 ? DEC_FLOATS : BF_IMG_BYTES / 4;
    float* W = smem;
    [[maybe_unused]] float* strips = smem + IMG_FLOATS;              // PREC > 0: one 256-byte landing strip per wave (tile_bf.h)
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    load_dec_image(W, dec_img, IMG_FLOATS);
    __syncthreads();
    const int lb = loss_batch_per_cloud ? loss_batch_per_cloud[cloud] : loss_batch;
    const float inv_lb = 1.0f / (float)lb;
    if (tid == 0) W[(PREC == 0 ? DEC_OFF_BOUT : BF_OFF_BOUT / 4) + 1] = inv_lb;     // the tile reads {fc_out's bias, 1 / B} as one pair
    __syncthreads();
    const float* pl = planes + (size_t)cloud * CLOUD_PLANE_FLOATS;
    const __amdgpu_buffer_rsrc_t plr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pl), 0, CLOUD_PLANE_FLOATS * 4, 0x00020000);
    const float* pc = p + (size_t)cloud * K * 3;
    const bool want_loss = want_loss_i != 0;
    const int ntiles = (K + 31) >> 5, q = lane >> 4;
#pragma unroll 1
    for (int tile = (int)blockIdx.y * (OPT_THREADS / 64) + wave; tile < ntiles; tile += (int)gridDim.y * (OPT_THREADS / 64)) {
        const int ia = tile * 32 + (lane & 15), ib = ia + 16;
        const int tpa = min(ia, K - 1), tpb = min(ib, K - 1);
        const float xa0 = pc[3 * tpa], xa1 = pc[3 * tpa + 1], xa2 = pc[3 * tpa + 2];
        const float xb0 = pc[3 * tpb], xb1 = pc[3 * tpb + 1], xb2 = pc[3 * tpb + 2];
        const f32x4 ppa = pix_encode(xa0, xa1, xa2, dc), ppb = pix_encode(xb0, xb1, xb2, dc);
        const float xqa = q == 0 ? xa0 : q == 1 ? xa1 : q == 2 ? xa2 : 1.f;          // component q of (x, y, z, 1): fc_p's B operand
        const float xqb = q == 0 ? xb0 : q == 1 ? xb1 : q == 2 ? xb2 : 1.f;
        float bce[2], dx[2][3];
        if constexpr (PREC == 0)
            decoder_tile3<MODE_OPT>(W, plr, ppa, ppb, xqa, xqb, lane, dc, thr, inv_lb, want_loss, bce, dx);
        else {
            decoder_tile3_bf<MODE_OPT, PREC>(W, plr, ppa, ppb, xqa, xqb, lane, dc, thr, want_loss, bce, dx, strips + 64 * wave);
            asm volatile("s_setprio 0");
        }
        if (lane < 16) {
            if (ia < K) G[(size_t)cloud * K + tpa] = f32x4{dx[0][0], dx[0][1], dx[0][2], bce[0]};
            if (ib < K) G[(size_t)cloud * K + tpb] = f32x4{dx[1][0], dx[1][1], dx[1][2], bce[1]};
        }
    }
}

}  // namespace ifd
