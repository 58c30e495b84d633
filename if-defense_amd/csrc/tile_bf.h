// The optimiser's decoder tile on the bf16 matrix core with f32-exact operand splits ("split precision", SURVEY 8f N4;
// ConvONet/src/conv_onet/models/decoder.py:83-93, layers.py:39-48).  Included by optimize_kernel.h inside namespace ifd,
// behind decoder_tile3 (whose geometry / mask / accumulator helpers it shares).
//
// Why: an f32 MFMA runs on the vector ALU's multipliers - it excludes every vector instruction on its SIMD
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/r04_pmc_fifo.txt), so the f32 tile's time is the SUM of its matrix and vector
// cycles and a bubble-free f32 kernel ends at 0.74 of the f32 peak.  A bf16 MFMA runs on the matrix core proper,
// 16x faster, beside the vector pipe.  Every f32 value is the exact sum of three bf16 pieces (8 + 8 + 8 mantissa bits, round
// to nearest: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)), so with both operands split
//     w x = sum_ij w_i x_j;   PREC 1 ("bf16x6"): w1 x1 + w1 x2 + w2 x1 + w2 x2 + w1 x3 + w3 x1   (dropped: <= 2^-26 |w x|)
//                             PREC 2 ("bf16x3"): w1 x1 + w1 x2 + w2 x1                           (dropped: <= 2^-17 |w x|)
// every product of pieces is exact in the MFMA's f32 accumulation, one instruction sums 32 of them with ONE rounding (the f32
// MFMA chain rounds 32 times), and the six-term form measures MORE accurate than the f32 MFMA chain against float64
// (profiles/r05_bf16x6_probe.txt: max error 3.9e-7 against 9.1e-7 on one layer) at half its cycles.
//
// The splits are made on the matrix pipe too: r = relu(x) - x1 is ONE MFMA with A = minus a selection matrix, B = the x1 piece
// that has just been packed, C = relu(x) (exact: x1 is within 2^-9 of C and the other products are exact zeros); the vector
// pipe only converts (v_cvt_pk_bf16_f32: 12 per 8 values and layer).  Per layer and 16-point sub-tile: 12 + 4 MFMAs of
// 16x16x32 (16 cycles each) against 16 of 16x16x4 f32 (32 cycles each) - and the vector instructions run beside them.
//
// Operand mapping (v_mfma_f32_16x16x32_bf16: M = output channel, N = point, K = input channel): lane (n = lane & 15, g = lane >> 4)
// holds A[m = n][k = 8 g + j] and B[k = 8 g + j][n], j = 0 ... 7, and D[4 g + r][n].  With two M-tiles a lane holds the channels
// bf_chan(g, e) = 16 (e >> 2) + 4 g + (e & 3), e = 4 mt + r, of its point - the f32 tile's layout - and k-slot 8 g + j of the next
// layer IS accumulator register j: activations stay in registers, the weight image is permuted instead (ifd_device.h).
#pragma once

#include "split_bf16.h"

struct WFragBF {
    bf16x8 a[3][2];     // [piece][M-tile]
};

// acc += W x from the pieces, small terms first; the two M-tiles are independent accumulator chains.  Term k multiplies weight
// piece TA[k] by activation piece TP[k]:  bf16x6: w3 x1, w1 x3, w2 x2, w2 x1, w1 x2, w1 x1;  bf16x3: w2 x1, w1 x2, w1 x1.
template <int PREC>
__device__ __forceinline__ void dense_terms(const WFragBF& A, const Pieces& P, Acc2& acc, int k0, int k1) {
    constexpr int TA6[6] = {2, 0, 1, 1, 0, 0}, TP6[6] = {0, 2, 1, 0, 1, 0};
    constexpr int TA3[3] = {1, 0, 0}, TP3[3] = {0, 1, 0};
#pragma unroll
    for (int k = k0; k < k1; ++k) {
        const int ta = PREC == 1 ? TA6[k] : TA3[k], tp = PREC == 1 ? TP6[k] : TP3[k];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc.t[mt] = mfma_bf(A.a[ta][mt], P.p[tp], acc.t[mt]);
    }
}
template <int PREC>
__device__ __forceinline__ void dense_bf(const WFragBF& A, const Pieces& P, Acc2& acc) {
    dense_terms<PREC>(A, P, acc, 0, PREC == 1 ? 6 : 3);
}

// One software-pipeline region: the 12 (6) MFMAs of one sub-tile's layer (acc += A Pin) beside the epilogue of the other's - its
// vector instructions in NS stages of <= 4 (epi(stage, x): bias / ReLU / mask), then the split of x into Pout, whose residual
// MFMAs and conversions are dealt between the layer's MFMAs.  The matrix pipe takes one v_mfma_f32_16x16x32_bf16 per 16
// cycles and the vector pipe runs beside it, so the stream alternates ONE MFMA with <= 4 vector instructions and puts >= 2
// other MFMAs between a residual MFMA and the conversion that reads it: a lone wave then keeps the matrix pipe busy (an in-order
// wave only overlaps what is adjacent in its instruction stream; left to itself hipcc issues the twelve MFMAs first and the
// split chain behind them with ~25 wait states between its steps).  The order is pinned with scheduling barriers.
#define BF_SB() __builtin_amdgcn_sched_barrier(0)
// Wave priorities (a tuning knob, off by default: measured within 2 %): the sections around the MLP at IFD_BF_PRIO_EDGE, the MLP
// sections at IFD_BF_PRIO_MLP.
#ifndef IFD_BF_PRIO_EDGE
#define IFD_BF_PRIO_EDGE 0
#endif
#ifndef IFD_BF_PRIO_MLP
#define IFD_BF_PRIO_MLP 0
#endif
#define BF_PRIO_STR2(x) #x
#define BF_PRIO_STR(x) BF_PRIO_STR2(x)
#define BF_PRIO(p) do { if (IFD_BF_PRIO_EDGE != IFD_BF_PRIO_MLP) asm volatile("s_setprio " BF_PRIO_STR(p)); } while (0)
template <int PREC>
__device__ __forceinline__ void dense_one(const WFragBF& A, const Pieces& P, Acc2& acc, int k) {        // MFMA k of the layer: term k / 2, M-tile k % 2
    constexpr int TA6[6] = {2, 0, 1, 1, 0, 0}, TP6[6] = {0, 2, 1, 0, 1, 0};
    constexpr int TA3[3] = {1, 0, 0}, TP3[3] = {0, 1, 0};
    const int t = k >> 1, mt = k & 1;
    const int ta = PREC == 1 ? TA6[t] : TA3[t], tp = PREC == 1 ? TP6[t] : TP3[t];
    acc.t[mt] = mfma_bf(A.a[ta][mt], P.p[tp], acc.t[mt]);
    BF_SB();
}
__device__ __forceinline__ void cvt4_bf(bf16x8& o, const f32x4& v, int h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[4 * h + j] = (__bf16)v[j];
    BF_SB();
}
template <int PREC, int NS, typename Epi>
__device__ __forceinline__ void region_bf(const WFragBF& A, const Pieces& Pin, Acc2& acc, const SelMat& sel, Pieces& Pout, Epi epi) {
    f32x8 x;
    int d = 0;                                     // next MFMA of the layer
    BF_SB();
#pragma unroll
    for (int st = 0; st < NS; ++st) {              // the epilogue's stages, one MFMA in front of each
        dense_one<PREC>(A, Pin, acc, d++);
        epi(st, x);
        BF_SB();
    }
    f32x4 c0 = {x[0], x[1], x[2], x[3]}, c1 = {x[4], x[5], x[6], x[7]};
    dense_one<PREC>(A, Pin, acc, d++);
    cvt4_bf(Pout.p[0], c0, 0);
    if (PREC == 1 || NS < 4) dense_one<PREC>(A, Pin, acc, d++);
    cvt4_bf(Pout.p[0], c1, 1);
    c0 = mfma_bf(sel.m[0], Pout.p[0], c0);         // residual 1 ...
    BF_SB();
    c1 = mfma_bf(sel.m[1], Pout.p[0], c1);
    BF_SB();
    dense_one<PREC>(A, Pin, acc, d++);             // ... its latency under the layer's next MFMAs
    if (PREC == 1) {
        dense_one<PREC>(A, Pin, acc, d++);
        cvt4_bf(Pout.p[1], c0, 0);
        dense_one<PREC>(A, Pin, acc, d++);
        cvt4_bf(Pout.p[1], c1, 1);
        c0 = mfma_bf(sel.m[0], Pout.p[1], c0);     // residual 2
        BF_SB();
        c1 = mfma_bf(sel.m[1], Pout.p[1], c1);
        BF_SB();
        while (d < 11) dense_one<PREC>(A, Pin, acc, d++);
        cvt4_bf(Pout.p[2], c0, 0);
        dense_one<PREC>(A, Pin, acc, d++);
        cvt4_bf(Pout.p[2], c1, 1);
        while (d < 12) dense_one<PREC>(A, Pin, acc, d++);
    } else {
        while (d < 6) dense_one<PREC>(A, Pin, acc, d++);
        cvt4_bf(Pout.p[1], c0, 0);
        cvt4_bf(Pout.p[1], c1, 1);
        Pout.p[2] = Pout.p[1];
    }
    BF_SB();
}

typedef __attribute__((address_space(3))) unsigned char lds_u8;

// The epilogues of region_bf in stages of <= 4 vector instructions.  Forward: ReLU (one v_max_i32 per value, relu8) and the sign-byte
// mask of decoder_tile3 (mask_alive_packed: two v_perm + one v_xnor per four values); backward: the mask applied (one SDWA
// v_and per value, masked()), for the a-masks followed by the residual add.
__device__ __forceinline__ void relu_mask_stage(const Acc2& a, Mask8& m, int st, f32x8& x) {
    if (st < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[4 * st + r] = __int_as_float(max(__float_as_int(a.t[st][r]), 0));
    } else {
        const int h = st - 2;
        const uint32_t hi = __builtin_amdgcn_perm(__float_as_uint(a.t[h][0]), __float_as_uint(a.t[h][1]), 0x0b090c0cu);
        const uint32_t lo = __builtin_amdgcn_perm(__float_as_uint(a.t[h][2]), __float_as_uint(a.t[h][3]), 0x0c0c0b09u);
        uint32_t w = ~(hi ^ lo);
        asm volatile("" : "+v"(w));
        m.w[h] = w;
    }
}
__device__ __forceinline__ void masked_stage(const Acc2& z, const Mask8& m, int st, f32x8& x) {
    x[4 * st + 0] = keep_alive<3>(z.t[st][0], m.w[st]); x[4 * st + 1] = keep_alive<2>(z.t[st][1], m.w[st]);
    x[4 * st + 2] = keep_alive<1>(z.t[st][2], m.w[st]); x[4 * st + 3] = keep_alive<0>(z.t[st][3], m.w[st]);
}
__device__ __forceinline__ void masked_add_stage(const Acc2& y, const Mask8& m, f32x8& dn, int st, f32x8& x) {
    const int h = st >> 1;
    if ((st & 1) == 0) {
        x[4 * h + 0] = keep_alive<3>(y.t[h][0], m.w[h]); x[4 * h + 1] = keep_alive<2>(y.t[h][1], m.w[h]);
        x[4 * h + 2] = keep_alive<1>(y.t[h][2], m.w[h]); x[4 * h + 3] = keep_alive<0>(y.t[h][3], m.w[h]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { dn[4 * h + r] += x[4 * h + r]; x[4 * h + r] = dn[4 * h + r]; }
    }
}

// (scalar: no packed-f32 instructions in this tile, see decoder_tile3_bf)
__device__ __forceinline__ Acc2 acc_add_s(const Acc2& a, const Acc2& b) {
    Acc2 r;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) r.t[mt][j] = a.t[mt][j] + b.t[mt][j];
    return r;
}
__device__ __forceinline__ void add8_s(f32x8& d, const f32x8& t) {
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] += t[r];
}

// A operands of one layer.  Forward: the entry of (piece, M-tile, lane) as it lies.  Transposed (backward: A = W^T, rows = input
// channel 16 mt + m, k-slot 8 g + j = output channel bf_chan(g, j)): ds_read_b64_tr_b16 hands lane L of a 16-lane block, for
// j' = 0 ... 3, element (L & 3) of the 8 bytes that lane 4 j' + (L >> 2) of the block addressed (profiles/r05_ds_read_tr16_probe.txt).
// So block g's lane R = 4 j' + mh addresses the four input channels 16 mt + 4 mh + (0 ... 3) of output row 16 jh + 4 g + j' - the
// entry (M-tile jh, lane (4 g + j', mh)) of the forward image at element 4 mt - and lane m receives W[16 jh + 4 g + j'][16 mt + m],
// j' = 0 ... 3: the four elements 4 jh + j' of its transposed operand.  Two reads (jh = 0, 1) per operand.
template <bool TRANSPOSED, int PREC>
__device__ __forceinline__ WFragBF load_wfrag_bf(const lds_u8* __restrict__ Wb, int layer, int lane_off, int lane_off_t) {
    WFragBF f;
    constexpr int NP = PREC == 1 ? 3 : 2;
    const lds_u8* base = Wb + layer * BF_LAYER_BYTES;
#pragma unroll
    for (int sp = 0; sp < NP; ++sp)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (!TRANSPOSED) {
                f.a[sp][mt] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(base + sp * BF_PIECE_BYTES + mt * 1024 + lane_off);
            } else {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(base + sp * BF_PIECE_BYTES + 0 * 1024 + lane_off_t + 8 * mt));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(base + sp * BF_PIECE_BYTES + 1 * 1024 + lane_off_t + 8 * mt));
                f.a[sp][mt] = __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
            }
        }
    if (PREC != 1) { f.a[2][0] = f.a[1][0]; f.a[2][1] = f.a[1][1]; }
    return f;
}

// ---------------------------------------------------------------------------------------------------------------------
// Two 16-point sub-tiles per wave, software-pipelined like decoder_tile3: the 12 MFMAs of one sub-tile's layer beside the
// epilogue (bias, ReLU, mask, split: ~27 vector instructions + 4 MFMAs) of the other's.
// ---------------------------------------------------------------------------------------------------------------------
template <int MODE, int PREC>
__device__ __forceinline__ void decoder_tile3_bf(const float* __restrict__ Wg, __amdgpu_buffer_rsrc_t planes,
                                                 const f32x4 ppa, const f32x4 ppb, float xqa, float xqb, int lane,
                                                 const DecConst dc, float thr, bool want_loss, float (&bce)[2], float (&dx)[2][3],
                                                 [[maybe_unused]] float* pf_strip, [[maybe_unused]] unsigned long long* tr = nullptr) {
    T2(0);                                       // (-DIFD_TRACE2 stamps: decoder_tile3's slot map, scripts/tile_trace.py)
    BF_PRIO(IFD_BF_PRIO_EDGE);
    const lds_u8* Wb = (const lds_u8*)Wg;
    const int n = lane & 15, q = lane >> 4;
    int lane_off = lane * BF_ENTRY_BYTES;
#ifdef IFD_BF_TIMING_TR_LINEAR                   // timing experiment, WRONG results: the transposed reads without their bank conflicts
    int lane_off_t = lane * 8;
#else
    int lane_off_t = ((lane & 3) * 16 + 4 * q + ((lane >> 2) & 3)) * BF_ENTRY_BYTES;
#endif
    int q4 = 4 * q;
    asm volatile("" : "+v"(lane_off), "+v"(lane_off_t), "+v"(q4));
    constexpr int AX0[3] = {0, 0, 1}, AX1[3] = {2, 1, 2};   // xz, xy, yz (common.py:243-248)
    const float ksc = ((0.5f * (float)(RES - 1)) * 2.f) / dc.sdiv;
    SubGeo geo[2];
    sub_geometry(geo[0], ppa, ksc);
    sub_geometry(geo[1], ppb, ksc);
    const float* Wf = reinterpret_cast<const float*>(Wg);
    int boff = BF_OFF_BIAS / 4 + q4;
    asm volatile("" : "+v"(boff));
    const float* Bq = Wf + boff;
    auto bias = [&](int layer) {
        Acc2 b;
        b.t[0] = *reinterpret_cast<const f32x4*>(Bq + layer * 32);
        b.t[1] = *reinterpret_cast<const f32x4*>(Bq + layer * 32 + 16);
        return b;
    };

    // ---- gather + forward bilinear sample (decoder_tile3's, unchanged) ------------------------------------------------
    f32x8 c[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 8; ++r) c[t][r] = 0.f;
    const int q16 = 16 * q;
#ifndef IFD_BF_TIMING_NOGATHER
#define IFD_BF_TIMING_NOGATHER 0      // timing experiments with WRONG results: 1 no backward re-gather, 2 no gather at all
#endif
    auto load_taps = [&](int P, f32x4 (&tap)[2][4][2], bool opaque) {
        const int a0 = AX0[P], a1 = AX1[P];
        if ((IFD_BF_TIMING_NOGATHER == 1 && opaque) || IFD_BF_TIMING_NOGATHER == 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) tap[t][k][mt] = f32x4{geo[t].w0[a0], geo[t].w1[a1], 0.5f, 0.25f};
            return;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int off = (geo[t].cell[a1] * RES + geo[t].cell[a0]) * (CH * 4) + (P * PLANE_FLOATS * 4 + q16);
            if (opaque) asm volatile("" : "+v"(off));
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
    // L2 prefetch of the tile's 384 tap lines (3 planes x 2 rows x [32 points x 2 columns]) with six LDS-DMA loads - one dword per lane
    // and line, no destination registers (buffer_load ... lds into this wave's 256-byte landing strip, never read): the backward
    // pass re-gathers the taps ~15 k cycles after the forward gather fetched them, by which time they have left the XCD's L2
    // (its 4 MB turn over in ~7 us under 32 CUs' gathers).  Lane (n, q) takes point n of sub-tile q & 1, column q >> 1.
#ifndef IFD_BF_PREFETCH
#define IFD_BF_PREFETCH 0      // measured: +11 % (bf16x6) / +27 % (bf16x3) - the kernel is bound by the tap TRAFFIC, not by its latency (profiles/r05_ab_bf_prefetch.txt)
#endif
    auto prefetch_taps = [&]() {
        const bool t1 = (q & 1) != 0;
#pragma unroll
        for (int P = 0; P < 3; ++P) {
            const int a0 = AX0[P], a1 = AX1[P];
            const int cx = (t1 ? geo[1].cell[a0] : geo[0].cell[a0]) + (q >> 1), cy = t1 ? geo[1].cell[a1] : geo[0].cell[a1];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int off = ((cy + h) * RES + cx) * (CH * 4) + P * PLANE_FLOATS * 4;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(planes, (__attribute__((address_space(3))) void*)pf_strip, 4, off, 0, 0, 0);
            }
        }
    };
    // NO packed-f32 instruction anywhere in this tile (nor, through -fno-slp-vectorize, in the rest of this kernel): beside a wave
    // that streams bf16 MFMAs, a v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 whose consumer does not follow back to back
    // occasionally delivers a wrong result on gfx950 (scripts/pk_mfma_coexec.hip, profiles/r05_pk_mfma_coexec.txt: 0 wrong of 5e9
    // beside an idle or an f32-MFMA partner, 16 ... 300 of 5e9 beside a bf16-MFMA partner - ~1 % of a launch's sub-tiles here).
    // decoder_tile3's packed sampling is therefore spelled out value by value: same operations, same order, same roundings.
    // The sampling Jacobian (IFD_BF_JAC; default: bf16x3 only): d c[ch] / d pix along the plane's two axes, summed over the planes that share an
    // axis - 3 x 8 values per lane and sub-tile, parked in scratch across the MLP - replaces the backward pass's re-gather of the taps
    // (48 KB of 128-byte lines per tile and wave, the kernel's largest HBM stream: profiles/r05_ab_bf_gather.txt) by 12 KB of
    // coalesced scratch stores and loads.  J[t][a][ch]: axis 0 <- planes 0, 1; axis 1 <- planes 1, 2; axis 2 <- planes 0, 2.
    // Measured on the bench workload (profiles/r05_ab_bf_jacobian.txt): bf16x3 4068 -> 4263 clouds/s (+4.8 %, -11 % cycles per step);
    // bf16x6 3531 -> 3458 (-2 %: 2 % fewer cycles, but the ~200 extra vector instructions per tile pull the clock from 2.14 to 2.06
    // GHz - that kernel sits at the socket's power limit, profiles/r05_power_clock.txt).  So: on for bf16x3, off for bf16x6.
#ifndef IFD_BF_JAC
#define IFD_BF_JAC (PREC == 2)
#endif
    [[maybe_unused]] float J[2][3][8];
    auto sample_fwd = [&](int P, const f32x4 (&tap)[2][4][2]) {
        const int a0 = AX0[P], a1 = AX1[P];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const SubGeo& g = geo[t];
            const float wnw = g.w0[a0] * g.w0[a1], wne = g.w1[a0] * g.w0[a1], wsw = g.w0[a0] * g.w1[a1],
                        wse = g.w1[a0] * g.w1[a1];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float s1 = tap[t][0][mt][j] * wnw;
                    s1 = __builtin_fmaf(tap[t][1][mt][j], wne, s1);
                    s1 = __builtin_fmaf(tap[t][2][mt][j], wsw, s1);
                    s1 = __builtin_fmaf(tap[t][3][mt][j], wse, s1);
                    c[t][4 * mt + j] = P == 0 ? s1 : c[t][4 * mt + j] + s1;
                    if (IFD_BF_JAC && MODE == MODE_OPT) {
                        const float nw = tap[t][0][mt][j], ne = tap[t][1][mt][j], sw = tap[t][2][mt][j], se = tap[t][3][mt][j];
                        const float jx = __builtin_fmaf(se - sw, g.w1[a1], (ne - nw) * g.w0[a1]);       // along a0
                        const float jy = __builtin_fmaf(se - ne, g.w1[a0], (sw - nw) * g.w0[a0]);       // along a1
                        const int ch = 4 * mt + j;
                        // first contributions: axis 0 and axis 2 from plane 0, axis 1 from plane 1
                        J[t][a0][ch] = (P == 0 || (P == 1 && a0 == 1)) ? jx : J[t][a0][ch] + jx;
                        J[t][a1][ch] = (P == 0 || (P == 1 && a1 == 1)) ? jy : J[t][a1][ch] + jy;
                    }
                }
        }
    };
    {
        f32x4 tap0[2][4][2], tap1[2][4][2], tap2[2][4][2];
        load_taps(0, tap0, false);
        load_taps(1, tap1, false);
        load_taps(2, tap2, false);
        __builtin_amdgcn_sched_barrier(0);
        T2(1);
#ifdef IFD_TRACE2
        if (tr != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        T2(2);
#endif
        // one explicit wait per plane (s_waitcnt vmcnt(32) / (16) / (0): the planes' 16 loads each, oldest first) instead of hipcc's
        // load-by-load waits inside the multiply-add chains
        __builtin_amdgcn_s_waitcnt(0x8F70);
        __builtin_amdgcn_sched_barrier(0);
        sample_fwd(0, tap0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x4F70);
        __builtin_amdgcn_sched_barrier(0);
        sample_fwd(1, tap1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
        sample_fwd(2, tap2);
        __builtin_amdgcn_sched_barrier(0);
        T2(3);
    }
    // park the Jacobian (12 x 16 bytes per lane; a 256-byte private array indexed through an opaque zero: knn_device.h "Parking")
    [[maybe_unused]] f32x4 jpark[16];
    if (IFD_BF_JAC && MODE == MODE_OPT) {
        const int z = opaque_zero();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    jpark[z + (t * 3 + a) * 2 + h] = f32x4{J[t][a][4 * h], J[t][a][4 * h + 1], J[t][a][4 * h + 2], J[t][a][4 * h + 3]};
    }

    const SelMat sel = make_selmat(lane);
    // ---- fc_p on the f32 matrix instruction (K = 4: x, y, z, 1 - two MFMAs per sub-tile, as in decoder_tile3) --------------
    Acc2 net[2];
    {
        const float ap0 = Wf[BF_OFF_WP / 4 + n * 4 + q], ap1 = Wf[BF_OFF_WP / 4 + (16 + n) * 4 + q];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float xq = t ? xqb : xqa;
            net[t].t[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap0, xq, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            net[t].t[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap1, xq, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
    }
    WFragBF A = load_wfrag_bf<false, PREC>(Wb, 0, lane_off, lane_off_t);
    Pieces PC[2];                                  // the sampled features c: the input of all five fc_c
    split_bf<PREC>(c[0], sel, PC[0]);
    split_bf<PREC>(c[1], sel, PC[1]);
    __builtin_amdgcn_sched_barrier(0);
    T2(4);
    BF_PRIO(IFD_BF_PRIO_MLP);

    // ---- forward MLP -----------------------------------------------------------------------------------------------
    Mask8 mask_a[2][NBLK], mask_h[2][NBLK];
    f32x8 wout;
    float bout = 0.f, ilb = 0.f;
    int c31 = 31;
    asm volatile("" : "+v"(c31));
    const unsigned long long sc2 = 0ull;           // (packed-ReLU scale of decoder_tile3: RELU_K = 0 here)
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        // fc_c: a = n + fc_c(c)   (bias folded into n)
        Acc2 a0 = net[0], a1 = net[1];
        dense_bf<PREC>(A, PC[0], a0);                                          // R1
        BF_SB();
        T2(5 + 6 * i + 0);
        const WFragBF A0 = load_wfrag_bf<false, PREC>(Wb, 3 * i + 1, lane_off, lane_off_t);   // R2: prefetch fc_0
        const Acc2 B0 = bias(3 * i + 1);
        Pieces PA0, PA1, PH0, PH1;
        region_bf<PREC, 4>(A, PC[1], a1, sel, PA0, [&](int st, f32x8& x) { relu_mask_stage(a0, mask_a[0][i], st, x); });
        T2(5 + 6 * i + 1);
        Acc2 h0 = B0;                                                          // R3
        region_bf<PREC, 4>(A0, PA0, h0, sel, PA1, [&](int st, f32x8& x) { relu_mask_stage(a1, mask_a[1][i], st, x); });
        T2(5 + 6 * i + 2);
        const WFragBF A1 = load_wfrag_bf<false, PREC>(Wb, 3 * i + 2, lane_off, lane_off_t);   // R4: prefetch fc_1
        const Acc2 B1 = bias(3 * i + 2);
        Acc2 h1 = B0;
        region_bf<PREC, 4>(A0, PA1, h1, sel, PH0, [&](int st, f32x8& x) { relu_mask_stage(h0, mask_h[0][i], st, x); });
        T2(5 + 6 * i + 3);
        Acc2 o0 = acc_add_s(B1, a0);                                           // R5
        region_bf<PREC, 4>(A1, PH0, o0, sel, PH1, [&](int st, f32x8& x) { relu_mask_stage(h1, mask_h[1][i], st, x); });
        T2(5 + 6 * i + 4);
        if (i + 1 < NBLK) {                                                    // R6: prefetch the next fc_c / the first fc_1^T
            A = load_wfrag_bf<false, PREC>(Wb, 3 * i + 3, lane_off, lane_off_t);
        } else {
            A = load_wfrag_bf<true, PREC>(Wb, 3 * i + 2, lane_off, lane_off_t);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(Wf + BF_OFF_WOUT / 4 + q4);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(Wf + BF_OFF_WOUT / 4 + 16 + q4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { wout[r] = t0[r]; wout[4 + r] = t1[r]; }
            const f32x2 bo = *reinterpret_cast<const f32x2*>(Wf + BF_OFF_BOUT / 4);
            bout = bo.x;
            ilb = bo.y;
        }
        Acc2 o1 = acc_add_s(B1, a1);
        dense_bf<PREC>(A1, PH1, o1);
        net[0] = o0;
        net[1] = o1;
        BF_SB();
        T2(5 + 6 * i + 5);
    }
    // ---- logit, loss derivative, seed of the backward pass (decoder_tile3's) ---------------------------------------------
    BF_PRIO(IFD_BF_PRIO_EDGE);
    f32x8 dn[2];
    Pieces PD[2];
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
            const float e = expf(-fabsf(logit));
            const float rc = __builtin_amdgcn_rcpf(1.f + e);
            const float sig = logit >= 0.f ? rc : e * rc;
            dl = (sig - thr) * ilb;
            bce[t] = want_loss ? fmaxf(logit, 0.f) - thr * logit + log1pf(e) : 0.f;
        } else {
            bce[t] = logit;
            dl = 1.f;
        }
        f32x8 dw;
#pragma unroll
        for (int r = 0; r < 8; ++r) dw[r] = dl * wout[r];
        dn[t] = masked(dw, mask_n);
        split_bf<PREC>(dn[t], sel, PD[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
    T2(35);
    BF_PRIO(IFD_BF_PRIO_MLP);

    if (IFD_BF_PREFETCH & 1) prefetch_taps();      // (for the re-gather behind the backward MLP; off: see IFD_BF_PREFETCH)
    // ---- backward (A holds fc_1[4]^T) --------------------------------------------------------------------------------------
    Acc2 dcc[2] = {acc_zero(), acc_zero()};
#pragma unroll
    for (int i = NBLK - 1; i >= 0; --i) {
        Acc2 z0 = acc_zero();                                                  // R1: fc_1^T dn (sub-tile 0)
        dense_bf<PREC>(A, PD[0], z0);
        BF_SB();
        T2(36 + 6 * (NBLK - 1 - i) + 0);
        const WFragBF A0 = load_wfrag_bf<true, PREC>(Wb, 3 * i + 1, lane_off, lane_off_t);    // R2: prefetch fc_0^T
        Acc2 z1 = acc_zero();
        Pieces PH0, PH1;
        region_bf<PREC, 2>(A, PD[1], z1, sel, PH0, [&](int st, f32x8& x) { masked_stage(z0, mask_h[0][i], st, x); });
        T2(36 + 6 * (NBLK - 1 - i) + 1);
        Acc2 y0 = acc_zero();                                                  // R3: fc_0^T dh (sub-tile 0)
        region_bf<PREC, 2>(A0, PH0, y0, sel, PH1, [&](int st, f32x8& x) { masked_stage(z1, mask_h[1][i], st, x); });
        T2(36 + 6 * (NBLK - 1 - i) + 2);
        const WFragBF Ac = load_wfrag_bf<true, PREC>(Wb, 3 * i, lane_off, lane_off_t);        // R4: prefetch fc_c^T
        Acc2 y1 = acc_zero();
        region_bf<PREC, 4>(A0, PH1, y1, sel, PD[0], [&](int st, f32x8& x) { masked_add_stage(y0, mask_a[0][i], dn[0], st, x); });   // delta a_i
        T2(36 + 6 * (NBLK - 1 - i) + 3);
        // R5: dc += fc_c^T da (sub-tile 0)
        {
            Pieces PD1;
            region_bf<PREC, 4>(Ac, PD[0], dcc[0], sel, PD1, [&](int st, f32x8& x) { masked_add_stage(y1, mask_a[1][i], dn[1], st, x); });
            PD[1] = PD1;
        }
        T2(36 + 6 * (NBLK - 1 - i) + 4);
        if (i > 0) A = load_wfrag_bf<true, PREC>(Wb, 3 * i - 1, lane_off, lane_off_t);        // R6: prefetch fc_1[i-1]^T
        dense_bf<PREC>(Ac, PD[1], dcc[1]);
        BF_SB();
        T2(36 + 6 * (NBLK - 1 - i) + 5);
    }
    // ---- fc_p backward and d c / d u through the re-gathered taps (decoder_tile3's) -----------------------------------------
    BF_PRIO(IFD_BF_PRIO_EDGE);
    float g[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) g[t][0] = g[t][1] = g[t][2] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wp = *reinterpret_cast<const f32x4*>(Wf + BF_OFF_WP / 4 + (16 * mt + j) * 4 + q4 * 4);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float d = dn[t][4 * mt + j];
                g[t][0] = fmaf(wp.x, d, g[t][0]); g[t][1] = fmaf(wp.y, d, g[t][1]); g[t][2] = fmaf(wp.z, d, g[t][2]);
            }
        }
    if (IFD_BF_JAC && MODE == MODE_OPT) {
        // d c / d u from the parked Jacobian: g[a] += lk[a] * sum_ch J[a][ch] dc[ch]  (even / odd channel partial sums)
        const f32x8 dcj[2] = {flat(dcc[0]), flat(dcc[1])};
        const int z = opaque_zero();
        T2(66);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const f32x4 j0 = jpark[z + (t * 3 + a) * 2], j1 = jpark[z + (t * 3 + a) * 2 + 1];
                float pe = j0[0] * dcj[t][0], po = j0[1] * dcj[t][1];
                pe = __builtin_fmaf(j0[2], dcj[t][2], pe); po = __builtin_fmaf(j0[3], dcj[t][3], po);
                pe = __builtin_fmaf(j1[0], dcj[t][4], pe); po = __builtin_fmaf(j1[1], dcj[t][5], po);
                pe = __builtin_fmaf(j1[2], dcj[t][6], pe); po = __builtin_fmaf(j1[3], dcj[t][7], po);
                g[t][a] = fmaf(geo[t].lk[a], pe + po, g[t][a]);
            }
        T2(69);
    } else {
    f32x8 dcf[2] = {flat(dcc[0]), flat(dcc[1])};
    auto sample_bwd = [&](int P, const f32x4 (&tap)[2][4][2]) {
        const int a0 = AX0[P], a1 = AX1[P];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const SubGeo& gg = geo[t];
            // even / odd channel partial sums like decoder_tile3's packed form (pairs added at the end): the same roundings
            float pnw[2] = {0.f, 0.f}, pne[2] = {0.f, 0.f}, psw[2] = {0.f, 0.f}, pse[2] = {0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = dcf[t][4 * mt + j];
                    pnw[j & 1] = __builtin_fmaf(tap[t][0][mt][j], d, pnw[j & 1]);
                    pne[j & 1] = __builtin_fmaf(tap[t][1][mt][j], d, pne[j & 1]);
                    psw[j & 1] = __builtin_fmaf(tap[t][2][mt][j], d, psw[j & 1]);
                    pse[j & 1] = __builtin_fmaf(tap[t][3][mt][j], d, pse[j & 1]);
                }
            float pgx[2], pgy[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                pgx[h] = __builtin_fmaf(pse[h] - psw[h], gg.w1[a1], (pne[h] - pnw[h]) * gg.w0[a1]);
                pgy[h] = __builtin_fmaf(pse[h] - pne[h], gg.w1[a0], (psw[h] - pnw[h]) * gg.w0[a0]);
            }
            const float gix = pgx[0] + pgx[1], giy = pgy[0] + pgy[1];
            g[t][a0] = fmaf(gg.lk[a0], gix, g[t][a0]);
            g[t][a1] = fmaf(gg.lk[a1], giy, g[t][a1]);
        }
    };
    {
        f32x4 btap0[2][4][2], tap1[2][4][2], tap2[2][4][2];
        T2(66);
        load_taps(0, btap0, true);
        load_taps(1, tap1, true);
        __builtin_amdgcn_sched_barrier(0);
        T2(67);
#ifdef IFD_TRACE2
        if (tr != nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        T2(68);
#endif
        __builtin_amdgcn_s_waitcnt(0x4F70);      // (whole planes, as in the forward gather)
        __builtin_amdgcn_sched_barrier(0);
        sample_bwd(0, btap0);
        __builtin_amdgcn_sched_barrier(0);
        load_taps(2, tap2, true);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x4F70);
        __builtin_amdgcn_sched_barrier(0);
        sample_bwd(1, tap1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
        sample_bwd(2, tap2);
        __builtin_amdgcn_sched_barrier(0);
        T2(69);
    }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 3; ++a) dx[t][a] = add_lane_xor32(add_lane_xor16(g[t][a]));
    T2(70);
}
