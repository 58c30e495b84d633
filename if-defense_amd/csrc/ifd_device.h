// Shared device-side definitions for libifd (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ifd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RES = 64;              // plane resolution
constexpr int CH = 32;               // c_dim == hidden
constexpr int NBLK = 5;
constexpr int PLANE_FLOATS = RES * RES * CH;        // one plane, channel-last
constexpr int CLOUD_PLANE_FLOATS = 3 * PLANE_FLOATS; // 1.5 MiB / cloud

// ---- decoder parameter image (identical in global memory and in LDS) -------------
// W[15][32][34]: layer L = 3*i + j (j: 0 fc_c[i], 1 blocks[i].fc_0, 2 blocks[i].fc_1), row = out
// channel o, column = wperm(in channel c), row stride 34 floats.  The 16x16x4 MFMA operand mapping of optimize.hip
// (lane = (n, q), A[m = n][k = q]) reads the A operands with ds_read(2)_b32: 32 banks, the lanes 0-31 (q = 0, 1) and
// 32-63 (q = 2, 3) are served together.
//   forward  gather W[16 mt + n][16 mt' + 4 q + r']:  bank (2 n + wperm(4 q) + const) mod 32 - wperm(4 q) = 0, 1, 16, 17:
//            the two q of a group differ in parity, 2 n covers the even banks: conflict-free;
//   backward gather W[16 mt' + 4 q + r'][16 mt + n]:  bank (8 q + wperm(n) + const) mod 32 - wperm maps the columns
//            0-15 onto {0-7, 16-23} (16-31 onto {8-15, 24-31}), a set disjoint from itself shifted by 8: conflict-free.
// (Rounds 1-2 kept the columns of a 16-block together, wperm(c) = (c & 16) + 4 (c & 3) + ((c >> 2) & 3): the backward gather
// was then 2-way - banks 8 q + [0, 16) - and owned 72 % of the kernel's SQ_LDS_BANK_CONFLICT cycles, 488 per 32-point tile;
// scripts/pmc_lds_attrib.sh.)
constexpr int W_STRIDE = 34;
constexpr int W_LAYER = 32 * W_STRIDE;               // 1088
#ifdef IFD_WPERM_OLD
__host__ __device__ constexpr int wperm(int c) { return (c & 16) + 4 * (c & 3) + ((c >> 2) & 3); }
#else
__host__ __device__ constexpr int wperm(int c) { return 8 * ((c >> 4) & 1) + ((c >> 2) & 1) + 2 * (c & 3) + 16 * ((c >> 3) & 1); }
#endif
constexpr int DEC_OFF_W = 0;
constexpr int DEC_OFF_BIAS = 15 * W_LAYER;           // 16320: [15][32]
constexpr int DEC_OFF_WP = DEC_OFF_BIAS + 15 * 32;   // [32][4] = {Wp[ch][0..2], bp[ch]}
constexpr int DEC_OFF_WOUT = DEC_OFF_WP + 32 * 4;    // [32]
constexpr int DEC_OFF_BOUT = DEC_OFF_WOUT + 32;      // [1] (+3 pad)
constexpr int DEC_FLOATS = DEC_OFF_BOUT + 4;         // 16964 floats = 67,856 B

// ---- decoder parameter image of the split-precision tiles (tile_bf.h) --------------------------------------
// The 15 layers as bf16 PIECES in the operand order of v_mfma_f32_16x16x32_bf16: w = w1 + w2 + w3 with w1 = bf16(w),
// w2 = bf16(w - w1), w3 = bf16(w - w1 - w2) (round to nearest even; the three pieces carry the 24 mantissa bits exactly).
//   [layer 15][piece 3][M-tile 2][lane 64][8 bf16]: lane (m = lane & 15, g = lane >> 4) of M-tile mt holds, for j = 0 ... 7,
//   W[16 mt + m][bf_chan(g, j)] - one ds_read_b128 per A operand; element j of the B operand of lane group g is the accumulator
//   register j of the layer before (channel bf_chan(g, j)): activations never leave registers, like in the f32 tile.
// The transposed operand of the backward pass comes out of the SAME image through ds_read_b64_tr_b16 (tile_bf.h load_wfrag_bf).
// Behind the layers, in f32: the biases [15][32] (fc_c's folded like in the f32 image), fc_p [32][4], fc_out [32], {fc_out's bias, 1 / B}.
__host__ __device__ constexpr int bf_chan(int g, int j) { return 16 * (j >> 2) + 4 * g + (j & 3); }
constexpr int BF_ENTRY_BYTES = 16;
constexpr int BF_PIECE_BYTES = 2 * 64 * BF_ENTRY_BYTES;           // 2048: both M-tiles of one piece
constexpr int BF_LAYER_BYTES = 3 * BF_PIECE_BYTES;               // 6144
constexpr int BF_OFF_BIAS = 15 * BF_LAYER_BYTES;                 // 92,160: float [15][32]
constexpr int BF_OFF_WP = BF_OFF_BIAS + 15 * 32 * 4;             // float [32][4]
constexpr int BF_OFF_WOUT = BF_OFF_WP + 32 * 4 * 4;              // float [32]
constexpr int BF_OFF_BOUT = BF_OFF_WOUT + 32 * 4;                // float [4]: fc_out's bias, 1 / B (written by the kernel), pad
constexpr int BF_IMG_BYTES = BF_OFF_BOUT + 16;                   // 94,736 B

// ---- packed ReLU of the optimiser tile (optimize.hip decoder_tile3) ---------------------------------------------------
// gfx950 has no packed f32 max, but VOP3P float instructions take the clamp modifier: v_pk_mul_f32 a, 2^-K clamp gives
// relu(a) * 2^-K for TWO values in one instruction (for a < 2^K; the clamp's upper end), half the vector instructions of a
// v_max per value.  The factor is undone for free: the optimiser kernel's copy of the parameter image holds fc_0 / fc_1 / fc_out
// multiplied by 2^K (exact), so every product w * relu(a) has the same bits as before; in the backward pass the transposed
// fc_1 / fc_0 products come out 2^K / 2^2K too large and the residual add that follows them (dn += mask * y) is a packed
// fma with 2^-2K instead of a packed add - same instruction count, same rounding (one rounding of the same real number).
// K = 40: activations must stay below 1.1e12 (they are O(1) ... O(1e3)), backward intermediates below 2^127 / 2^80 = 1.4e14.
#ifndef IFD_RELU_K
#define IFD_RELU_K 0                  // 0: one v_max per value, plain image (default); 40: the packed form (measured -0.3 %)
#endif
constexpr int RELU_K = IFD_RELU_K;
// (only ever used to initialise the constexpr variables below: called with a run-time context the recursion would become
// a real recursive device function - stack in scratch, s_swappc - which is what a first version of this did)
__host__ __device__ constexpr float pow2i(int e) { return e == 0 ? 1.f : (e > 0 ? 2.f * pow2i(e - 1) : 0.5f * pow2i(e + 1)); }
constexpr float RELU_UP = pow2i(RELU_K);            // 2^K:   factor on fc_0 / fc_1 / fc_out in the optimiser's image
constexpr float RELU_DN = pow2i(-RELU_K);           // 2^-K:  the packed ReLU's multiplier
constexpr float RELU_DN2 = pow2i(-2 * RELU_K);      // 2^-2K: undoes the two transposed products of a block in the backward pass

constexpr int MAXK = 1024;           // points per cloud held in LDS (persistent optimiser kernel)
constexpr int LARGE_MAXK = 10000;    // largest cloud of the launch-per-step path (optimize.hip, "large" section)
constexpr int LARGE_LDS_MAXK = 4096; // ... up to here its repulsion accumulators sit in LDS, above in global memory
constexpr int OPT_THREADS = 512;     // 8 waves: 2 per SIMD (256 VGPRs each), two points per thread
constexpr float FIX_SCALE = 1099511627776.0f;        // 2^40: fixed-point scale of the neighbour scatter
constexpr float FIX_INV = 1.0f / 1099511627776.0f;

// ---- ONet-Opt variant (onet.hip): shipped config ONet/configs/onet_mn40.yaml ------------------------------
constexpr int ONET_H = 256;          // decoder hidden size (onet/models/decoder.py:89)
constexpr int ONET_C = 512;          // latent code (onet_mn40.yaml:18)
constexpr int ONET_ENC_H = 512;      // encoder hidden size (onet_mn40.yaml:16-17)
constexpr int ONET_NCBN = 11;        // bn_0 / bn_1 of 5 blocks + the final bn

// x + (x of lane ^ 16) and x + (x of lane ^ 32) through gfx950's row / half swaps (v_permlane16_swap, v_permlane32_swap:
// one vector instruction, no LDS crossbar round trip like the ds_bpermute behind __shfl_xor).  Same pairs, and the float
// sum commutes: bit-identical to the shuffle form.
// [pcsamp:tile.lane_reduce]
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float add_lane_xor16(float x) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float add_lane_xor32(float x) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

struct DecConst {
    float sdiv;      // (float)(1 + padding + 10e-6)           common.py:250
    float uclamp;    // (float)(1 - 10e-6)                     common.py:255
};

}  // namespace ifd
