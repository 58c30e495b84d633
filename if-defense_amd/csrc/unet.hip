// The encoder's shared 2-D U-Net (ConvONet/src/encoder/unet.py:225-239; depth 4, 32 start filters, concat
// merge, transpose up-conv) as hand-written implicit-GEMM convolutions on v_mfma_f32_32x32x2_f32.
//
//   layout    activations NHWC (channel-last) float32, image index = cloud * 3 + plane; weights re-packed on the
//             host to [tap][Cin][Cout].
//   block     4 waves; one wave = 32 output pixels x 32 output channels (16 accumulator VGPRs).  The block's
//             pixel tile (+1 halo) and the matching weight slab are staged through LDS in chunks of 16 input
//             channels; per chunk a wave issues 9 taps x 8 k-steps = 72 MFMAs.
//   mapping   M = pixel, N = output channel, K = (tap, input channel):  A[m = l&31][k = l>>5] from the LDS pixel
//             tile, B[k = l>>5][n = l&31] from the LDS weight slab, D: lane = channel n, register r = pixel
//             (r&3) + 8(r>>2) + 4(l>>5)  ->  NHWC stores are 128-byte rows, and the 2x2 max-pool partners of a
//             pixel live in the same lane (fused pooling epilogue).
//   fusions   bias + ReLU, max-pool (second output), channel concat of two inputs (skip connections),
//             ConvTranspose2d(k=2, s=2) as four 1x1 convolutions with a strided store.
// Results are independent of the batch size (fixed summation order) - unlike a library convolution whose
// algorithm choice depends on it - which the bitwise sharding-invariance of the whole path relies on.
#include "ifd_device.h"
#include "ifd_internal.h"

namespace ifd {

constexpr int CK = 16;               // input channels per LDS chunk
constexpr int CKP = CK + 1;          // padded pixel stride in LDS (floats)
#ifndef IFD_UNET_PW
#define IFD_UNET_PW 2                 // 32-pixel sub-tiles per wave of the 3x3 layers at 16^2 ... 64^2 (A/B: -DIFD_UNET_PW=1)
#endif

struct ConvArgs {
    const float* in0;      // first input  [N][H][W][C0]
    const float* in1;      // second input [N][H][W][C1] (concat along channels) or nullptr
    const float* w;        // [taps][C0 + C1][Cout]
    const float* bias;     // [Cout]
    float* out;            // [N][H][W][Cout]           (UP: [N][2H][2W][Cout])
    float* pool_out;       // [N][H/2][W/2][Cout] or nullptr
    int H, W, C0, C1, Cout;
    int relu;
    const float* fuse_w;   // FUSE: [Cout][32] weights of a 1x1 convolution applied to this layer's output (conv_final)
    const float* fuse_b;   // FUSE: its bias [32]; a.out then receives the 1x1 convolution's result
};

// TW: tile width in pixels (16 or 8); a wave covers 32/TW rows x TW columns.  PG x CG = 4 waves:
// PG pixel groups (stacked vertically) x CG groups of 32 output channels.  KS: 3 (pad 1) or 1.
// UP: ConvTranspose2d(k=2,s=2): the block computes all 4 (dy,dx) output taps of its input pixels - four accumulators per
// wave on one A operand (the input pixel), the input tile read once (HBM-bound layers: it was read once per tap).
// FUSE (Cout = 32, CG = 1): a following conv1x1 (32 -> 32, bias, no ReLU: conv_final, unet.py:238) is applied to the
// block's activated tile before it leaves the CU - the tile goes through LDS once to turn the accumulator layout
// (lane = channel) into the A-operand layout (lane = pixel), 16 more MFMAs per wave, and the intermediate tensor is
// neither written nor read back (the stand-alone 1x1 kernel was HBM-bound: 1.4 ms at 45 TFLOP/s).  Same MFMA sequence
// over the 32 channels as the stand-alone kernel -> bit-identical results.
// PW: 32-pixel sub-tiles per wave, stacked vertically (1 or 2).  With 2 a wave runs two independent accumulator chains on
// every B operand (weight) it reads from LDS and a block covers twice the pixels per staged weight slab and per barrier.
template <int TW, int PG, int CG, int KS, bool UP, bool FUSE = false, int PW = 1>
__global__ __launch_bounds__(256, (PW == 1 && !UP) ? 5 : 3) void conv_kernel(ConvArgs a) {   // PW 1: <= 96 VGPRs, five blocks (20 waves) per CU
    constexpr int RW = 32 / TW;                 // rows per 32-pixel sub-tile
    constexpr int TH = PG * PW * RW;            // tile height
    constexpr int HALO = KS / 2;
    constexpr int LW = TW + 2 * HALO, LH = TH + 2 * HALO;
    constexpr int TAPS = KS * KS;
    constexpr int WT = UP ? 4 : TAPS;           // weight taps staged per chunk
    constexpr int NACC = UP ? 4 : PW;           // accumulators per wave
    static_assert(!UP || (KS == 1 && PW == 1), "transpose convolution: 1x1 input footprint, one sub-tile");
    constexpr int NCO = 32 * CG;                // output channels per block
    __shared__ __attribute__((aligned(16))) float s_in[LH * LW * CKP];
    __shared__ __attribute__((aligned(16))) float s_w[WT * CK * NCO];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pg = wave % PG, cg = wave / PG;
    const int tiles_x = a.W / TW;
    const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile % tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    const int co_blocks = a.Cout / NCO;
    const int cob = (int)blockIdx.y;
    const int co0 = cob * NCO;
    const int n = blockIdx.z;
    const int Cin = a.C0 + a.C1;

    f32x16 acc[NACC];
#pragma unroll
    for (int pw = 0; pw < NACC; ++pw)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pw][r] = 0.f;

    const int m = lane & 31, kh = lane >> 5;
    const int py = pg * PW * RW + m / TW, px = m % TW;                 // pixel of this lane inside the tile (sub-tile 0)

    // Global -> register -> LDS staging, one chunk ahead: the loads of chunk c+1 are issued before the MFMAs of chunk c
    // and land in registers under them; after the barrier they are written to LDS.  (Round 1 staged inside the chunk,
    // one load per loop iteration with a full vmcnt(0) wait each: ~8 serial L2 round trips per chunk in front of 72 MFMAs.)
    // (Walking several tiles per block with the look-ahead running across the tile boundary was measured too: the
    // longer live ranges cost registers / occupancy, 82.5 ms against 74.0 ms for the whole U-Net.)
    constexpr int IN_N = LH * LW * (CK / 4), IN_IT = (IN_N + 255) / 256;      // f32x4 loads of the pixel tile per thread
    constexpr int W_N = WT * CK * (NCO / 4), W_IT = (W_N + 255) / 256;        // ... of the weight slab
    f32x4 rin[IN_IT], rw[W_IT];
    auto fetch = [&](int c0) {
        const float* src = c0 < a.C0 ? a.in0 : a.in1;
        const int cs = c0 < a.C0 ? a.C0 : a.C1;                        // channel count of the source tensor
        const int cl = c0 < a.C0 ? c0 : c0 - a.C0;                     // channel offset inside it
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {                           // (TH+2)x(TW+2) pixel tile, 16 channels, zero outside
            const int i = tid + it * 256;
            const int q4 = i % (CK / 4), pix = i / (CK / 4);
            const int ly = pix / LW, lx = pix % LW;
            const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
            rin[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((IN_N % 256 == 0 || i < IN_N) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                rin[it] = *reinterpret_cast<const f32x4*>(src + (((size_t)n * a.H + gy) * a.W + gx) * cs + cl + 4 * q4);
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {                            // weight slab [taps][16][NCO]
            const int i = tid + it * 256;
            const int q4 = i % (NCO / 4), row = i / (NCO / 4);         // row = tap * CK + k
            const int tap = row / CK, k = row % CK;
            if (W_N % 256 == 0 || i < W_N)
                rw[it] = *reinterpret_cast<const f32x4*>(a.w + ((size_t)tap * Cin + c0 + k) * a.Cout + co0 + 4 * q4);
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < Cin; c0 += CK) {
        __syncthreads();                                               // the previous chunk's MFMAs have read the LDS tiles
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int i = tid + it * 256;
            if (IN_N % 256 == 0 || i < IN_N) {
                float* d = s_in + (i / (CK / 4)) * CKP + 4 * (i % (CK / 4));
                d[0] = rin[it].x; d[1] = rin[it].y; d[2] = rin[it].z; d[3] = rin[it].w;
            }
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * 256;
            if (W_N % 256 == 0 || i < W_N) *reinterpret_cast<f32x4*>(s_w + 4 * i) = rw[it];
        }
        __syncthreads();
        if (c0 + CK < Cin) fetch(c0 + CK);
        if constexpr (UP) {
            const float* ap = s_in + (py * LW + px) * CKP + kh;
            const float* bp = s_w + kh * NCO + cg * 32 + m;
#pragma unroll
            for (int ks = 0; ks < CK / 2; ++ks) {
                const float av = ap[2 * ks];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[(t * CK + 2 * ks) * NCO], acc[t], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int dy = tap / KS, dx = tap % KS;
                const float* ap = s_in + ((py + dy) * LW + px + dx) * CKP + kh;
                const float* bp = s_w + (tap * CK + kh) * NCO + cg * 32 + m;
#pragma unroll
                for (int ks = 0; ks < CK / 2; ++ks) {
                    const float b = bp[2 * ks * NCO];
#pragma unroll
                    for (int pw = 0; pw < PW; ++pw)
                        acc[pw] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[pw * RW * LW * CKP + 2 * ks], b, acc[pw], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: bias, ReLU, NHWC store (lane = output channel), optional fused 2x2 max-pool --------------
    const int co = co0 + cg * 32 + m;
    const float bv = a.bias[co];
    if constexpr (FUSE) __syncthreads();                               // every wave is done with the weight slab
#pragma unroll
    for (int pw = 0; pw < NACC; ++pw) {                                // sub-tile (UP: output tap dy * 2 + dx) of this wave
        const int yw = y0 + (UP ? pg : pg * PW + pw) * RW;             // first row of this 32-pixel sub-tile
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = acc[pw][r] + bv;
            if (a.relu) v[r] = fmaxf(v[r], 0.f);
        }
        if constexpr (FUSE) {
            static_assert(CG == 1 && !UP && TAPS * CK * NCO >= 128 * 33, "FUSE: one 32-channel group, tile fits the weight slab");
            float* T = s_w + wave * 32 * 33;                           // [32 pixels of this wave][33], wave-private rows
#pragma unroll
            for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * kh) * 33 + m] = v[r];
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
            const float* fw = a.fuse_w + kh * 32 + m;                  // B[k = channel][n = output channel]
            float bw[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) bw[ks] = fw[2 * ks * 32];
            const float* tp = T + m * 33 + kh;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(tp[2 * ks], bw[ks], acc2, 0, 0, 0);
            const float fb = a.fuse_b[m];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc2[r] + fb;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = (r & 3) + 8 * (r >> 2) + 4 * kh;           // pixel index inside the sub-tile
            const int oy = yw + mm / TW, ox = x0 + mm % TW;
            if (UP) {
                const int uy = 2 * oy + (pw >> 1), ux = 2 * ox + (pw & 1);
                a.out[(((size_t)n * 2 * a.H + uy) * 2 * a.W + ux) * a.Cout + co] = v[r];
            } else {
                a.out[(((size_t)n * a.H + oy) * a.W + ox) * a.Cout + co] = v[r];
            }
        }
        if (!UP && a.pool_out != nullptr) {
            // partners of pixel mm: mm+1 (register r+1) and mm+TW (TW=16: r+8, TW=8: r+4) - all in this lane
            constexpr int RSTEP = TW == 16 ? 8 : 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool top_left = (r & 1) == 0 && ((r / RSTEP) & 1) == 0;
                if (top_left) {
                    const int mm = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int oy = yw + mm / TW, ox = x0 + mm % TW;
                    const float pv = fmaxf(fmaxf(v[r], v[r + 1]), fmaxf(v[r + RSTEP], v[r + RSTEP + 1]));
                    a.pool_out[(((size_t)n * (a.H / 2) + oy / 2) * (a.W / 2) + ox / 2) * a.Cout + co] = pv;
                }
            }
        }
    }
}

template <int TW, int PG, int CG, int KS, bool UP, bool FUSE = false, int PW = 1>
static hipError_t launch_conv(const ConvArgs& a, int n_img, hipStream_t s) {
    constexpr int TH = PG * PW * (32 / TW);
    const dim3 grid((a.H / TH) * (a.W / TW), a.Cout / (32 * CG), n_img);
    hipLaunchKernelGGL((conv_kernel<TW, PG, CG, KS, UP, FUSE, PW>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// conv3x3 / conv1x1 / transpose-conv dispatch on the image size (64/32/16 -> 8x16 pixel tiles, 8 -> 8x8 tiles)
static hipError_t conv(const float* in0, int C0, const float* in1, int C1, const float* w, const float* b, float* out,
                       float* pool_out, int HW, int Cout, int ks, bool up, bool relu, int n_img, hipStream_t s,
                       const float* fuse_w = nullptr, const float* fuse_b = nullptr) {
    ConvArgs a{in0, in1, w, b, out, pool_out, HW, HW, C0, C1, Cout, relu ? 1 : 0, fuse_w, fuse_b};
    if (fuse_w != nullptr) {
        if (HW < 16 || ks != 3 || up || Cout != 32) return hipErrorInvalidValue;
        return launch_conv<16, 4, 1, 3, false, true, IFD_UNET_PW>(a, n_img, s);
    }
    if (HW >= 16) {
        if (up) return launch_conv<16, 4, 1, 1, true>(a, n_img, s);
        if (ks == 3) return launch_conv<16, 4, 1, 3, false, false, IFD_UNET_PW>(a, n_img, s);
        return launch_conv<16, 4, 1, 1, false>(a, n_img, s);
    }
    if (up) return launch_conv<8, 2, 2, 1, true>(a, n_img, s);
    if (ks == 3) return launch_conv<8, 2, 2, 3, false>(a, n_img, s);
    return launch_conv<8, 2, 2, 1, false>(a, n_img, s);
}

size_t unet_workspace_floats(int n_img) {
    // d0a d0 u2up u2a u2 (64^2x32) | d1a d1 u1up u1a u1 (32^2x64) | p1 (32^2x32) | d2a d2 u0up u0a u0 (16^2x128)
    // | p2 (16^2x64) | d3a d3 (8^2x256) | p3 (8^2x128)
    return (size_t)n_img * (5 * 131072 + 5 * 65536 + 32768 + 5 * 32768 + 16384 + 2 * 16384 + 8192);
}

// x [n_img][64][64][32] -> out [n_img][64][64][32]; wd = device copy of the re-packed U-Net weights.
hipError_t launch_unet(const UNetWeights& W, const float* x, float* out, float* ws, int n_img, hipStream_t s) {
    size_t o = 0;
    auto take = [&](size_t per_img) { float* p = ws + o; o += per_img * (size_t)n_img; return p; };
    float *d0a = take(131072), *d0 = take(131072), *u2up = take(131072), *u2a = take(131072), *u2 = take(131072);
    float *d1a = take(65536), *d1 = take(65536), *u1up = take(65536), *u1a = take(65536), *u1 = take(65536);
    float* p1 = take(32768);
    float *d2a = take(32768), *d2 = take(32768), *u0up = take(32768), *u0a = take(32768), *u0 = take(32768);
    float* p2 = take(16384);
    float *d3a = take(16384), *d3 = take(16384);
    float* p3 = take(8192);
    (void)u2;                                // conv_final is fused into the layer that produced u2; the slot stays in the layout
    hipError_t e;
#define IFD_TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
    // encoder pathway (DownConv, unet.py:66-72): conv-relu, conv-relu, pool (not on the last level)
    IFD_TRY(conv(x, 32, nullptr, 0, W.down_w[0][0], W.down_b[0][0], d0a, nullptr, 64, 32, 3, false, true, n_img, s));
    IFD_TRY(conv(d0a, 32, nullptr, 0, W.down_w[0][1], W.down_b[0][1], d0, p1, 64, 32, 3, false, true, n_img, s));
    IFD_TRY(conv(p1, 32, nullptr, 0, W.down_w[1][0], W.down_b[1][0], d1a, nullptr, 32, 64, 3, false, true, n_img, s));
    IFD_TRY(conv(d1a, 64, nullptr, 0, W.down_w[1][1], W.down_b[1][1], d1, p2, 32, 64, 3, false, true, n_img, s));
    IFD_TRY(conv(p2, 64, nullptr, 0, W.down_w[2][0], W.down_b[2][0], d2a, nullptr, 16, 128, 3, false, true, n_img, s));
    IFD_TRY(conv(d2a, 128, nullptr, 0, W.down_w[2][1], W.down_b[2][1], d2, p3, 16, 128, 3, false, true, n_img, s));
    IFD_TRY(conv(p3, 128, nullptr, 0, W.down_w[3][0], W.down_b[3][0], d3a, nullptr, 8, 256, 3, false, true, n_img, s));
    IFD_TRY(conv(d3a, 256, nullptr, 0, W.down_w[3][1], W.down_b[3][1], d3, nullptr, 8, 256, 3, false, true, n_img, s));
    // decoder pathway (UpConv, unet.py:101-114): upconv, cat(up, skip), conv-relu, conv-relu
    IFD_TRY(conv(d3, 256, nullptr, 0, W.up_t_w[0], W.up_t_b[0], u0up, nullptr, 8, 128, 1, true, false, n_img, s));
    IFD_TRY(conv(u0up, 128, d2, 128, W.up_w[0][0], W.up_b[0][0], u0a, nullptr, 16, 128, 3, false, true, n_img, s));
    IFD_TRY(conv(u0a, 128, nullptr, 0, W.up_w[0][1], W.up_b[0][1], u0, nullptr, 16, 128, 3, false, true, n_img, s));
    IFD_TRY(conv(u0, 128, nullptr, 0, W.up_t_w[1], W.up_t_b[1], u1up, nullptr, 16, 64, 1, true, false, n_img, s));
    IFD_TRY(conv(u1up, 64, d1, 64, W.up_w[1][0], W.up_b[1][0], u1a, nullptr, 32, 64, 3, false, true, n_img, s));
    IFD_TRY(conv(u1a, 64, nullptr, 0, W.up_w[1][1], W.up_b[1][1], u1, nullptr, 32, 64, 3, false, true, n_img, s));
    IFD_TRY(conv(u1, 64, nullptr, 0, W.up_t_w[2], W.up_t_b[2], u2up, nullptr, 32, 32, 1, true, false, n_img, s));
    IFD_TRY(conv(u2up, 32, d0, 32, W.up_w[2][0], W.up_b[2][0], u2a, nullptr, 64, 32, 3, false, true, n_img, s));
    // last conv-relu with conv_final (conv1x1, unet.py:238) fused into its epilogue
    IFD_TRY(conv(u2a, 32, nullptr, 0, W.up_w[2][1], W.up_b[2][1], out, nullptr, 64, 32, 3, false, true, n_img, s, W.fin_w,
                 W.fin_b));
#undef IFD_TRY
    return hipSuccess;
}

}  // namespace ifd
