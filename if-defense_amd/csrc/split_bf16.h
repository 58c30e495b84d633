// Exact three-piece bf16 splits of f32 values, with the residuals taken on the matrix pipe (the generic half of the split-precision
// tiles: tile_bf.h for the ConvONet decoder, onet_kernel.h for the ONet decoder).  Included inside namespace ifd.
//   x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)   (round to nearest even; 8 + 8 + 8 mantissa bits)
// Operand map of v_mfma_f32_16x16x32_bf16: lane (n = lane & 15, g = lane >> 4) holds A[m = n][k = 8 g + j], B[k = 8 g + j][n], j = 0 ... 7,
// and D[4 g + r][n].  A lane's eight values of a 32-channel group are the channels bf_chan(g, e) = 16 (e >> 2) + 4 g + (e & 3) of
// its point - two M-tiles of the accumulator layout - so the residual r = x - x1 is D = C + A B with C = x, B = the x1 piece and
// A = MINUS the selection matrix of the M-tile (exact: every product is 0 or -x1, and x1 is within 2^-9 of x).
// NOTE (gfx950 erratum, scripts/pk_mfma_coexec.hip): kernels that issue bf16 MFMAs must not contain packed-f32 vector instructions
// (v_pk_mul / fma / add_f32) - build their translation unit with -fno-slp-vectorize and write f32 vector arithmetic element by element.
#pragma once

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct Pieces {
    bf16x8 p[3];        // x1, x2, x3 of the lane's eight values: the B operands of the next layer
};
struct SelMat {
    bf16x8 m[2];        // per M-tile: -1 at the k-slot that holds the M-tile's row channel, 0 elsewhere
};

// lane (m, g), M-tile mt: element j is -1 iff bf_chan(g, j) == 16 mt + m, i.e. j = 4 mt + (m - 4 g) with 0 <= m - 4 g < 4
__device__ __forceinline__ SelMat make_selmat(int lane) {
    const int d = (lane & 15) - 4 * (lane >> 4);
    const unsigned int val = (d & ~3) == 0 ? (0xBF80u << (16 * (d & 1))) : 0u;       // bf16(-1.0) in the element's half of its dword
    const unsigned int v0 = (d >> 1) == 0 ? val : 0u, v1 = (d >> 1) == 1 ? val : 0u;
    typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
    SelMat s;
    s.m[0] = __builtin_bit_cast(bf16x8, u32x4s{v0, v1, 0u, 0u});
    s.m[1] = __builtin_bit_cast(bf16x8, u32x4s{0u, 0u, v0, v1});
    return s;
}

__device__ __forceinline__ bf16x8 cvt8_bf(const f32x8& v) {        // four v_cvt_pk_bf16_f32 (round to nearest even)
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
    return o;
}
__device__ __forceinline__ f32x4 mfma_bf(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// x -> x1, x2 (, x3); the residuals on the matrix pipe.  PREC 2 stops at two pieces.
template <int PREC>
__device__ __forceinline__ void split_bf(const f32x8& x, const SelMat& sel, Pieces& P) {
    P.p[0] = cvt8_bf(x);
    f32x4 c0 = {x[0], x[1], x[2], x[3]}, c1 = {x[4], x[5], x[6], x[7]};
    c0 = mfma_bf(sel.m[0], P.p[0], c0);
    c1 = mfma_bf(sel.m[1], P.p[0], c1);
    P.p[1] = cvt8_bf(f32x8{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]});
    if (PREC == 1) {
        c0 = mfma_bf(sel.m[0], P.p[1], c0);
        c1 = mfma_bf(sel.m[1], P.p[1], c1);
        P.p[2] = cvt8_bf(f32x8{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]});
    } else {
        P.p[2] = P.p[1];
    }
}
