// Pre-processing in front of the encoder / optimiser, one workgroup per cloud:
//   sor_kernel      SORDefense.outlier_removal (ConvONet/defense/SOR.py:22-49): float64 expanded-form squared
//                   distances, the 2 nearest non-first neighbours' mean, keep value <= mean + alpha * std.
//   prepare_kernel  preprocess_pc (ConvONet/opt_defense.py:114-146: centre, scale by the largest bbox extent,
//                   * padding_scale, random 600-subset without replacement) and init_points (:149-179: 1024
//                   indices with replacement, + N(0, sigma^2) noise, clamp to +-0.5*padding_scale).
// The reference draws from unseeded global RNGs; here every draw is a pure function of
// (seed, global cloud index, draw index) via Philox-4x32-10, so results do not depend on how clouds are
// batched or sharded over GPUs.  For parity tests the draws can be passed in explicitly.
#include "ifd_device.h"
#include "ifd_internal.h"

namespace ifd {

constexpr int PREP_THREADS = 1024;
// PREP_MAXK (ifd_internal.h) = 10,000: the largest input cloud these kernels accept (the LDS of prepare_kernel: 16 bytes per
// point).  Up to SOR_NARROW_MAXK points sor_kernel keeps the cloud in LDS in double (32 bytes per point); above, in float
// with the doubles re-made per pair (12 bytes per point) - same values, see sor_kernel.
constexpr int SOR_NARROW_MAXK = 4096;
constexpr int PREP_SORT_MAX = 4096;      // up to this many optimised points per cloud leave prepare_kernel in Morton order (rank by counting: O(n^2))
// entries of prepare_kernel's key array: the K subset keys, or the n_opt Morton keys of the optimised points if there are more of
// them (rounded so that the doubles behind the array stay 8-byte aligned: 12 K + 4 nkey bytes in front of them)
__host__ __device__ inline int prep_nkey(int K, int n_opt_sorted) {
    return n_opt_sorted > K ? n_opt_sorted + ((3 * K + n_opt_sorted) & 1) : K;
}

// ---- Philox-4x32-10 (Salmon et al., SC'11) ---------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)

// deterministic block reductions (fixed order), doubles
__device__ __forceinline__ double block_sum_d(double v, double* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += scratch[w];
    return s;
}
__device__ __forceinline__ float block_minmax_f(float v, bool is_max, float* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, u) : fminf(v, u);
    }
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float s = scratch[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s = is_max ? fmaxf(s, scratch[w]) : fminf(s, scratch[w]);
    return s;
}

// ---------------------------------------------------------------------------------------------------
// WIDE (K > SOR_NARROW_MAXK): the cloud sits in LDS as floats and every pair converts its candidate and squares it again.
// The values are the same as the narrow layout's: a float converts to double exactly, a product of two such doubles is
// exact (48 significant bits), so |x|^2 = (x x + y y) + z z has the same two roundings however it is contracted.
// NB: sorted candidates kept per point: k_nn + 1 of them are used.  The shipped k = 2 runs with NB = 3 - with eight, a candidate
// entered the insertion network whenever it beat the EIGHTH best (three times as often as the third, and on a wave of 64 lanes
// nearly every iteration) and bubbled through eight levels instead of three: 3.03 -> 1.13 ms per 2468 clouds, same values.
template <bool WIDE, int NB = 8>
__global__ __launch_bounds__(PREP_THREADS) void sor_kernel(const float* __restrict__ pc, int K, int k_nn, double alpha,
                                                            uint8_t* __restrict__ keep, double* __restrict__ value_out) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    constexpr int PPT = WIDE ? (PREP_MAXK + PREP_THREADS - 1) / PREP_THREADS : SOR_NARROW_MAXK / PREP_THREADS;     // 10 : 4
    double* scratch = dsm;                // [16]
    double* X = dsm + 16;                 // narrow: [K][3] doubles, then XX [K] = |x|^2 (the launch sizes the LDS by K)
    double* XX = X + 3 * K;
    float* Xf = reinterpret_cast<float*>(dsm + 16);       // wide: [K][3] floats
    double val[PPT];                      // value of point tid + 1024 r (registers)
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = pc + (size_t)b * K * 3;
    for (int i = tid; i < K; i += PREP_THREADS) {
        if (WIDE) {
            Xf[3 * i] = p[3 * i]; Xf[3 * i + 1] = p[3 * i + 1]; Xf[3 * i + 2] = p[3 * i + 2];
        } else {
            const double x = (double)p[3 * i], y = (double)p[3 * i + 1], z = (double)p[3 * i + 2];
            X[3 * i] = x; X[3 * i + 1] = y; X[3 * i + 2] = z;
            XX[i] = x * x + y * y + z * z;
        }
    }
    __syncthreads();
    double vsum = 0.0;
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        const int i = tid + r * PREP_THREADS;
        val[r] = 0.0;
        if (i >= K) continue;
        double x, y, z, xx;
        if (WIDE) {
            x = (double)Xf[3 * i]; y = (double)Xf[3 * i + 1]; z = (double)Xf[3 * i + 2];
            xx = x * x + y * y + z * z;
        } else {
            x = X[3 * i]; y = X[3 * i + 1]; z = X[3 * i + 2]; xx = XX[i];
        }
        // k_nn + 1 smallest of dist[i][j] = xx_j + (-2 x_i.x_j) + xx_i over ALL j (self included, as the reference);
        // k_nn <= 7
        double best[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) best[q] = INFINITY;
        for (int j = 0; j < K; ++j) {
            double xj, yj, zj, xxj;
            if (WIDE) {
                xj = (double)Xf[3 * j]; yj = (double)Xf[3 * j + 1]; zj = (double)Xf[3 * j + 2];
                xxj = xj * xj + yj * yj + zj * zj;
            } else {
                xj = X[3 * j]; yj = X[3 * j + 1]; zj = X[3 * j + 2]; xxj = XX[j];
            }
            const double inner = -2.0 * (x * xj + y * yj + z * zj);
            double d = (xxj + inner) + xx;
            if (d < best[NB - 1]) {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const bool c = d < best[q];
                    const double lo = c ? d : best[q];
                    d = c ? best[q] : d;
                    best[q] = lo;
                }
            }
        }
        double v = 0.0;                    // mean of neighbours 1..k_nn (the smallest, "self", is dropped)
#pragma unroll
        for (int q = 1; q < NB; ++q) v += q <= k_nn ? best[q] : 0.0;
        v /= (double)k_nn;
        val[r] = v;
        vsum += v;
        if (value_out) value_out[(size_t)b * K + i] = v;
    }
    const double mean = block_sum_d(vsum, scratch) / (double)K;
    double sq = 0.0;
#pragma unroll
    for (int r = 0; r < PPT; ++r)
        if (tid + r * PREP_THREADS < K) { const double d = val[r] - mean; sq += d * d; }
    const double var = block_sum_d(sq, scratch) / (double)(K - 1);       // torch.std: unbiased
    const double thr = mean + alpha * sqrt(var);
#pragma unroll
    for (int r = 0; r < PPT; ++r)
        if (tid + r * PREP_THREADS < K) keep[(size_t)b * K + tid + r * PREP_THREADS] = val[r] <= thr ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PREP_THREADS) void prepare_kernel(const float* __restrict__ pc, const uint8_t* __restrict__ keep,
                                                                int K, PrepArgs A, const int32_t* __restrict__ sel_idx,
                                                                const int32_t* __restrict__ init_idx,
                                                                const float* __restrict__ noise, float* __restrict__ sel,
                                                                int32_t* __restrict__ t_per_cloud, float* __restrict__ init,
                                                                int32_t* __restrict__ n_kept, float* __restrict__ proc_out) {
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    // the optimised points leave in Morton order when the draws are the library's own (see the end of the kernel)
    const bool morton = init_idx == nullptr && noise == nullptr && A.n_opt <= PREP_SORT_MAX && A.no_morton == 0;
    const int nkey = prep_nkey(K, morton ? A.n_opt : 0);
    float* P = fsm;                                                   // [K][3] kept points, then processed
    uint32_t* KEY = reinterpret_cast<uint32_t*>(P + 3 * K);           // [K] random keys for the subset ...
    int* POS = reinterpret_cast<int*>(KEY);                           // ... in the place of the compaction prefix (done by then)
    float* scratch = reinterpret_cast<float*>(KEY + nkey);            // [64]  (8-byte aligned: holds doubles)
    int* s_n = reinterpret_cast<int*>(scratch + 62);
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint32_t gcloud = (uint32_t)(A.cloud_base + b);
    const float* p = pc + (size_t)b * K * 3;
    const uint8_t* km = keep ? keep + (size_t)b * K : nullptr;

    // ---- stable compaction of the kept points (boolean-mask indexing keeps the original order) --------
    // (exclusive prefix sum of the keep flags, 1024 points at a time: ballot + popcount inside a wave, the 16 wave totals through
    // LDS.  Rounds 1-3 had thread 0 walk the mask: 1024 dependent global byte loads and LDS stores in front of everything else.)
    {
        int* wtot = reinterpret_cast<int*>(scratch);                  // [16] wave totals of the current chunk
        int base = 0;
        for (int c0 = 0; c0 < K; c0 += PREP_THREADS) {
            const int i = c0 + tid;
            const bool kept = i < K && (!km || km[i]);
            const unsigned long long m = __ballot(kept);
            const int lane = tid & 63, wave = tid >> 6;
            const int before = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wtot[wave] = __popcll(m);
            __syncthreads();
            int woff = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < PREP_THREADS / 64; ++w) { const int t = wtot[w]; woff += w < wave ? t : 0; tot += t; }
            if (i < K) POS[i] = base + woff + before;
            base += tot;
            __syncthreads();
        }
        if (tid == 0) *s_n = base;
    }
    __syncthreads();
    const int n = *s_n;
    __syncthreads();
    for (int i = tid; i < K; i += PREP_THREADS)
        if (!km || km[i]) { P[3 * POS[i]] = p[3 * i]; P[3 * POS[i] + 1] = p[3 * i + 1]; P[3 * POS[i] + 2] = p[3 * i + 2]; }
    __syncthreads();
    if (tid == 0 && n_kept) n_kept[b] = n;

    // ---- centre, scale (opt_defense.py:122-127) --------------------------------------------------------
    double* dscr = reinterpret_cast<double*>(scratch);
    float cen[3], ext = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double s = 0.0;
        for (int i = tid; i < n; i += PREP_THREADS) s += (double)P[3 * i + a];
        cen[a] = (float)(block_sum_d(s, dscr) / (double)n);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float mx = -INFINITY, mn = INFINITY;
        for (int i = tid; i < n; i += PREP_THREADS) { const float v = P[3 * i + a] - cen[a]; mx = fmaxf(mx, v); mn = fminf(mn, v); }
        mx = block_minmax_f(mx, true, scratch);
        mn = block_minmax_f(mn, false, scratch);
        ext = fmaxf(ext, mx - mn);
    }
    __syncthreads();
    for (int i = tid; i < n; i += PREP_THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) P[3 * i + a] = (P[3 * i + a] - cen[a]) / ext * A.padding_scale;
    }
    __syncthreads();
    if (proc_out)
        for (int i = tid; i < n * 3; i += PREP_THREADS) proc_out[(size_t)b * K * 3 + i] = P[i];

    // ---- encoder subset: T points without replacement, in random order (np.random.choice(replace=False)) -
    const int T = A.n_sel;
    float* so = sel + (size_t)b * T * 3;
    int tcount;
    if (n > T) {
        tcount = T;
        if (sel_idx) {
            for (int t = tid; t < T; t += PREP_THREADS) {
                const int j = sel_idx[(size_t)b * T + t];
                so[3 * t] = P[3 * j]; so[3 * t + 1] = P[3 * j + 1]; so[3 * t + 2] = P[3 * j + 2];
            }
        } else {
            for (int i = tid; i < n; i += PREP_THREADS) KEY[i] = philox(gcloud, (uint32_t)i, 1u, 0u, A.seed_lo, A.seed_hi).x;
            __syncthreads();
            for (int i = tid; i < n; i += PREP_THREADS) {
                const uint32_t ki = KEY[i];
                int rank = 0;                                  // position of i in the key-sorted order (ties by index)
                for (int j = 0; j < n; ++j) { const uint32_t kj = KEY[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
                if (rank < T) { so[3 * rank] = P[3 * i]; so[3 * rank + 1] = P[3 * i + 1]; so[3 * rank + 2] = P[3 * i + 2]; }
            }
        }
    } else {
        tcount = n;                                            // fewer points than the encoder subset: use them all
        for (int t = tid; t < T; t += PREP_THREADS) {
            const bool v = t < n;
            so[3 * t] = v ? P[3 * t] : 0.f; so[3 * t + 1] = v ? P[3 * t + 1] : 0.f; so[3 * t + 2] = v ? P[3 * t + 2] : 0.f;
        }
    }
    if (tid == 0) t_per_cloud[b] = tcount;

    // ---- init_points: indices with replacement + gaussian noise, clamped ---------------------------------
    // The reference's draws (torch.randint + torch.randn, opt_defense.py:149-179) are i.i.d.: the ORDER of the n_opt optimised points
    // carries no information.  With the library's own draws the points are therefore written in MORTON ORDER of their coordinates
    // (30-bit Z-curve, ties by draw index): the 32 points of a decoder tile are then neighbours on the surface, the 12 x 32 bilinear
    // taps of a tile fall into a few rows of each feature plane instead of all over it, and the optimiser's gathers - its HBM-side
    // stream - run 0.8 % (f32) / 4.9 % (bf16x6) faster (profiles/r06_ab_bf_locality.txt).  Explicit draws (init_idx / noise: the
    // parity tests' recorded draws) keep their order.
    const float lim = 0.5f * A.padding_scale;
    float* io = init + (size_t)b * A.n_opt * 3;
    auto draw = [&](int t, float (&o)[3]) {
        int j;
        float g0, g1, g2;
        if (init_idx) {
            j = init_idx[(size_t)b * A.n_opt + t];
        } else {
            const U4 r = philox(gcloud, (uint32_t)t, 2u, 0u, A.seed_lo, A.seed_hi);
            j = (int)(((uint64_t)r.x * (uint64_t)n) >> 32);    // uniform in [0, n)
        }
        if (noise) {
            const float* nz = noise + ((size_t)b * A.n_opt + t) * 3;
            g0 = nz[0]; g1 = nz[1]; g2 = nz[2];
        } else {
            const U4 r = philox(gcloud, (uint32_t)t, 3u, 0u, A.seed_lo, A.seed_hi);
            const float ra = sqrtf(-2.f * logf(u01(r.x))), rb = sqrtf(-2.f * logf(u01(r.z)));
            float s0, c0, s1, c1;
            sincosf(6.28318530717958647692f * u01(r.y), &s0, &c0);
            sincosf(6.28318530717958647692f * u01(r.w), &s1, &c1);
            g0 = ra * c0; g1 = ra * s0; g2 = rb * c1;
        }
        const float gs[3] = {g0, g1, g2};
#pragma unroll
        for (int a = 0; a < 3; ++a) o[a] = fminf(fmaxf(P[3 * j + a] + gs[a] * A.init_sigma, -lim), lim);
    };
    if (!morton) {
        for (int t = tid; t < A.n_opt; t += PREP_THREADS) {
            float o[3];
            draw(t, o);
            io[3 * t] = o[0]; io[3 * t + 1] = o[1]; io[3 * t + 2] = o[2];
        }
        return;
    }
    auto spread = [](uint32_t v) {                             // 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FFu;
        v = (v | (v << 8)) & 0x0300F00Fu;
        v = (v | (v << 4)) & 0x030C30C3u;
        v = (v | (v << 2)) & 0x09249249u;
        return v;
    };
    __syncthreads();                                           // (the subset's keys are done with)
    for (int t = tid; t < A.n_opt; t += PREP_THREADS) {
        float o[3];
        draw(t, o);
        uint32_t q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) q[a] = (uint32_t)fminf(fmaxf((o[a] + 0.5f) * 1024.f, 0.f), 1023.f);
        KEY[t] = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
    }
    __syncthreads();
    for (int t = tid; t < A.n_opt; t += PREP_THREADS) {
        const uint32_t kt = KEY[t];
        int rank = 0;                                          // position of draw t in the key-sorted order (ties by draw index)
        for (int u = 0; u < A.n_opt; ++u) { const uint32_t ku = KEY[u]; rank += (ku < kt || (ku == kt && u < t)) ? 1 : 0; }
        float o[3];
        draw(t, o);
        io[3 * rank] = o[0]; io[3 * rank + 1] = o[1]; io[3 * rank + 2] = o[2];
    }
}

// LDS by the cloud size: SOR 32 K + 128 B up to 4096 points (131,200 B there, 32,896 B at 1024), 12 K + 128 B above
// (120,128 B at 10,000); prepare 16 K + 256 B (160,256 B at 10,000)
static size_t sor_lds(int K) { return K <= SOR_NARROW_MAXK ? ((size_t)4 * K + 16) * sizeof(double) : (size_t)12 * K + 4 * (K & 1) + 128; }
static size_t prep_lds(int K, int n_opt = 0) {         // n_opt: the Morton keys of the optimised points share the subset keys' array
    return (size_t)12 * K + (size_t)4 * prep_nkey(K, n_opt <= PREP_SORT_MAX ? n_opt : 0) + 64 * 4;      // (sized for the sorted case whatever the hook says)
}

hipError_t configure_prep_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sor_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sor_lds(SOR_NARROW_MAXK));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sor_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sor_lds(PREP_MAXK));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sor_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sor_lds(SOR_NARROW_MAXK));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sor_kernel<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sor_lds(PREP_MAXK));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(prepare_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)prep_lds(PREP_MAXK));
}

hipError_t launch_sor(const float* pc, int B, int K, int k_nn, double alpha, uint8_t* keep, double* value, hipStream_t s) {
    if (K <= SOR_NARROW_MAXK) {
        if (k_nn <= 2) hipLaunchKernelGGL((sor_kernel<false, 3>), dim3(B), dim3(PREP_THREADS), sor_lds(K), s, pc, K, k_nn, alpha, keep, value);
        else hipLaunchKernelGGL((sor_kernel<false, 8>), dim3(B), dim3(PREP_THREADS), sor_lds(K), s, pc, K, k_nn, alpha, keep, value);
    } else {
        if (k_nn <= 2) hipLaunchKernelGGL((sor_kernel<true, 3>), dim3(B), dim3(PREP_THREADS), sor_lds(K), s, pc, K, k_nn, alpha, keep, value);
        else hipLaunchKernelGGL((sor_kernel<true, 8>), dim3(B), dim3(PREP_THREADS), sor_lds(K), s, pc, K, k_nn, alpha, keep, value);
    }
    return hipGetLastError();
}

hipError_t launch_prepare(const float* pc, const uint8_t* keep, int B, int K, const PrepArgs& a, const int32_t* sel_idx,
                          const int32_t* init_idx, const float* noise, float* sel, int32_t* t_per_cloud, float* init,
                          int32_t* n_kept, float* proc_out, hipStream_t s) {
    hipLaunchKernelGGL(prepare_kernel, dim3(B), dim3(PREP_THREADS), prep_lds(K, a.n_opt), s, pc, keep, K, a, sel_idx, init_idx, noise,
                       sel, t_per_cloud, init, n_kept, proc_out);
    return hipGetLastError();
}

}  // namespace ifd
