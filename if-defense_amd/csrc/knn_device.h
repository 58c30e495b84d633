// Shared device code of the optimiser kernels (ConvONet-Opt: optimize.hip, ONet-Opt: onet.hip): exact 5-NN from
// certified neighbour lists, the repulsion loss / gradient (defense/repulsion_loss.py:18-54), block helpers, the
// per-step kNN phase and the fused Adam update.  Header-only, everything is __forceinline__.
#pragma once
#include "ifd_device.h"
#include "ifd_internal.h"

namespace ifd {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// 5-NN + repulsion
// ---------------------------------------------------------------------------------------------
// [pcsamp:knn.scan]
struct Top5 {
    float d0, d1, d2, d3, d4;
    int i0, i1, i2, i3, i4;
};

__device__ __forceinline__ void top5_init(Top5& t) {
    t.d0 = t.d1 = t.d2 = t.d3 = t.d4 = INFINITY;
    t.i0 = t.i1 = t.i2 = t.i3 = t.i4 = 0;
}

// Branch-free sorted insertion (bubble the new key down with selects).  NOTE: the obvious nested-ternary form is turned
// into exec-mask control flow by hipcc (4x the code, I-cache thrash) - keep this shape.  Ties in d are broken by the
// smaller index.  Exact ties DO occur (clamped coordinates on flat faces), and
// every path - the scan, the exact list path, the key path via its ambiguity check - must pick the same neighbour:
// the 5 smallest (distance, index) pairs.
__device__ __forceinline__ void top5_insert_lex(Top5& t, float d, int j) {
    bool c; float lo; int li;
    c = d < t.d0 || (d == t.d0 && j < t.i0); lo = c ? d : t.d0; li = c ? j : t.i0; d = c ? t.d0 : d; j = c ? t.i0 : j; t.d0 = lo; t.i0 = li;
    c = d < t.d1 || (d == t.d1 && j < t.i1); lo = c ? d : t.d1; li = c ? j : t.i1; d = c ? t.d1 : d; j = c ? t.i1 : j; t.d1 = lo; t.i1 = li;
    c = d < t.d2 || (d == t.d2 && j < t.i2); lo = c ? d : t.d2; li = c ? j : t.i2; d = c ? t.d2 : d; j = c ? t.i2 : j; t.d2 = lo; t.i2 = li;
    c = d < t.d3 || (d == t.d3 && j < t.i3); lo = c ? d : t.d3; li = c ? j : t.i3; d = c ? t.d3 : d; j = c ? t.i3 : j; t.d3 = lo; t.i3 = li;
    c = d < t.d4 || (d == t.d4 && j < t.i4); t.d4 = c ? d : t.d4; t.i4 = c ? j : t.i4;
}

__device__ __forceinline__ void top5_insert(Top5& t, float d, int j) {
    if (d < t.d4) top5_insert_lex(t, d, j);     // rare after the first few dozen candidates; (d, index) order: a plain
                                                // strict-< bubble lets a carried element overtake an equal one
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

// The REFERENCE's neighbour choice, bug for bug (validation only: ifd_opt_params.knn_reference_form; ConvONet/defense/pn_utils.py:72-83):
//     inner = -2 * matmul(pc^T, pc);  xx = sum(pc ** 2);  dist = xx + inner + xx^T;  topk(-dist, k + 1)[:, :, 1:]
// in float32 with the accumulation order torch's CPU kernels use for it (verified bit for bit against torch.matmul / torch.sum in
// tests/test_oracle_golden.py on the host that runs it): dot_ij = fma(z_i, z_j, fma(y_i, y_j, x_i * x_j)), xx_i = (x_i^2 + y_i^2) +
// z_i^2 with every square rounded, dist_ij = (xx_j + (-2 dot_ij)) + xx_i.  The expanded form carries ~1e-7 of absolute noise, so (i)
// candidates whose true squared distances differ by less than that swap ranks, and (ii) column 0 of the top-6 - dropped as "self" - is
// the OTHER point of a pair closer than ~1.5e-4 (dist_ii is itself +-1e-8, not 0): self then stays in as a neighbour with a zero
// difference vector (loss term, no gradient), and the pair's term is missing.  Brute force over all K points, self included.
// torch.topk(k = 6) of a 1024-wide row on the CPU is libstdc++'s std::partial_sort over (value, index) pairs with the comparator
// "x.value > y.value" (ATen/native/cpu/TopKImpl.h: k * 64 <= n): a heap of the six best so far, replaced at the top only by a
// STRICTLY better candidate, then sort_heap.  Equal values - and the expanded form produces them: dist_ii and the distance to a
// converged pair partner are both exactly 0.0 or -7.45e-9, two candidates at the sixth place share their f32 distance - come out
// in the order that heap leaves them in, which is neither index order nor stable.  The heap operations of bits/stl_heap.h are
// restated here (__adjust_heap / __push_heap / __pop_heap, len <= 6) so that ties fall the same way.
struct RefHeap {
    float v[6];      // value = -dist  (comp(a, b) = a.v > b.v)
    int i[6];
};
__device__ __forceinline__ void ref_adjust_heap(RefHeap& h, int hole, int len, float val, int idx) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (h.v[child] > h.v[child - 1]) --child;
        h.v[hole] = h.v[child]; h.i[hole] = h.i[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h.v[hole] = h.v[child - 1]; h.i[hole] = h.i[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;                    // __push_heap
    while (hole > top && h.v[parent] > val) {
        h.v[hole] = h.v[parent]; h.i[hole] = h.i[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h.v[hole] = val; h.i[hole] = idx;
}
__device__ __forceinline__ float ref_xx(const f32x4& p);
__device__ __forceinline__ float rounded_mul(float a, float b) {      // a product that must be ROUNDED before it is added: hipcc
    float p = a * b;                                                   // contracts a * b + c into an fma wherever it can, also
    asm volatile("" : "+v"(p));                                        // through __fmul_rn - the empty asm makes the product opaque
    return p;
}
__device__ __forceinline__ float ref_xx(const f32x4& p) { return (rounded_mul(p.x, p.x) + rounded_mul(p.y, p.y)) + rounded_mul(p.z, p.z); }
__device__ __forceinline__ float ref_neg_dist(const f32x4& xi, float xxi, const f32x4& xj) {
    const float dot = __builtin_fmaf(xi.z, xj.z, __builtin_fmaf(xi.y, xj.y, rounded_mul(xi.x, xj.x)));
    return -((ref_xx(xj) + -2.f * dot) + xxi);      // (-2 dot is exact: a contraction of this sum rounds the same)
}
// the reference's five neighbours of point i (K >= 6): topk(-dist, 6)[1:]
__device__ __forceinline__ void knn_ref_point(const f32x4* __restrict__ X, int K, int i, Top5& t) {
    const f32x4 xi = X[i];
    const float xxi = ref_xx(xi);
    RefHeap h;
#pragma unroll
    for (int j = 0; j < 6; ++j) { h.v[j] = ref_neg_dist(xi, xxi, X[j]); h.i[j] = j; }
    for (int parent = 2; parent >= 0; --parent) {                      // std::make_heap, len = 6
        const float v = h.v[parent];
        const int id = h.i[parent];
        ref_adjust_heap(h, parent, 6, v, id);
    }
    for (int j = 6; j < K; ++j) {                                      // std::__heap_select
        const float v = ref_neg_dist(xi, xxi, X[j]);
        if (v > h.v[0]) ref_adjust_heap(h, 0, 6, v, j);                // __pop_heap with the result slot outside the heap
    }
    for (int last = 5; last >= 1; --last) {                            // std::__sort_heap
        const float v = h.v[last];
        const int id = h.i[last];
        h.v[last] = h.v[0]; h.i[last] = h.i[0];
        ref_adjust_heap(h, 0, last, v, id);
    }
    t.i0 = h.i[1]; t.i1 = h.i[2]; t.i2 = h.i[3]; t.i3 = h.i[4]; t.i4 = h.i[5];      // column 0 dropped, whatever it is
    t.d0 = t.d1 = t.d2 = t.d3 = 0.f;
    t.d4 = fmaxf(-h.v[5], 0.f);
}
__device__ __forceinline__ void knn_scan_ref2(const f32x4* __restrict__ X, int K, int pa, int pb, Top5& ta, Top5& tb) {
    knn_ref_point(X, K, min(pa, K - 1), ta);
    knn_ref_point(X, K, min(pb, K - 1), tb);
}

// ---------------------------------------------------------------------------------------------
// Certified neighbour lists: exact 5-NN at O(LIST_M) per point per step.
//
//   build:  for each point i store every j with |x_j - x_i| < rho_i, rho_i^2 = alpha2_i * (an upper bound of i's
//       squared 5-NN distance) - a LIST_F-entry front ball and a LIST_B-entry ring; the alpha2 adapt so that the
//       counts stay in range.  x0_i = x_i at build time.  Whole-cloud rebuilds (every wave, same step) start an
//       epoch; short-lived ("fragile") certificates are refreshed individually by their own wave.
//   step:  the 5 nearest list members are the true 5-NN iff  r5 < rho_i - |x_i - x0_i| - (D(now) + D(t_i)), with
//       D(t) = max_j (|x_j(t) - x0_j| + D(t_j))  bounding every point's displacement since the epoch start (a point
//       outside the list was >= rho_i away at build time).  Soft margin violated -> new lists for the NEXT step;
//       certificate violated -> this wave runs the exact brute-force scan for this step.
//   Either way every step uses the exact 5-NN set = the five smallest (distance, index) pairs, like the scan.
// ---------------------------------------------------------------------------------------------
// Sizes re-tuned in round 2 on the bench workload (scripts/ab_bench.sh; kernel launch of 2468 clouds): front / total
// 16 / 48: 867 ms, 24 / 48: 865, 32 / 48: 864, 40 / 48: 866, 24 / 40: 881, 32 / 56: 857, 16 / 64: 896, 24 / 64: 855, 32 / 64: 852,
// 40 / 64: 853, 48 / 64: 855 - longer lists mean fewer whole-cloud rebuilds (~140 k cycles per wave each) for more entries
// evaluated per step, and the evaluation became the cheaper side once it ran on key networks.
#ifndef IFD_LIST_F
#define IFD_LIST_F 32
#endif
constexpr int LIST_F = IFD_LIST_F;               // "front": every point within rho_f at build time (evaluated every step)
#ifndef IFD_LIST_M
#define IFD_LIST_M 64          // <= 64: knn_build_one fills a list's unused slots one per lane
#endif
constexpr int LIST_B = IFD_LIST_M - IFD_LIST_F;               // "back" : the ring rho_f <= d < rho_b (evaluated only when the front fails)
constexpr int LIST_M = LIST_F + LIST_B;  // uint16 entries per point; lists live in global memory (L2-resident)

// Per-step diagnostics of the kNN phase: event counts live in LDS (one atomic by lane 0 per event - they are rare or
// once per wave-step), not in registers that would have to survive the decoder tiles.  Summed into ifd_get_counters.
enum { CN_REBUILD = 0, CN_BRUTE, CN_PASS, CN_TIER2, CN_EXACT, CN_REFRESH, CN_TARGETS, CN_COUNT };
struct KnnCounters {
    unsigned int* lds;       // [CN_COUNT]
    int lane;
    __device__ __forceinline__ void bump(int which, unsigned int n = 1u) const {
        if (lane == 0) atomicAdd(lds + which, n);
    }
#if defined(IFD_PROF) || defined(IFD_TRACE)   // diagnostic builds only
    // IFD_PROF: per-wave cycle accumulators in LDS.  IFD_TRACE: time stamps of ONE step of cloud 0 in global memory
    // (null on every other step / cloud).  pc[PC_T] = time stamp of the last marker.
    unsigned long long* pc = nullptr;
#endif
};
enum { PC_BUILD = 0, PC_EVAL, PC_REP, PC_TILES, PC_WAIT, PC_ADAM, PC_T, PC_KNN0, PC_TILE0, PC_COUNT = 8 };
#if defined(IFD_PROF)
#define PROF_T0() do { if (cn.lane == 0 && cn.pc) cn.pc[PC_T] = __builtin_readcyclecounter(); } while (0)
#define PROF_ACC(v) do { if (cn.lane == 0 && cn.pc) { const unsigned long long n_ = __builtin_readcyclecounter(); cn.pc[v] += n_ - cn.pc[PC_T]; cn.pc[PC_T] = n_; } } while (0)
#elif defined(IFD_TRACE)
#define PROF_T0() do { if (cn.lane == 0 && cn.pc) cn.pc[PC_T] = __builtin_readcyclecounter(); } while (0)
#define PROF_ACC(v) do { if (cn.lane == 0 && cn.pc) cn.pc[v] = __builtin_readcyclecounter(); } while (0)
#else
#define PROF_T0()
#define PROF_ACC(v)
#endif
// -DIFD_TRACE only: finer stamps inside the kNN phase (slots 24 ...; the waits they force shift the phase a little)
#ifdef IFD_TRACE
#define TRACE_STAMP(slot, wait_asm) do { asm volatile(wait_asm ::: "memory"); if (cn.lane == 0 && cn.pc) cn.pc[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define TRACE_STAMP(slot, wait_asm)
#endif
#define pc_build PC_BUILD
#define pc_eval PC_EVAL
#define pc_rep PC_REP
#define pc_tiles PC_TILES
#define pc_wait PC_WAIT
#define pc_adam PC_ADAM

// [pcsamp:knn.exact_lists]
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
        top5_insert_lex(ta, da, cja);
        top5_insert_lex(tb, db, cjb);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// [pcsamp:knn.keys]
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

// [pcsamp:knn.build]
// One target of the wave-cooperative ("transposed") list build: the lanes hold the K candidate points in registers
// (16 each).  Every lane first collects its hits as two 16-bit masks (front ball / ring), the list positions come from
// wave prefix sums of the hit counts, and each lane then stores its (typically 0-2) hits - ~250 instructions per
// target instead of ~480 for a ballot / mbcnt compaction per candidate slice.  tf / tb: squared front / back radii
// (wave-uniform).  Front hits beyond LIST_F go to the ring segment behind the ring's own hits, so the ring stays
// complete when only the front overflows.  Unused slots get the dummy index.
__device__ __forceinline__ void knn_build_one(const float (&cx)[16], const float (&cy)[16], const float (&cz)[16],
                                              const f32x4 xi, int i, int lane, float tf, float tb,
                                              uint16_t* __restrict__ lst, int& nf_out, int& nb_out) {
    unsigned int hf = 0u, hb = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = lane + 64 * k;
        const float ex = cx[k] - xi.x, ey = cy[k] - xi.y, ez = cz[k] - xi.z;
        const float d = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
        const bool in_b = d < tb && j != i;
        const bool in_f = d < tf && in_b;
        hf |= (in_f ? 1u : 0u) << k;
        hb |= ((in_b && !in_f) ? 1u : 0u) << k;
    }
    const int cf = __popc(hf), cb = __popc(hb);
    // wave-inclusive prefix sums of both counts at once (cf | cb << 16; the totals are <= 1024): shifted adds inside the
    // rows of 16 through DPP (zero fill), the three row totals through readlane - no LDS crossbar (__shfl_up is a
    // ds_bpermute: the former version was a chain of six dependent LDS round trips per target)
    int pfb = cf | (cb << 16);
    pfb += __builtin_amdgcn_update_dpp(0, pfb, 0x111, 0xf, 0xf, true);     // row_shr:1
    pfb += __builtin_amdgcn_update_dpp(0, pfb, 0x112, 0xf, 0xf, true);     // row_shr:2
    pfb += __builtin_amdgcn_update_dpp(0, pfb, 0x114, 0xf, 0xf, true);     // row_shr:4
    pfb += __builtin_amdgcn_update_dpp(0, pfb, 0x118, 0xf, 0xf, true);     // row_shr:8
    const int r0 = __builtin_amdgcn_readlane(pfb, 15), r1 = __builtin_amdgcn_readlane(pfb, 31);
    const int r2 = __builtin_amdgcn_readlane(pfb, 47), r3 = __builtin_amdgcn_readlane(pfb, 63);
    const int row = lane >> 4;
    pfb += (row >= 1 ? r0 : 0) + (row >= 2 ? r1 : 0) + (row >= 3 ? r2 : 0);
    const int tot = r0 + r1 + r2 + r3;
    const int pf = pfb & 0xffff, pb = pfb >> 16;
    const int nf = tot & 0xffff, nb = tot >> 16;
    int at_f = pf - cf, at_b = pb - cb;
    while (hf != 0u) {
        const int k = __builtin_ctz(hf);
        hf &= hf - 1u;
        const int pos = at_f++;
        const int slot = pos < LIST_F ? pos : LIST_F + nb + (pos - LIST_F);       // overflowing front hits: behind the ring's
        if (slot < LIST_M) lst[slot] = (uint16_t)(lane + 64 * k);
    }
    while (hb != 0u) {
        const int k = __builtin_ctz(hb);
        hb &= hb - 1u;
        const int pos = at_b++;
        if (pos < LIST_B) lst[LIST_F + pos] = (uint16_t)(lane + 64 * k);
    }
    const int nb_all = nb + max(nf - LIST_F, 0);                                   // ring members + front hits that did not fit
    // unused slots point at the dummy X[MAXK] (the key evaluation does not look at counts)
    if (lane < LIST_M && lane >= (lane < LIST_F ? nf : LIST_F + nb_all)) lst[lane] = (uint16_t)MAXK;
    nf_out = nf;
    nb_out = nb_all;
}

__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// [pcsamp:knn.refresh]
// (Re)build the lists of the flagged points of this wave (need_a / need_b per lane; d4a / d4b: upper bounds of their
// squared 5-NN distances at the current positions).  Targets are taken one at a time off the ballot mask, so the
// cost is ~1 us for loading the candidates plus ~0.5 us per target - whole-cloud rebuilds (all flagged) and the
// individual refreshes of short-lived ("fragile") certificates share this code.  A ball that overflows its list is
// shrunk in proportion to the overshoot (hit count ~ r^2 on a surface) and rebuilt on the spot.
// pa / pb: the lane's two points (the target of lane l is read off lane l: any thread -> point mapping works).
__device__ __forceinline__ void knn_refresh(const f32x4* __restrict__ X, int K, int pa, int pb, int lane,
                                            uint16_t* __restrict__ lists, bool need_a, bool need_b, float d4a,
                                            float d4b, KnnPt& ka, KnnPt& kb, float dbase, float mv,
                                            const KnnCounters& cn) {
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
        int r_nf = 0, r_nb = 0;
        float r_tf = 0.f, r_tb = 0.f;
        unsigned long long mask = __ballot(half ? need_b : need_a);
#pragma unroll 1
        while (mask != 0ull) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1ull;
            const int i = __builtin_amdgcn_readlane(half ? pb : pa, l);
            float tf = readlane_f(my_tf, l), tb = readlane_f(my_tb, l);
            const f32x4 xi = X[i];                                                   // wave-uniform address
            uint16_t* lst = lists + (size_t)i * LIST_M;
            int nf, nb;
            cn.bump(CN_TARGETS);
#pragma unroll 1
            for (int rep = 0;; ++rep) {
                knn_build_one(cx, cy, cz, xi, i, lane, tf, tb, lst, nf, nb);
                if ((nf <= LIST_F && nb <= LIST_B) || rep == 3) break;
                cn.bump(CN_PASS);
                const float n_in = (float)(min(nf, LIST_F) + nb);                   // points inside the back radius
                float nf_new = (float)nf;
                if (nf > LIST_F) { tf *= (0.7f * LIST_F) / (float)nf; nf_new = 0.7f * LIST_F; }
                if (n_in - nf_new > 0.85f * LIST_B) tb *= (nf_new + 0.75f * LIST_B) / n_in;
                tb = fmaxf(tb, tf);
            }
            // the owning lane only records the outcome (four moves); its state is derived from it once, after the loop,
            // by all owners at the same time - inside the loop the square roots and divides below cost the whole wave
            // ~70 instructions per target for one lane's benefit
            if (lane == l) { r_nf = nf; r_nb = nb; r_tf = tf; r_tb = tb; }
        }
        if (half ? need_b : need_a) {
            const int nf = r_nf, nb = r_nb;
            const float tf = r_tf, tb = r_tb;
            kp.cnt_f = min(nf, LIST_F);
            kp.rho_f = nf <= LIST_F ? sqrtf(tf) : 0.f;          // front complete only if everything fitted
            kp.cnt_b = nb <= LIST_B ? nb : -1;
            kp.rho_b = sqrtf(tb);
            // carry the (possibly shrunk) radii forward as multiples of the 5-NN bound; grow slowly when sparse
            if (d4 > 0.f) { kp.al_f = tf / d4; kp.al_b = tb / d4; }
#ifndef IFD_FILL_Q
#define IFD_FILL_Q 2          // grow a radius while its list is less than IFD_FILL_Q / 4 full
#endif
            if (4 * nf < IFD_FILL_Q * LIST_F) kp.al_f *= 1.15f;
            if (4 * (nf + nb) < IFD_FILL_Q * LIST_M) kp.al_b *= 1.15f;
            kp.al_f = fminf(fmaxf(kp.al_f, 1.1f), 6.f);
            kp.al_b = fminf(fmaxf(kp.al_b, 1.2f), 30.f);
            kp.al_f = fminf(kp.al_f, kp.al_b);
            kp.x0 = X[half ? pb : pa];
            kp.dbase = dbase;
            // expected lifetime of the certificate ~ (rho - r5 - 6 mv) / (~2 mv per step)
#ifndef IFD_FRAG_MULT
#define IFD_FRAG_MULT 16.f
#endif
            kp.frag = kp.rho_b - sqrtf(d4) < IFD_FRAG_MULT * mv;
        }
    }
}

// [pcsamp:knn.resolve]
// (distance, index)-lexicographic minimum over the wave, in every lane: DPP inside the rows of 16 (xor 1, xor 2, the two
// mirrors), then the row / half swaps of gfx950 - twelve vector instructions instead of twelve ds_bpermute round trips
// (__shfl_xor) per call; knn_resolve_failures calls it five times per target and the whole workgroup waits for the
// wave that runs it.
__device__ __forceinline__ void lexmin_pair(float& bd, int& bj, float od, int oj) {
    const bool take = od < bd || (od == bd && oj < bj);
    bd = take ? od : bd;
    bj = take ? oj : bj;
}
template <int CTRL>
__device__ __forceinline__ void lexmin_dpp(float& bd, int& bj) {
    const float od = __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(bd), CTRL, 0xf, 0xf, false));
    const int oj = __builtin_amdgcn_update_dpp(0, bj, CTRL, 0xf, 0xf, false);
    lexmin_pair(bd, bj, od, oj);
}
__device__ __forceinline__ void lexmin_wave(float& bd, int& bj) {
    lexmin_dpp<0xB1>(bd, bj);        // quad_perm [1,0,3,2]
    lexmin_dpp<0x4E>(bd, bj);        // quad_perm [2,3,0,1]
    lexmin_dpp<0x141>(bd, bj);       // row_half_mirror
    lexmin_dpp<0x140>(bd, bj);       // row_mirror
    {
        const u32x2_t d = __builtin_amdgcn_permlane16_swap(__float_as_uint(bd), __float_as_uint(bd), false, false);
        const u32x2_t j = __builtin_amdgcn_permlane16_swap((unsigned int)bj, (unsigned int)bj, false, false);
        bd = __uint_as_float(d.x); bj = (int)j.x;
        lexmin_pair(bd, bj, __uint_as_float(d.y), (int)j.y);
    }
    {
        const u32x2_t d = __builtin_amdgcn_permlane32_swap(__float_as_uint(bd), __float_as_uint(bd), false, false);
        const u32x2_t j = __builtin_amdgcn_permlane32_swap((unsigned int)bj, (unsigned int)bj, false, false);
        bd = __uint_as_float(d.x); bj = (int)j.x;
        lexmin_pair(bd, bj, __uint_as_float(d.y), (int)j.y);
    }
}

// Exact 5-NN of a FEW points of this wave whose certificate failed (crowded balls, late failures): the wave's
// lanes hold the K candidates (16 each), per target five rounds of a (distance, index)-lexicographic minimum -
// ~500 instructions per target instead of a ~20 k-instruction scan by every lane of the wave.  Falls back to that
// scan when many lanes failed.
__device__ __forceinline__ void knn_resolve_failures(const f32x4* __restrict__ X, int K, int lane, int pa, int pb,
                                                     bool fail_a, bool fail_b, Top5& ta, Top5& tb) {
    const unsigned long long ma = __ballot(fail_a), mb = __ballot(fail_b);
    if (__popcll(ma) + __popcll(mb) > 24) {
        knn_scan2(X, K, pa, pb, ta, tb);
        return;
    }
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
        Top5& t = half ? tb : ta;
        unsigned long long mask = half ? mb : ma;
#pragma unroll 1
        while (mask != 0ull) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1ull;
            const int i = __builtin_amdgcn_readlane(half ? pb : pa, l);
            const f32x4 xi = X[i];
            float d[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float ex = cx[k] - xi.x, ey = cy[k] - xi.y, ez = cz[k] - xi.z;
                d[k] = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
                if (lane + 64 * k == i || lane + 64 * k >= K) d[k] = INFINITY;
            }
            float rd[5];
            int rj[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                float bd = d[0];
                int bk = 0;
#pragma unroll
                for (int k = 1; k < 16; ++k)
                    if (d[k] < bd) { bd = d[k]; bk = k; }            // ascending k = ascending index: ties keep the smaller
                int bj = lane + 64 * bk;
                lexmin_wave(bd, bj);
                rd[r] = bd;
                rj[r] = bj;
                if ((bj & 63) == lane) {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k == (bj >> 6)) d[k] = INFINITY;
                }
            }
            if (lane == l) {
                t.d0 = rd[0]; t.d1 = rd[1]; t.d2 = rd[2]; t.d3 = rd[3]; t.d4 = rd[4];
                t.i0 = rj[0]; t.i1 = rj[1]; t.i2 = rj[2]; t.i3 = rj[3]; t.i4 = rj[4];
            }
        }
    }
}

// [pcsamp:rep.standalone]
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

// Fixed-point accumulator of the repulsion gradient in LDS (optimiser kernels).  The neighbour -> point scatter of
// index_points' backward (repulsion_loss.py:43-47) is the one place where many threads add into the same point; LDS
// float atomics would make the sum depend on their order.  Integer sums do not: every term is rounded to a multiple of
// 2^-23, x and y share one 64-bit word (x in the high half; the halves' sums separate exactly as long as each stays inside
// 32 bits: |sum| < 256), z has a 32-bit word.  Headroom: a term's magnitude is |d loss / d d| = w (1 + (r - d) 2 d / h^2)
// <= 1.4 with the reference's r = 0.07, h = 0.03 (maximum near d = h; coincident points contribute 0), a point receives its
// own five centre terms plus one term from every point that counts it among its 5 nearest, and the in-degree of a 5-NN
// graph in three dimensions is at most 5 x 12 (kissing number): |sum| <= 65 x 1.4 = 91 < 256, whatever the input
// (test_repulsion_accumulators_do_not_wrap_on_a_tight_cluster holds the kernel against ifd_repulsion's 64-bit sums).  12 bytes of atomics per term instead of 24 (the phase is bound by LDS atomic throughput, not by arithmetic), 3
// conversion instructions per component instead of 11, 12 KB of LDS instead of 24.
__device__ __forceinline__ int opaque_zero_nomem() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
struct RepAcc {
    long long* xy;       // [MAXK]
    int* z;              // [MAXK]
};
constexpr float FIX32_SCALE = 8388608.0f;            // 2^23
constexpr float FIX32_INV = 1.0f / 8388608.0f;
__device__ __forceinline__ int fix32(float g) { return (int)rintf(g * FIX32_SCALE); }      // v_mul, v_rndne, v_cvt_i32 (saturating)
__device__ __forceinline__ long long pack_xy(int x, int y) { return (long long)(((unsigned long long)(unsigned int)x << 32)) + (long long)y; }
__device__ __forceinline__ void unpack_xy(long long s, int& x, int& y) {
    y = (int)(unsigned int)(unsigned long long)s;
    x = (int)((s - (long long)y) >> 32);
}

// Headroom check of the sums a point received this step (Adam phase, large_take_f): a sum at or beyond 2^30 (= 128.0; the
// bound derived above is 91) is within a factor two of wrapping into the neighbouring component / the sign bit, and a
// single saturated term (fix32 saturates at +-256: |coef * e| that large needs non-reference radius / h parameters) lands
// there too.  The launch's sticky overflow word counts such points; ifd_optimize_status() turns it into IFD_ERR_OVERFLOW.
__device__ __forceinline__ void rep_overflow_check(const int (&fi)[3], unsigned long long* status) {
    auto mag = [](int v) { return v < 0 ? 0u - (unsigned int)v : (unsigned int)v; };          // (|INT_MIN| = 2^31)
    const unsigned int big = max(max(mag(fi[0]), mag(fi[1])), mag(fi[2]));
    if (__builtin_expect(big >= (1u << 30), 0) && status != nullptr)
        __hip_atomic_fetch_add(status + STATUS_OVERFLOW, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// Split clouds (optimize.hip, "cooperative mode"): S = 2 or 4 workgroups - on as many CUs - share one cloud.  Member m owns
// the points {vt, vt + 512 : vt in [m * 512 / S, (m + 1) * 512 / S)}: their neighbour lists, their repulsion terms, their
// decoder tiles and their Adam update; every member keeps the whole cloud's positions in its LDS.  What crosses CUs goes
// through this per-cloud block of global memory: neighbour terms of points another member owns (fixed-point integer atomics:
// the total is the same integer whatever the order or the split), the updated positions, the displacement maxima of the
// neighbour-list certificates, and two monotonic arrival counters.
// Memory ordering: EVERY access to this block is an agent-scope atomic (coherent across the XCDs' L2s by itself: sc1
// write-through stores / sc1 loads), so publishing needs no cache write-back or invalidate - a member makes its stores
// and atomics complete (coop_publish: s_waitcnt vmcnt(0), which on gfx9 also covers stores; a workgroup-scope release fence
// is exactly that, and a compiler barrier) before it arrives at a counter, and a member that has seen the counter reads
// the data with agent-scope loads.  (A formal agent-scope release / acquire pair adds an L2 write-back of everything the
// XCD has dirtied - the parked scratch state of 32 workgroups - and an invalidate: 35 us per step, measured.)
// ---------------------------------------------------------------------------------------------
// [pcsamp:coop]
constexpr int MAX_COOP_WAVES = 16;    // owner waves of a split cloud: S members x (8 at S = 2, 4 at S = 4) = MAX_WAVES
struct CoopWs {
    // Positions, maxima and flags are double-buffered by step parity: a member that has passed bar_step(t) may run the whole
    // of step t + 1 - and publish its results - while a slower member is still copying the step-t values; the buffer of parity
    // p is only written again in step t + 2, i.e. behind bar_step(t + 1), at which the slower member arrives after its copy.
    // (With the repulsion term on, the bar_knn rendezvous in the middle of a step closed that window as well; without it
    // nothing did.)
    f32x4 X[2][MAXK];                 // positions after the last Adam step (written by the owners)
    long long Fxy[MAXK];              // neighbour terms received from other members, packed like RepAcc
    int Fz[MAXK];
    float L[MAXK][2];                 // last step only: BCE term, repulsion term of every point
    float scal[2][MAX_COOP_WAVES][2]; // per owner wave of the cloud: displacement maximum, step-length maximum of the next step
    int flag[2][4];                   // per member: whole-cloud list rebuild requested
    unsigned int bar_knn, bar_step;   // arrivals: members that have sent their neighbour terms; members that finished their step
    unsigned int pad[2];
};
struct CoopView {
    CoopWs* ws;
    int member;
    int par;                          // parity of the exchange buffers this step publishes into ((step + 1) & 1)
};
template <int S>
__device__ __forceinline__ bool coop_owns(int j, int member) {
    return S == 1 || ((j & (OPT_THREADS - 1)) / (OPT_THREADS / S)) == member;
}
__device__ __forceinline__ float coop_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coop_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Every wave that stored to / did atomics on the exchange block calls this BEFORE the workgroup barrier or atomic that lets a
// member arrive: on gfx9 one counter (vmcnt) covers loads, stores and atomics, and `s_waitcnt vmcnt(0)` returns when all of
// them have been acknowledged by the L2 (sc1: written through).  The workgroup-scope release fence alone is NOT that: on
// gfx950 it only waits for lgkmcnt (round-3 advisor finding, checked in the ISA: tests/test_abi_cpu.py greps the kernel).
__device__ __forceinline__ void coop_publish() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
__device__ __forceinline__ void coop_arrive(unsigned int* ctr) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Wait until `ctr` reaches `target` (call it wave-uniformly).  The wait is BOUNDED: members of a split cloud are co-resident
// by construction (one workgroup per CU, launch_optimize), but a CU mask, another process holding CUs with its own split
// launch, or a member that died would otherwise leave the survivors spinning for ever (return a status,
// never hang / exit).  After `limit` ticks of the 100 MHz wall clock (s_memrealtime: time the queue spends descheduled counts
// too, hence a generous default of 30 s - IFD_COOP_TIMEOUT_MS) the waiter raises the context's sticky time-out word and the
// time-out word of the current call (ifd_internal.h); every waiter that sees the CURRENT call's word falls out at once, the
// kernel leaves its step loop at the next workgroup barrier (results are garbage), and ifd_optimize_status() returns
// IFD_ERR_TIMEOUT.  Returns false if it gave up.
__device__ __forceinline__ bool coop_wait(unsigned int* ctr, unsigned int target, unsigned long long* status, unsigned int limit) {
    bool ok = true;
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned int it = 0;; ++it) {
            if ((it & 31u) == 0u && status != nullptr) {
                if (__hip_atomic_load(status + STATUS_TIMEOUT_CUR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { ok = false; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)limit) {
                    __hip_atomic_fetch_add(status + STATUS_TIMEOUT_CUR, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(status + STATUS_TIMEOUT, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
            if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}
// one neighbour (or centre) term into the accumulators of point j.  Split clouds accumulate the terms of points another
// member owns in their OWN LDS entries for those points first (unused otherwise) and send every entry once per step
// (coop_flush_remote): 20 scattered agent-scope atomics per lane in the middle of the phase measured 22 k cycles per step.
template <int S>
__device__ __forceinline__ void rep_scatter(const RepAcc F, const CoopView cv, int j, int fx, int fy, int fz) {
    atomicAdd(reinterpret_cast<unsigned long long*>(F.xy + j), (unsigned long long)pack_xy(fx, fy));
    atomicAdd(F.z + j, fz);
}
// what this member's points sent to points of other members: one coalesced pass of integer atomics onto the owners' global
// accumulators (one wave, after every owner wave of the member has finished its repulsion terms), entries cleared
template <int S>
__device__ __forceinline__ void coop_flush_remote(const RepAcc F, const CoopView cv, int K, int lane) {
    for (int j = lane; j < K; j += 64) {
        if (coop_owns<S>(j, cv.member)) continue;
        const long long fxy = F.xy[j];
        const int fz = F.z[j];
        if (fxy != 0) {
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(cv.ws->Fxy + j), (unsigned long long)fxy,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            F.xy[j] = 0;
        }
        if (fz != 0) {
            __hip_atomic_fetch_add(cv.ws->Fz + j, fz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            F.z[j] = 0;
        }
    }
}

// [pcsamp:knn.rep_terms]
// The two owned points together (optimiser): one instruction stream with the two independent chains interleaved -
// two rep_point calls under separate `if (p < K)` branches cannot overlap their LDS / sqrt / exp / divide latencies.
// The centre's own share (minus what its five neighbours receive) goes into F with the same fixed-point atomics: integer
// addition commutes, so F[i] ends up as the same sum whatever the order, and no register has to carry it to the Adam phase.
template <int S = 1>
__device__ __forceinline__ void rep_point2(const f32x4* __restrict__ X, const RepAcc F, int K, int pa, int pb,
                                           const Top5& ta, const Top5& tb, const RepConst rc, float& loss_a,
                                           float& loss_b, const CoopView cv = CoopView{nullptr, 0, 0}) {
    const bool va = pa < K, vb = pb < K;
    const f32x4 xa = X[min(pa, K - 1)], xb = X[min(pb, K - 1)];
    const int ia[5] = {ta.i0, ta.i1, ta.i2, ta.i3, ta.i4}, ib[5] = {tb.i0, tb.i1, tb.i2, tb.i3, tb.i4};
    float la = 0.f, lb = 0.f;
    int gca[3] = {0, 0, 0}, gcb[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int ja = va ? ia[k] : 0, jb = vb ? ib[k] : 0;
        const f32x4 qa = X[ja], qb = X[jb];
        int f[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // hipcc contracts a * b + c into an fma wherever it likes (-ffp-contract=fast ignores pragmas), and it likes
            // different places in the one-chain instantiation of split clouds and in this two-chain one.  So every
            // multiply-add below is SPELLED as an fma and no product is left next to an add it could be fused into: the
            // bits do not depend on the instantiation.
            const f32x4 xi = s ? xb : xa, xj = s ? qb : qa;
            const float ex = xj.x - xi.x, ey = xj.y - xi.y, ez = xj.z - xi.z;
            const float d2raw = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
            const float d2 = fmaxf(d2raw, rc.eps);
            // sqrt, 1 / h, 1 / d and exp through the hardware instructions (v_sqrt_f32, v_rcp_f32, v_exp_f32: 1 ulp each)
            // instead of the IEEE expansions (~10 dependent instructions per divide / sqrt, ~15 per expf; -DIFD_EXACT_REP
            // brings them back): -1.3 % of the kernel on the bench workload, and the hot kernel's gradient stays at
            // 3.6e-7 ... 6.4e-7 of the reference gradient's maximum (4e-7 ... 6e-7 with the expansions) - the f32 rounding
            // of the terms themselves dominates.
#ifndef IFD_EXACT_REP
            const float d = __builtin_amdgcn_sqrtf(d2);
            const float ih = __builtin_amdgcn_rcpf(rc.h);
            const float q = d * ih;
            const float w = __expf(-(q * q));
            const float rd = rc.radius - d, rw = rd * w;
            (s ? lb : la) = __builtin_fmaf(rd, w, s ? lb : la);
            const float dd = __builtin_fmaf(-rw, (q + q) * ih, -w);
            const float coef = d2raw > rc.eps ? dd * __builtin_amdgcn_rcpf(d) : 0.f;
#else
            const float d = sqrtf(d2);
            const float q = d / rc.h;
            const float w = expf(-(q * q));
            const float rd = rc.radius - d, rw = rd * w;
            (s ? lb : la) = __builtin_fmaf(rd, w, s ? lb : la);
            const float dd = __builtin_fmaf(-rw, (q + q) / rc.h, -w);
            const float coef = d2raw > rc.eps ? dd / d : 0.f;
#endif
            f[s][0] = fix32(coef * ex);
            f[s][1] = fix32(coef * ey);
            f[s][2] = fix32(coef * ez);
        }
        if (va) {
            gca[0] -= f[0][0]; gca[1] -= f[0][1]; gca[2] -= f[0][2];
            rep_scatter<S>(F, cv, ja, f[0][0], f[0][1], f[0][2]);
        }
        if (vb) {
            gcb[0] -= f[1][0]; gcb[1] -= f[1][1]; gcb[2] -= f[1][2];
            rep_scatter<S>(F, cv, jb, f[1][0], f[1][1], f[1][2]);
        }
    }
    if (va) {
        atomicAdd(reinterpret_cast<unsigned long long*>(F.xy + pa), (unsigned long long)pack_xy(gca[0], gca[1]));
        atomicAdd(F.z + pa, gca[2]);
    }
    if (vb) {
        atomicAdd(reinterpret_cast<unsigned long long*>(F.xy + pb), (unsigned long long)pack_xy(gcb[0], gcb[1]));
        atomicAdd(F.z + pb, gcb[2]);
    }
    loss_a = la;
    loss_b = lb;
}

// ---------------------------------------------------------------------------------------------
// block helpers
// ---------------------------------------------------------------------------------------------
// [pcsamp:block_helpers]
constexpr int OWN_WAVES = OPT_THREADS / 64;   // waves whose threads own points (kNN / Adam duty): threads [0, 512)
constexpr int MAX_WAVES = 16;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// max over the wave in every lane: rotations inside the rows of 16 (DPP), then the row and half swaps - no LDS crossbar
// (__shfl_xor compiles to ds_bpermute: six dependent LDS round trips; the Adam phase does this twice per step, at the end
// of the step's critical path: -0.45 % of the kernel on the bench workload)
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x128, 0xf, 0xf, false)));   // row_ror:8
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x124, 0xf, 0xf, false)));   // row_ror:4
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x122, 0xf, 0xf, false)));   // row_ror:2
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x121, 0xf, 0xf, false)));   // row_ror:1
    const u32x2_t a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a.x), __uint_as_float(a.y));
    const u32x2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b.x), __uint_as_float(b.y));
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
    // (every multiply-add SPELLED: hipcc contracts a * a + b * b + c * c differently in different instantiations - the fused tail of
    // the split-precision optimiser and the stand-alone kernel differed in the last bit of one cloud's radius, round 6)
    const float da = pa < K ? sqrtf(__builtin_fmaf(a.z, a.z, __builtin_fmaf(a.y, a.y, a.x * a.x))) : 0.f;
    const float db = pb < K ? sqrtf(__builtin_fmaf(b.z, b.z, __builtin_fmaf(b.y, b.y, b.x * b.x))) : 0.f;
    const float md = block_max(fmaxf(da, db), scratch);
    if (pa < K) X[pa] = f32x4{a.x / md, a.y / md, a.z / md, 0.f};
    if (pb < K) X[pb] = f32x4{b.x / md, b.y / md, b.z / md, 0.f};
    __syncthreads();
}


// [pcsamp:knn.phase]
// Block-shared scalars of the neighbour-list protocol (LDS).
struct KnnShared {
    float* dmaxbuf;                  // [2][MAX_WAVES] per-wave max displacement from the epoch reference (next step)
    float* movebuf;                  // [2][MAX_WAVES] per-wave max single-step move
    volatile int* rebuild_flag;      // [2] by step parity: whole-cloud rebuild requested
};

// kNN + repulsion of the two points (pa, pb) this lane owns, for one optimiser step (all lanes of an owner wave call
// it together).  Leaves the loss terms in rep_loss_a/b and scatters the neighbour AND centre gradients into F (fixed point).
template <int S = 1>
__device__ __forceinline__ void knn_phase(const f32x4* __restrict__ X, const RepAcc F, int K, int pa, int pb,
                                          int wave, int lane, int step, bool last, int scan_every_step /* 1: exact scan, 2: the reference's form */,
                                          const uint16_t* La, const uint16_t* Lb, uint16_t* cloud_lists, KnnPt& ka,
                                          KnnPt& kb, const KnnShared& sh, const RepConst rc, float& rep_loss_a,
                                          float& rep_loss_b, KnnCounters& cn, const CoopView cv = CoopView{nullptr, 0, 0}) {
    float* const dmaxbuf = sh.dmaxbuf;
    float* const movebuf = sh.movebuf;
    volatile int* const rebuild_flag = sh.rebuild_flag;
    Top5 ta, tb;
    top5_init(ta);
    top5_init(tb);
    const int ia = min(pa, K - 1), ib = min(pb, K - 1);
    if (scan_every_step == 2) {
        knn_scan_ref2(X, K, pa, pb, ta, tb);
    } else if (scan_every_step) {
        knn_scan2(X, K, pa, pb, ta, tb);
    } else {
        // A whole-cloud rebuild (new epoch) only when more than IFD_REBUILD_MIN long-lived certificates are about to expire;
        // fewer are refreshed individually by their own waves like the fragile ones (~250 instructions per list against
        // ~300 k cycles per wave for the rebuild).  Round 3, kernel launch of the bench workload / 512 clouds on the
        // trained-like field (ms): 0: 829.9 / 182.1, 4: 824.3 / 179.9, 16: 825.5 / 176.9, 64: 831.9 / 174.6 (whole-cloud
        // rebuilds per cloud on the trained-like field 12.8 -> 9.7 -> 8.1 -> 6.9).
#ifndef IFD_REBUILD_MIN
#define IFD_REBUILD_MIN 16
#endif
        const bool force = step == 0 || rebuild_flag[step & 1] > IFD_REBUILD_MIN;      // block-uniform
        float dmax = 0.f, mv = 0.f;
#pragma unroll
        for (int w = 0; w < (S > 1 ? MAX_WAVES : OWN_WAVES); ++w) {      // (split clouds: slots 8 ... hold the other members' maxima)
            dmax = fmaxf(dmax, dmaxbuf[(step & 1) * MAX_WAVES + w]);
            mv = fmaxf(mv, movebuf[(step & 1) * MAX_WAVES + w]);
        }
        // Will the certificate survive one more step?  The analytic bound is 4 mv (r5 grows <= 2 mv, own and the others'
        // displacement <= mv each; mv = largest move of the step just taken).  A certificate that does fail only costs an
        // exact per-point query (knn_resolve_failures, ~1 us), never correctness, while a soft failure of a long-lived
        // certificate triggers a whole-cloud rebuild (~300 k cycles per wave) - so the margin is tuned, not derived.
        // Round 3, kernel launch of the bench workload / 512 clouds on the trained-like field (ms): 4 mv 836.6 / 186.5,
        // 3 mv 832.3, 2 mv 830.7, 1 mv 828.0 / 182.1, 0.5 mv 827.2 / 182.8, 0: 826.8 / 184.2 (certificate failures 4.6 k ->
        // 19 k -> 33 k -> 46 k per 512 clouds on the trained-like field).
#ifndef IFD_SOFT_SLACK
#define IFD_SOFT_SLACK 1.f
#endif
        const float soft_slack = IFD_SOFT_SLACK * mv;
        TRACE_STAMP(25, "s_waitcnt lgkmcnt(0)");              // flags and displacement maxima read
        PROF_T0();
        const bool need_a = pa < K && (force || ka.pend), need_b = pb < K && (force || kb.pend);
        if (__any(need_a || need_b)) {
            float d4a, d4b;        // upper bounds of the squared 5-NN distances at the current positions
            if (force) {
                // ---- synchronous whole-cloud rebuild (every owner wave, this step): new epoch ---------
                cn.bump(CN_REBUILD);
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
                    cn.bump(CN_PASS);
                    knn_scan2(X, K, pa, pb, ta, tb);
                }
                d4a = ta.d4;
                d4b = tb.d4;
                dmax = 0.f;
            } else {
                // ---- individual refresh of fragile certificates: r5 grows by at most 2 mv per step ----
                cn.bump(CN_REFRESH);
                const float ra = ka.r5p + 2.f * mv, rb = kb.r5p + 2.f * mv;
                d4a = ra * ra;
                d4b = rb * rb;
            }
            knn_refresh(X, K, pa, pb, lane, cloud_lists, need_a, need_b, d4a, d4b, ka, kb, dmax, mv, cn);
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
            TRACE_STAMP(26, "s_waitcnt vmcnt(0)");            // list words here
            list_keys6_2<0, LIST_F / 8>(X, wa, wb, ia, ib, qa, qb);
            TRACE_STAMP(27, "s_nop 0");                        // front evaluated
            float r5a = sqrtf(key_d_upper(qa.k4)), r5b = sqrtf(key_d_upper(qb.k4));
            // every point outside a ball of build radius rho is now farther than rho - (spent budget)
            const bool ok1 = (pa >= K || r5a < (ka.rho_f - hs_a) * 0.99999f - 1e-7f) &&
                             (pb >= K || r5b < (kb.rho_f - hs_b) * 0.99999f - 1e-7f);
            bool fail_a = false, fail_b = false;      // hard certificate failures (only possible once the ring was needed)
            if (!__all(ok1)) {
                cn.bump(CN_TIER2);
                list_keys6_2<LIST_F / 8, LIST_M / 8>(X, wa, wb, ia, ib, qa, qb);
                r5a = sqrtf(key_d_upper(qa.k4));
                r5b = sqrtf(key_d_upper(qb.k4));
                fail_a = pa < K && !(ka.cnt_b >= 0 && r5a < (ka.rho_b - hs_a) * 0.99999f - 1e-7f);
                fail_b = pb < K && !(kb.cnt_b >= 0 && r5b < (kb.rho_b - hs_b) * 0.99999f - 1e-7f);
                // will it still hold next step?  (crowded balls, cnt_b < 0, are served by the exact query anyway)
                soft_a = pa >= K || ka.cnt_b < 0 || r5a < (ka.rho_b - spent_a) * 0.99999f - 1e-7f - soft_slack;
                soft_b = pb >= K || kb.cnt_b < 0 || r5b < (kb.rho_b - spent_b) * 0.99999f - 1e-7f - soft_slack;
            }
            // (Measured in round 2: 91 % of the wave-steps get here with some point that needs the ring, and in 86 % of those
            // more than 8 of the wave's 128 points do - letting the wave evaluate such points' lists one at a time, one
            // entry per lane and six wave-wide minima each, only paid below that count and came out 0.8 % slower overall.)
            // A near-tie between the 5th and 6th key (equal above the index bits: ~10 % of the wave-steps have one
            // somewhere among their 128 points) is settled for THAT point by the exact wave-cooperative query below
            // (~2 k cycles) - re-running the whole wave on the exact insertion path cost ~70 k cycles each time.
            const bool amb_a = pa < K && !fail_a && keys6_ambiguous(qa), amb_b = pb < K && !fail_b && keys6_ambiguous(qb);
            {
            keys6_to_top5(qa, ta);
            keys6_to_top5(qb, tb);
            if (__any(amb_a || amb_b)) cn.bump(CN_EXACT);
            if (__builtin_expect(__any(fail_a || fail_b), 0)) cn.bump(CN_BRUTE);      // certificate failed
            if (__any(fail_a || fail_b || amb_a || amb_b))     // exact query for those points, this step
                knn_resolve_failures(X, K, lane, pa, pb, fail_a || amb_a, fail_b || amb_b, ta, tb);
            }
        }
        if (exact) {
            // ---- exact path: sorted insertion with indices (last step, near-ties) --------------------
            cn.bump(CN_EXACT);
            soft_a = soft_b = true;
            top5_init(ta);
            top5_init(tb);
            list_top5_2<0, LIST_F>(X, La, Lb, ka.cnt_f, kb.cnt_f, ia, ib, ta, tb);
            const bool ok1 = (pa >= K || sqrtf(ta.d4) < (ka.rho_f - hs_a) * 0.99999f - 1e-7f) &&
                             (pb >= K || sqrtf(tb.d4) < (kb.rho_f - hs_b) * 0.99999f - 1e-7f);
            if (!__all(ok1)) {
                list_top5_2<LIST_F, LIST_M>(X, La, Lb, LIST_F + ka.cnt_b, LIST_F + kb.cnt_b, ia, ib, ta, tb);
                const float r5a = sqrtf(ta.d4), r5b = sqrtf(tb.d4);
                const bool fail_a = pa < K && !(ka.cnt_b >= 0 && r5a < (ka.rho_b - hs_a) * 0.99999f - 1e-7f);
                const bool fail_b = pb < K && !(kb.cnt_b >= 0 && r5b < (kb.rho_b - hs_b) * 0.99999f - 1e-7f);
                if (__builtin_expect(__any(fail_a || fail_b), 0)) {
                    cn.bump(CN_BRUTE);
                    knn_resolve_failures(X, K, lane, pa, pb, fail_a, fail_b, ta, tb);
                }
                soft_a = pa >= K || ka.cnt_b < 0 || r5a < (ka.rho_b - spent_a) * 0.99999f - 1e-7f - soft_slack;
                soft_b = pb >= K || kb.cnt_b < 0 || r5b < (kb.rho_b - spent_b) * 0.99999f - 1e-7f - soft_slack;
            }
        }
        ka.r5p = sqrtf(ta.d4);
        kb.r5p = sqrtf(tb.d4);
        // a certificate about to expire: fragile ones are refreshed individually next step, the others
        // mean the epoch is old -> whole-cloud rebuild next step
        // (rebuild_flag counts the long-lived certificates about to expire; up to IFD_REBUILD_MIN of them are refreshed
        // individually like the fragile ones instead of starting a new epoch for the whole cloud)
        ka.pend = !soft_a && (ka.frag || IFD_REBUILD_MIN > 0);
        kb.pend = !soft_b && (kb.frag || IFD_REBUILD_MIN > 0);
        const int nbad = __popcll(__ballot(!soft_a && !ka.frag)) + __popcll(__ballot(!soft_b && !kb.frag));
        if (nbad != 0 && lane == 0) atomicAdd(const_cast<int*>(rebuild_flag) + ((step + 1) & 1), nbad);
        PROF_ACC(pc_eval);
    }
    rep_point2<S>(X, F, K, pa, pb, ta, tb, rc, rep_loss_a, rep_loss_b, cv);
    PROF_ACC(pc_rep);
}

// [pcsamp:adam.update]
// Adam state of the two owned points (torch.optim.Adam keeps exp_avg / exp_avg_sq per coordinate).
struct AdamState {
    float mm[6], vv[6];
};

// Fused Adam update of the two owned points (torch/optim/adam.py _single_tensor_adam).  G: occupancy gradient,
// F: repulsion gradient in fixed point, centre and neighbour terms (scaled by rep_scale).  step_size = lr / (1 - beta1^t)
// and bc2 = sqrt(1 - beta2^t) come from a per-step table the host computes in double like torch's Python scalars
// (adam_table in api.cpp).  Also reduces this wave's displacement / step-length maxima for the neighbour-list
// certificates of the next step.
template <int S = 1>
__device__ __forceinline__ void adam_phase(f32x4* __restrict__ X, const f32x4* __restrict__ G, const RepAcc F,
                                           int K, int pa, int pb, float step_size, float bc2, float rep_scale,
                                           AdamState& st, float (&xnew)[2][3], float& mv2_out,
                                           unsigned long long* status = nullptr, const CoopView cv = CoopView{nullptr, 0, 0}) {
    float (&mm)[6] = st.mm;
    float (&vv)[6] = st.vv;
    // ---- Adam (torch/optim/adam.py _single_tensor_adam: lerp form, eps added after the bias-
    //      corrected sqrt) ---------------------------------------------------------------------
    float mv2 = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pt = q ? pb : pa;
        xnew[q][0] = xnew[q][1] = xnew[q][2] = 0.f;
        if (pt < K) {
            const f32x4 go = G[pt];
            const f32x4 x = X[pt];
            const float gocc[3] = {go.x, go.y, go.z};
            float xs[3] = {x.x, x.y, x.z};
            float msq = 0.f;
            int fi[3];
            long long fxy = F.xy[pt];
            fi[2] = F.z[pt];
            if (S > 1) {        // what the other members' points sent here: integer sums, the same total as in one workgroup
                fxy += (long long)__hip_atomic_exchange(reinterpret_cast<unsigned long long*>(cv.ws->Fxy + pt), 0ull,
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fi[2] += __hip_atomic_exchange(cv.ws->Fz + pt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unpack_xy(fxy, fi[0], fi[1]);
            {   // clear the accumulators with a zero made HERE: the literal 0 of a 64-bit store is a register pair the compiler hoists
                // out of the step loop and spills across the decoder tiles - its reload was a scratch round trip per point
                const int z0 = opaque_zero_nomem();
                int* fw = reinterpret_cast<int*>(F.xy + pt);
                fw[0] = z0;
                fw[1] = z0;
                F.z[pt] = z0;
            }
            rep_overflow_check(fi, status);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                // (every multiply-add spelled as an fma, see rep_point2: the one- and the two-point instantiations of this
                // function must agree bit for bit)
                const float gn = (float)fi[a] * FIX32_INV;
                const float g = __builtin_fmaf(gn, rep_scale, gocc[a]);
                float& mr = mm[3 * q + a];
                float& vr = vv[3 * q + a];
                mr = __builtin_fmaf(g - mr, 1.f - 0.9f, mr);                             // lerp_(grad, 1 - beta1)
                vr = __builtin_fmaf((1.f - 0.999f) * g, g, vr * 0.999f);                 // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
                // IEEE sqrt and divisions, like torch's kernels (the hardware v_sqrt_f32 / v_rcp_f32 forms measured -0.25 %
                // of the kernel on the bench workload: not taken for the update itself)
                const float denom = sqrtf(vr) / bc2 + 1e-8f;
                const float ratio = mr / denom;
                const float upd = step_size * ratio;
                xs[a] = __builtin_fmaf(-step_size, ratio, xs[a]);                        // addcdiv_(exp_avg, denom, value = -step_size)
                msq = fmaf(upd, upd, msq);
                xnew[q][a] = xs[a];
            }
            mv2 = fmaxf(mv2, msq);
            X[pt] = f32x4{xs[0], xs[1], xs[2], 1.f};     // .w = 1: fc_p's bias input on the matrix pipe (optimize.hip)
            if (S > 1) {
                float* xg = reinterpret_cast<float*>(cv.ws->X[cv.par] + pt);
                coop_st(xg, xs[0]); coop_st(xg + 1, xs[1]); coop_st(xg + 2, xs[2]);
            }
        }
    }
    mv2_out = mv2;
}

// [pcsamp:adam.displacement]
// Second half of the Adam phase: this wave's displacement / step-length maxima for the neighbour-list certificates of
// the next step.  It is the only part that needs the parked per-point state (x0, dbase: a scratch round trip issued just
// before the mid-step barrier), so it runs LAST - after the update, the sampling coordinates and the moments - and the
// round trip of the wave that reached the barrier last hides under that work instead of stalling it.
__device__ __forceinline__ void adam_displacement(int K, int pa, int pb, int wave, int lane, int step,
                                                  const float (&xnew)[2][3], float mv2, const KnnPt& ka, const KnnPt& kb,
                                                  const KnnShared& sh, bool owner_wave = true) {
    float dmax2 = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int pt = q ? pb : pa;
        if (pt < K) {
            const f32x4 x0 = q ? kb.x0 : ka.x0;
            const float dsq = (xnew[q][0] - x0.x) * (xnew[q][0] - x0.x) + (xnew[q][1] - x0.y) * (xnew[q][1] - x0.y) +
                              (xnew[q][2] - x0.z) * (xnew[q][2] - x0.z);
            dmax2 = fmaxf(dmax2, sqrtf(dsq) + (q ? kb.dbase : ka.dbase));
        }
    }
    dmax2 = wave_max(dmax2);
    mv2 = wave_max(mv2);
    if (lane == 0 && owner_wave) {      // (split clouds: the waves without owner threads have no slot - theirs would be a peer's)
        sh.dmaxbuf[((step + 1) & 1) * MAX_WAVES + wave] = dmax2 * 1.00001f + 1e-7f;
        sh.movebuf[((step + 1) & 1) * MAX_WAVES + wave] = sqrtf(mv2);
    }
}

// ---------------------------------------------------------------------------------------------
// Parking: per-thread state that must survive a register-hungry phase is written to a private (scratch) array by hand,
// in one batch of 16-byte stores, and read back in one batch - instead of leaving it to the register allocator, which
// spills such values one by one and reloads them lazily, each with its own memory round trip, inside the latency-bound
// phases (and whose choices move by +-15 % with semantically neutral edits).  The slot index goes through an opaque zero
// (one per park / unpark site, with a memory clobber), so the compiler can neither forward a store to the matching load
// nor index the array statically; and the array is kept at 256 bytes: AMDGPU's alloca-to-vector promotion turned a
// 128-byte one into a register vector with dynamic indexing (3,000 spilled registers, 4x the step time), and a volatile
// one into uncached flat accesses.
// ---------------------------------------------------------------------------------------------
// [pcsamp:park]
__device__ __forceinline__ int opaque_zero() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z) : : "memory");
    return z;
}
constexpr int PARK_KNN = 0;       // 4 slots per KnnPt: ka at 0, kb at 4
constexpr int PARK_SLOTS = 16;     // (8 used; see the note on alloca promotion above)

__device__ __forceinline__ void park_knnpt(f32x4* park, int z, int base, const KnnPt& k) {
    park[z + base + 0] = f32x4{__int_as_float(k.cnt_f), __int_as_float(k.cnt_b), k.rho_f, k.rho_b};
    park[z + base + 1] = f32x4{k.al_f, k.al_b, k.dbase, k.r5p};
    park[z + base + 2] = f32x4{k.x0.x, k.x0.y, k.x0.z, __int_as_float((k.frag ? 1 : 0) | (k.pend ? 2 : 0))};
}
__device__ __forceinline__ void unpark_knnpt(const f32x4* park, int z, int base, KnnPt& k) {
    const f32x4 a = park[z + base + 0], b = park[z + base + 1], c = park[z + base + 2];
    k.cnt_f = __float_as_int(a.x); k.cnt_b = __float_as_int(a.y); k.rho_f = a.z; k.rho_b = a.w;
    k.al_f = b.x; k.al_b = b.y; k.dbase = b.z; k.r5p = b.w;
    k.x0 = f32x4{c.x, c.y, c.z, 0.f};
    const int fl = __float_as_int(c.w);
    k.frag = (fl & 1) != 0;
    k.pend = (fl & 2) != 0;
}
// [pcsamp:adam.moments]
// The Adam moments of the owner threads sleep in LDS between Adam phases: mv[3][OPT_THREADS] float4, slot k of thread
// t at mv[k * OPT_THREADS + t] (conflict-free 16-byte accesses).
__device__ __forceinline__ void store_adam(f32x4* mv, int t, const AdamState& st) {
    mv[t] = f32x4{st.mm[0], st.mm[1], st.mm[2], st.mm[3]};
    mv[OPT_THREADS + t] = f32x4{st.mm[4], st.mm[5], st.vv[0], st.vv[1]};
    mv[2 * OPT_THREADS + t] = f32x4{st.vv[2], st.vv[3], st.vv[4], st.vv[5]};
}
__device__ __forceinline__ void load_adam(const f32x4* mv, int t, AdamState& st) {
    const f32x4 a = mv[t], b = mv[OPT_THREADS + t], c = mv[2 * OPT_THREADS + t];
    st.mm[0] = a.x; st.mm[1] = a.y; st.mm[2] = a.z; st.mm[3] = a.w; st.mm[4] = b.x; st.mm[5] = b.y;
    st.vv[0] = b.z; st.vv[1] = b.w; st.vv[2] = c.x; st.vv[3] = c.y; st.vv[4] = c.z; st.vv[5] = c.w;
}

}  // namespace ifd
