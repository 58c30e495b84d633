// ONet-Mesh path (ONet/remesh_defense.py; BASELINE config #4): occupancy grid by
// multi-resolution iso-surface extraction -> marching cubes -> area-weighted surface sampling, all on the GPU.
//
//   generate_from_latent (im2mesh/onet/generation.py:88-135) evaluates the decoder on a 33^3 grid and refines it twice
//   around the surface (MISE, im2mesh/utils/libmise/mise.pyx) up to 129^3, extracts the iso-surface of the padded
//   grid (extract_mesh :155-178, libmcubes) and remesh_defense.py:150-170 draws 1024 area-weighted surface samples.
//
// MISE as dense arrays instead of an octree with a hash map: per cloud val[P^3] / known[P^3] (P = R + 1 grid points
// per axis) and one "subdivided" flag array per level.  A leaf voxel is subdivided when the known points of its
// CLOSED cube lie on both sides of the threshold (mise.pyx:181-215: every known point marks the <= 8 leaf voxels that
// touch it - including big unsubdivided neighbours, which is why activation can cascade over several rounds), which
// adds the 27 points of the half-size lattice (:217-262).  Rounds repeat until no point is pending, like the
// query / update loop of generation.py:112-127.  The values come from the MFMA decoder of onet.hip; comparisons with the
// threshold are done in double like the reference (float32 logits widened to float64, generation.py:124).
#include <cstring>

#include "ifd_device.h"
#include "ifd_internal.h"
#include "mc_table_data.h"

namespace ifd {

// ---------------------------------------------------------------------------------------------
// MISE bookkeeping
// ---------------------------------------------------------------------------------------------
__global__ void mise_init_kernel(MiseGrid g) {
    // all lattice points of the coarsest level are pending (mise.pyx:76-86)
    const int cloud = blockIdx.y;
    const int n0 = g.res0 + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n0 * n0 * n0) return;
    const int s = 1 << g.depth;
    const int x = (i / (n0 * n0)) * s, y = ((i / n0) % n0) * s, z = (i % n0) * s;
    const int idx = (x * g.P + y) * g.P + z;
    g.pend[(size_t)cloud * g.pend_stride + idx] = 1;
    g.list[(size_t)cloud * g.cap + i] = idx;
    if (i == 0) g.count[cloud] = n0 * n0 * n0;
}

// Opens an update round: prev = the number of points the round has just evaluated, count = 0 (the queue mise_apply_kernel
// refills).  A cloud with prev == 0 has learnt nothing since its last update: its marks would be the ones already applied, so
// the mark / apply blocks of that cloud return at once - clouds that finish early stop costing anything, and a whole round
// without work (the host enqueues one round ahead of the counts it has seen, api.cpp) is a handful of empty launches.
__global__ void mise_begin_kernel(MiseGrid g, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    g.prev[b] = min(g.count[b], g.cap);
    g.count[b] = 0;
}

// mixed[level][voxel] = leaf && known points of the closed cube on both sides of the threshold
__global__ void mise_mark_kernel(MiseGrid g, int level) {
    const int cloud = blockIdx.y;
    if (g.prev[cloud] == 0) return;                                 // block-uniform
    const int nv = g.res0 << level;                                 // voxels per axis at this level
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv * nv * nv) return;
    const int vx = v / (nv * nv), vy = (v / nv) % nv, vz = v % nv;
    uint8_t* mix = g.mix + (size_t)cloud * g.sub_total + g.sub_off[level];
    const uint8_t* sub = g.sub + (size_t)cloud * g.sub_total;
    bool leaf = sub[g.sub_off[level] + v] == 0;
    if (leaf && level > 0) {
        const int np = nv >> 1;
        leaf = sub[g.sub_off[level - 1] + ((vx >> 1) * np + (vy >> 1)) * np + (vz >> 1)] != 0;      // the voxel exists
    }
    bool pos = false, neg = false;
    if (leaf) {
        const int s = 1 << (g.depth - level);
        const float* val = g.val + (size_t)cloud * g.P3;
        const uint8_t* known = g.known + (size_t)cloud * g.P3;
        for (int a = 0; a <= s; ++a)
            for (int b = 0; b <= s; ++b)
                for (int c = 0; c <= s; ++c) {
                    const int idx = ((vx * s + a) * g.P + vy * s + b) * g.P + vz * s + c;
                    if (known[idx]) {
                        const double f = (double)val[idx];
                        pos |= f >= g.threshold;
                        neg |= f <= g.threshold;
                    }
                }
    }
    mix[v] = (pos && neg) ? 1 : 0;
}

// subdivide the marked voxels: flag them, queue the not-yet-known points of the half-size lattice.  Queue slots are
// reserved with ONE atomic per wave (the per-thread claims are prefix-summed across the wave first): thousands of
// single-slot atomics on one counter per cloud serialise (measured: 16 % of the whole mesh path).
__global__ void mise_apply_kernel(MiseGrid g, int level) {
    const int cloud = blockIdx.y;
    if (g.prev[cloud] == 0) return;                                 // block-uniform (before the wave-level prefix sum)
    const int nv = g.res0 << level;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = v < nv * nv * nv && g.mix[(size_t)cloud * g.sub_total + g.sub_off[level] + v] != 0;
    int mine[27], n = 0;
    if (active) {
        g.sub[(size_t)cloud * g.sub_total + g.sub_off[level] + v] = 1;
        const int vx = v / (nv * nv), vy = (v / nv) % nv, vz = v % nv;
        const int s = 1 << (g.depth - level), h = s >> 1;
        const uint8_t* known = g.known + (size_t)cloud * g.P3;
        unsigned int* pend = reinterpret_cast<unsigned int*>(g.pend + (size_t)cloud * g.pend_stride);   // byte flags, word atomics
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int idx = ((vx * s + a * h) * g.P + vy * s + b * h) * g.P + vz * s + c * h;
                    bool claim = false;
                    if (!known[idx]) {
                        const unsigned int bit = 1u << (8 * (idx & 3));
                        claim = !(atomicOr(pend + (idx >> 2), bit) & bit);
                    }
                    mine[a * 9 + b * 3 + c] = claim ? idx : -1;
                    n += claim ? 1 : 0;
                }
    }
    // wave-inclusive prefix sum of n, one reservation per wave
    const int lane = threadIdx.x & 63;
    int incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const int total = __shfl(incl, 63);
    int base = 0;
    if (lane == 63 && total > 0) base = atomicAdd(g.count + cloud, total);
    base = __shfl(base, 63);
    int at = base + incl - n;
    if (n > 0) {
#pragma unroll
        for (int k = 0; k < 27; ++k)
            if (mine[k] >= 0) {
                if (at < g.cap) g.list[(size_t)cloud * g.cap + at] = mine[k];
                ++at;
            }
    }
}

// to_dense (mise.pyx:130-166): unknown entries take the value of their predecessor along x, then y, then z
__global__ void mise_fill_kernel(MiseGrid g, int axis) {
    const int cloud = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.P * g.P) return;
    const int u = t / g.P, w = t % g.P;
    float* val = g.val + (size_t)cloud * g.P3;
    uint8_t* known = g.known + (size_t)cloud * g.P3;
    const int stride = axis == 0 ? g.P * g.P : g.P;                  // x or y: neighbouring threads walk neighbouring z
    const int base = axis == 0 ? u * g.P + w : u * g.P * g.P + w;
    for (int i = 1; i < g.P; ++i) {
        const int idx = base + i * stride;
        if (!known[idx] && known[idx - stride]) { val[idx] = val[idx - stride]; known[idx] = 1; }
    }
}

// the z pass walks contiguous memory: 64 lines per block go through LDS so that global accesses stay coalesced
constexpr int FILLZ_LINES = 64;
__global__ __launch_bounds__(256) void mise_fill_z_kernel(MiseGrid g) {
    extern __shared__ float fz[];                                     // [FILLZ_LINES][P] values, then known bytes
    const int cloud = blockIdx.y, P = g.P;
    const int line0 = blockIdx.x * FILLZ_LINES, nlines = min(FILLZ_LINES, P * P - line0);
    if (nlines <= 0) return;
    uint8_t* kz = reinterpret_cast<uint8_t*>(fz + FILLZ_LINES * P);
    float* val = g.val + (size_t)cloud * g.P3 + (size_t)line0 * P;
    uint8_t* known = g.known + (size_t)cloud * g.P3 + (size_t)line0 * P;
    for (int i = threadIdx.x; i < nlines * P; i += 256) { fz[i] = val[i]; kz[i] = known[i]; }
    __syncthreads();
    if ((int)threadIdx.x < nlines) {
        float* l = fz + threadIdx.x * P;
        uint8_t* k = kz + threadIdx.x * P;
        for (int i = 1; i < P; ++i)
            if (!k[i] && k[i - 1]) { l[i] = l[i - 1]; k[i] = 1; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nlines * P; i += 256) { val[i] = fz[i]; known[i] = kz[i]; }
}

hipError_t launch_mise_init(const MiseGrid& g, int B, hipStream_t s) {
    hipError_t e = hipMemsetAsync(g.known, 0, (size_t)B * g.P3, s);
    if (e == hipSuccess) e = hipMemsetAsync(g.pend, 0, (size_t)B * g.pend_stride, s);
    if (e == hipSuccess) e = hipMemsetAsync(g.sub, 0, (size_t)B * g.sub_total, s);
    if (e != hipSuccess) return e;
    const int n0 = (g.res0 + 1) * (g.res0 + 1) * (g.res0 + 1);
    hipLaunchKernelGGL(mise_init_kernel, dim3((n0 + 255) / 256, B), dim3(256), 0, s, g);
    return hipGetLastError();
}

hipError_t launch_mise_update(const MiseGrid& g, int B, hipStream_t s) {
    // the evaluated points are known now; decisions of this round use the state at its start (mise.pyx:196-215:
    // marks first, then subdivision of the voxels that existed), so: reset the queue, mark every level, then apply
    hipLaunchKernelGGL(mise_begin_kernel, dim3((B + 255) / 256), dim3(256), 0, s, g, B);
    for (int l = 0; l < g.depth; ++l) {
        const int nv = g.res0 << l, n = nv * nv * nv;
        hipLaunchKernelGGL(mise_mark_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, g, l);
    }
    for (int l = 0; l < g.depth; ++l) {
        const int nv = g.res0 << l, n = nv * nv * nv;
        hipLaunchKernelGGL(mise_apply_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, g, l);
    }
    return hipGetLastError();
}

hipError_t launch_mise_fill(const MiseGrid& g, int B, hipStream_t s) {
    for (int axis = 0; axis < 2; ++axis)
        hipLaunchKernelGGL(mise_fill_kernel, dim3((g.P * g.P + 255) / 256, B), dim3(256), 0, s, g, axis);
    const size_t lds = (size_t)FILLZ_LINES * g.P * 5;
    hipLaunchKernelGGL(mise_fill_z_kernel, dim3((g.P * g.P + FILLZ_LINES - 1) / FILLZ_LINES, B), dim3(256), lds, s, g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// marching cubes on the padded grid (generation.py:155-178; libmcubes/marchingcubes.h:23-193)
//
// Corner / edge numbering and the inside test (value <= isovalue) are the reference's (= Bourke's): corners 0..3 =
// (0,0,0) (1,0,0) (1,1,0) (0,1,0), 4..7 the same at z + 1; edges 0-3 bottom ring, 4-7 top ring, 8-11 verticals.
// Vertices are the linear iso-crossings of the grid edges, identical to the reference's.  The polygonisation is the
// reference's as well, triangle for triangle: kMcTriTable (mc_table_data.h) records what its compiled libmcubes emits for
// each of the 256 sign configurations of a single cube - edge triples in its order and winding - obtained by running that
// library (scripts/probe_mc_table.py; the classic Lorensen-Cline table in this file's numbering).  Round 1 generated its
// own table by tracing the face loops and fanning them: the same vertices and triangle counts, but other diagonals in
// the fans, i.e. a different (if equivalent) surface.
// ---------------------------------------------------------------------------------------------
struct McTable {
    int8_t tri[256][16];          // up to 5 triangles as edge triples, -1 terminated
    uint8_t ntri[256];
};
__constant__ McTable c_mc;

static void mc_build_table(McTable& T) {
    for (int cfg = 0; cfg < 256; ++cfg) {
        int n = 0;
        for (int i = 0; i < 16; ++i) {
            T.tri[cfg][i] = kMcTriTable[cfg][i];
            if (i % 3 == 2 && kMcTriTable[cfg][i] >= 0) ++n;
        }
        T.ntri[cfg] = (uint8_t)n;
    }
}

hipError_t mc_upload_table() {
    static McTable T;
    static bool built = false;
    if (!built) { mc_build_table(T); built = true; }
    return hipMemcpyToSymbol(HIP_SYMBOL(c_mc), &T, sizeof(McTable));
}
void mc_host_table(int8_t (*tri)[16], uint8_t* ntri) {
    McTable T;
    mc_build_table(T);
    std::memcpy(tri, T.tri, sizeof(T.tri));
    std::memcpy(ntri, T.ntri, sizeof(T.ntri));
}

__device__ __forceinline__ float padded_value(const float* __restrict__ val, int P, int x, int y, int z) {
    // np.pad(occ_hat, 1, 'constant', constant_values=-1e6) (generation.py:168-169): padded index -> grid index - 1
    if (x < 1 || y < 1 || z < 1 || x > P || y > P || z > P) return -1e6f;
    return val[((x - 1) * P + (y - 1)) * P + (z - 1)];
}

__device__ __forceinline__ int cube_config(const float* __restrict__ val, int P, int x, int y, int z, double iso, double (&f)[8]) {
    const int ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    int cfg = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f[m] = (double)padded_value(val, P, x + ox[m], y + oy[m], z + oz[m]);
        if (f[m] <= iso) cfg |= 1 << m;                                // marchingcubes.h:62-64
    }
    return cfg;
}

// pass 1: triangles per cube (padded grid has P + 2 points -> P + 1 cubes per axis)
__global__ void mc_count_kernel(const float* __restrict__ val, int P, double iso, int* __restrict__ ntri) {
    const int cloud = blockIdx.y, NC = P + 1;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= NC * NC * NC) return;
    double f[8];
    const int cfg = cube_config(val + (size_t)cloud * P * P * P, P, c / (NC * NC), (c / NC) % NC, c % NC, iso, f);
    ntri[(size_t)cloud * NC * NC * NC + c] = c_mc.ntri[cfg];
}

// exclusive scan of the per-cube counts, one block per cloud (deterministic triangle order = cube order)
__global__ __launch_bounds__(1024) void mc_scan_kernel(int* __restrict__ ntri, int n, int* __restrict__ total) {
    __shared__ int part[1024];
    __shared__ int carry;
    int* a = ntri + (size_t)blockIdx.x * n;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024 * 8) {
        int v[8], sum = 0;
        const int i0 = base + threadIdx.x * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = i0 + j < n ? a[i0 + j] : 0; sum += v[j]; }
        part[threadIdx.x] = sum;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {                           // Hillis-Steele inclusive scan of the partials
            const int t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        int run = carry + part[threadIdx.x] - sum;
#pragma unroll
        for (int j = 0; j < 8; ++j) { if (i0 + j < n) a[i0 + j] = run; run += v[j]; }
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[blockIdx.x] = carry;
}

// The same exclusive scan in three phases over chunks of 2048 cubes (round 5: the one-block-per-cloud scan above walked 2.2 M
// counts in 268 steps of 22 barriers each, 64 workgroups on 256 CUs - 3.3 ms per 64 clouds; integer sums: the same offsets).
constexpr int SCAN_CHUNK = 2048;
__global__ __launch_bounds__(256) void mc_chunk_sum_kernel(const int* __restrict__ ntri, int n, int* __restrict__ sums, size_t sums_stride) {
    __shared__ int wsum[4];
    const int* a = ntri + (size_t)blockIdx.y * n;
    const int i0 = blockIdx.x * SCAN_CHUNK + threadIdx.x * 8;
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += i0 + j < n ? a[i0 + j] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) sums[(size_t)blockIdx.y * sums_stride + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(1024) void mc_chunk_scan_kernel(int* __restrict__ sums, size_t sums_stride, int nchunks, int* __restrict__ total) {
    __shared__ int part[1024];
    __shared__ int carry;
    int* a = sums + (size_t)blockIdx.x * sums_stride;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nchunks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nchunks ? a[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nchunks) a[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[blockIdx.x] = carry;
}
__global__ __launch_bounds__(256) void mc_chunk_apply_kernel(int* __restrict__ ntri, int n, const int* __restrict__ sums, size_t sums_stride) {
    __shared__ int wsum[4];
    int* a = ntri + (size_t)blockIdx.y * n;
    const int i0 = blockIdx.x * SCAN_CHUNK + threadIdx.x * 8, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int v[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = i0 + j < n ? a[i0 + j] : 0; sum += v[j]; }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = sums[(size_t)blockIdx.y * sums_stride + blockIdx.x] + incl - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int j = 0; j < 8; ++j) { if (i0 + j < n) a[i0 + j] = run; run += v[j]; }
}

// pass 2: emit the triangles (vertex coordinates in the decoder's frame, generation.py:171-176) and their areas.  One thread per
// TRIANGLE (round 5; one thread per cube left 95 % of every wave idle behind the two or three lanes whose cube is cut by the
// surface): triangle `at` of a cloud belongs to the last cube whose offset is <= at (cubes without triangles share their
// successor's offset and are stepped over by the search), t = at - offset of that cube.  Same arithmetic, same order.
__global__ void mc_emit_kernel(const float* __restrict__ val, int P, double iso, float box, const int* __restrict__ offs,
                               const int* __restrict__ ntri_total, int cap, float* __restrict__ tris, double* __restrict__ area) {
    const int cloud = blockIdx.y, NC = P + 1, ncube = NC * NC * NC;
    const int at = blockIdx.x * blockDim.x + threadIdx.x;
    if (at >= min(ntri_total[cloud], cap)) return;
    const int* of = offs + (size_t)cloud * ncube;
    int lo = 0, hi = ncube - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (of[mid] <= at) lo = mid; else hi = mid - 1;
    }
    const int c = lo, t = at - of[c];
    const int x = c / (NC * NC), y = (c / NC) % NC, z = c % NC;
    double f[8];
    const int cfg = cube_config(val + (size_t)cloud * P * P * P, P, x, y, z, iso, f);
    const int ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    const int ec[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    double p[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = c_mc.tri[cfg][3 * t + k], a = ec[e][0], b = ec[e][1];
        // mc_isovalue_interpolation (marchingcubes.cpp:290-297)
        const double w = f[b] == f[a] ? 0.5 : (iso - f[a]) / (f[b] - f[a]);
        const double gx = x + ox[a] + (ox[b] - ox[a]) * w, gy = y + oy[a] + (oy[b] - oy[a]) * w,
                     gz = z + oz[a] + (oz[b] - oz[a]) * w;
        // libmcubes' +0.5 shift undone, padding undone, normalised to the bounding box (generation.py:171-176)
        p[k][0] = (double)box * ((gx - 1.0) / (double)(P - 1) - 0.5);
        p[k][1] = (double)box * ((gy - 1.0) / (double)(P - 1) - 0.5);
        p[k][2] = (double)box * ((gz - 1.0) / (double)(P - 1) - 0.5);
    }
    float* o = tris + ((size_t)cloud * cap + at) * 9;
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[3 * k] = (float)p[k][0]; o[3 * k + 1] = (float)p[k][1]; o[3 * k + 2] = (float)p[k][2]; }
    const double ux = p[1][0] - p[0][0], uy = p[1][1] - p[0][1], uz = p[1][2] - p[0][2];
    const double vx = p[2][0] - p[0][0], vy = p[2][1] - p[0][1], vz = p[2][2] - p[0][2];
    const double cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
    area[(size_t)cloud * cap + at] = 0.5 * sqrt(cx * cx + cy * cy + cz * cz);
}

// inclusive scan of the triangle areas (double), one block per cloud
__global__ __launch_bounds__(1024) void area_scan_kernel(double* __restrict__ area, const int* __restrict__ ntri_total, int cap) {
    __shared__ double part[1024];
    __shared__ double carry;
    const int n = min(ntri_total[blockIdx.x], cap);
    double* a = area + (size_t)blockIdx.x * cap;
    if (threadIdx.x == 0) carry = 0.0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024 * 8) {
        double v[8], sum = 0.0;
        const int i0 = base + threadIdx.x * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = i0 + j < n ? a[i0 + j] : 0.0; sum += v[j]; }
        part[threadIdx.x] = sum;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const double t = threadIdx.x >= o ? part[threadIdx.x - o] : 0.0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        double run = carry + part[threadIdx.x] - sum;
#pragma unroll
        for (int j = 0; j < 8; ++j) { run += v[j]; if (i0 + j < n) a[i0 + j] = run; }
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
}

// Philox-4x32-10 (same generator as prep.hip), keyed by (seed, global cloud index), counter = sample index
__device__ __forceinline__ void philox4(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// trimesh.sample.sample_surface (remesh_defense.py:155-156): pick a face with probability ~ area, then a uniform
// point of it (two uniforms, reflected into the triangle).  out [B][n][3]; empty meshes leave their rows untouched.
__global__ void sample_surface_kernel(const float* __restrict__ tris, const double* __restrict__ cum_area,
                                      const int* __restrict__ ntri_total, int cap, int n, uint32_t seed_lo, uint32_t seed_hi,
                                      int cloud_base, float* __restrict__ out) {
    const int cloud = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nt = min(ntri_total[cloud], cap);
    if (i >= n || nt == 0) return;
    uint32_t r[4];
    philox4(seed_lo, seed_hi, (uint32_t)i, 0x5a3fu, (uint32_t)(cloud_base + cloud), 0u, r);
    const double* ca = cum_area + (size_t)cloud * cap;
    const double target = ((double)r[0] + (double)r[1] * 4294967296.0) * (1.0 / 18446744073709551616.0) * ca[nt - 1];
    int lo = 0, hi = nt - 1;                                           // first index with cum_area > target
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ca[mid] > target) hi = mid; else lo = mid + 1;
    }
    float u = (float)r[2] * (1.0f / 4294967296.0f), v = (float)r[3] * (1.0f / 4294967296.0f);
    if (u + v > 1.f) { u = 1.f - u; v = 1.f - v; }
    const float* t = tris + ((size_t)cloud * cap + lo) * 9;
    float* o = out + ((size_t)cloud * n + i) * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) o[a] = t[a] + u * (t[3 + a] - t[a]) + v * (t[6 + a] - t[a]);
}

hipError_t launch_marching_cubes(const float* val, int B, int P, double iso, float box, int* cube_offs, int* ntri_total,
                                 int cap, float* tris, double* area, hipStream_t s) {
    const int NC = P + 1, n = NC * NC * NC;
    hipLaunchKernelGGL(mc_count_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, val, P, iso, cube_offs);
    // chunk sums live in the area buffer, which nobody reads before mc_emit_kernel writes it (double[cap] per cloud = 2 cap ints)
    const int nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if ((size_t)nchunks <= (size_t)cap * 2) {
        int* sums = reinterpret_cast<int*>(area);
        const size_t stride = (size_t)cap * 2;
        hipLaunchKernelGGL(mc_chunk_sum_kernel, dim3(nchunks, B), dim3(256), 0, s, cube_offs, n, sums, stride);
        hipLaunchKernelGGL(mc_chunk_scan_kernel, dim3(B), dim3(1024), 0, s, sums, stride, nchunks, ntri_total);
        hipLaunchKernelGGL(mc_chunk_apply_kernel, dim3(nchunks, B), dim3(256), 0, s, cube_offs, n, sums, stride);
    } else {
        hipLaunchKernelGGL(mc_scan_kernel, dim3(B), dim3(1024), 0, s, cube_offs, n, ntri_total);
    }
    hipLaunchKernelGGL(mc_emit_kernel, dim3((cap + 255) / 256, B), dim3(256), 0, s, val, P, iso, box, cube_offs, ntri_total, cap, tris, area);
    hipLaunchKernelGGL(area_scan_kernel, dim3(B), dim3(1024), 0, s, area, ntri_total, cap);
    return hipGetLastError();
}

hipError_t launch_sample_surface(const float* tris, const double* cum_area, const int* ntri_total, int B, int cap, int n,
                                 uint64_t seed, int cloud_base, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sample_surface_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, tris, cum_area, ntri_total, cap, n,
                       (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), cloud_base, out);
    return hipGetLastError();
}

}  // namespace ifd
