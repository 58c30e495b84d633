// C ABI of libifd.so (see include/ifd.h).  Host-side only: context, weight re-packing, argument validation and kernel
// launches.  Never throws, never exits.  Every call runs on the context's own device (DeviceGuard).  Calls only enqueue
// work on the caller's stream, with these exceptions, which block the host: ifd_create / ifd_onet_create / ifd_destroy
// (allocation, upload), ifd_get_counters (device-to-host copy), ifd_onet_mesh_sample (the MISE loop is driven from the device; the
// host waits on an event per round - one round BEHIND what it has enqueued - only to learn when the queues have run empty), and any call that needs MORE context workspace than every
// earlier call on that context (the old buffer is freed after a device synchronisation; steady-state calls never do).
#include "../../include/ifd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

#include "ifd_internal.h"

// device-side size of the counter buffer: the IFD_N_COUNTERS public slots + the per-wave time stamps of -DIFD_TRACE builds
constexpr int N_COUNTERS_DEV = IFD_N_COUNTERS + 8 * 32;
static_assert(N_COUNTERS_DEV == ifd::DEV_COUNTERS, "the sticky status words sit right behind the counters (ifd_internal.h)");
constexpr int N_COUNTERS_ALLOC = ifd::DEV_COUNTERS_TOTAL;      // + overflow / timeout status words (never cleared by an optimise call)

using namespace ifd;

namespace {

thread_local std::string g_create_error;

// Offsets (in floats) of every tensor in the canonical weight order documented in ifd.h.
struct WeightMap {
    size_t dec_fc_p_w, dec_fc_p_b, dec_fc_c_w[5], dec_fc_c_b[5];
    size_t dec_fc0_w[5], dec_fc0_b[5], dec_fc1_w[5], dec_fc1_b[5], dec_out_w, dec_out_b;
    size_t enc_pos_w, enc_pos_b, enc_fc0_w[5], enc_fc0_b[5], enc_fc1_w[5], enc_fc1_b[5], enc_sc_w[5];
    size_t enc_fcc_w, enc_fcc_b;
    size_t down_w[4][2], down_b[4][2];
    size_t up_t_w[3], up_t_b[3], up_w[3][2], up_b[3][2], fin_w, fin_b;
    size_t total;
};

WeightMap make_weight_map() {
    WeightMap m{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += n; return r; };
    m.dec_fc_p_w = take(32 * 3); m.dec_fc_p_b = take(32);
    for (int i = 0; i < 5; ++i) { m.dec_fc_c_w[i] = take(1024); m.dec_fc_c_b[i] = take(32); }
    for (int i = 0; i < 5; ++i) {
        m.dec_fc0_w[i] = take(1024); m.dec_fc0_b[i] = take(32);
        m.dec_fc1_w[i] = take(1024); m.dec_fc1_b[i] = take(32);
    }
    m.dec_out_w = take(32); m.dec_out_b = take(1);
    m.enc_pos_w = take(64 * 3); m.enc_pos_b = take(64);
    for (int i = 0; i < 5; ++i) {
        m.enc_fc0_w[i] = take(32 * 64); m.enc_fc0_b[i] = take(32);
        m.enc_fc1_w[i] = take(32 * 32); m.enc_fc1_b[i] = take(32);
        m.enc_sc_w[i] = take(32 * 64);
    }
    m.enc_fcc_w = take(1024); m.enc_fcc_b = take(32);
    const int ch[4] = {32, 64, 128, 256};
    int cin = 32;
    for (int i = 0; i < 4; ++i) {
        m.down_w[i][0] = take((size_t)ch[i] * cin * 9); m.down_b[i][0] = take(ch[i]);
        m.down_w[i][1] = take((size_t)ch[i] * ch[i] * 9); m.down_b[i][1] = take(ch[i]);
        cin = ch[i];
    }
    for (int i = 0; i < 3; ++i) {
        const int co = cin / 2;
        m.up_t_w[i] = take((size_t)cin * co * 4); m.up_t_b[i] = take(co);
        m.up_w[i][0] = take((size_t)co * 2 * co * 9); m.up_b[i][0] = take(co);
        m.up_w[i][1] = take((size_t)co * co * 9); m.up_b[i][1] = take(co);
        cin = co;
    }
    m.fin_w = take(32 * 32); m.fin_b = take(32);
    m.total = o;
    return m;
}

const WeightMap& wmap() {
    static const WeightMap m = make_weight_map();
    return m;
}

}  // namespace

constexpr int LARGE_MAX_GROUPS = 4;   // stream groups of the launch-per-step path (large_optimize_in_groups)

struct ifd_ctx {
    int device = 0;
    ifd_config cfg{};
    std::vector<float> w;          // host copy, canonical order
    float* d_dec_img = nullptr;    // decoder parameter image (ifd_device.h layout)
    float* d_dec_img_bf = nullptr; // ... and the bf16 piece image of the split-precision tiles (ifd_opt_params.precision 1 / 2)
    float* d_dec_img_opt = nullptr;// ... the persistent optimiser's copy: fc_0 / fc_1 / fc_out scaled by 2^RELU_K (ifd_device.h)
    float* d_w = nullptr;          // the whole canonical weight vector on the device (encoder kernels index it)
    EncPointOffsets eo{};
    float* d_unet = nullptr;       // re-packed U-Net weights ([tap][Cin][Cout])
    float* d_enc_img = nullptr;    // aligned copy of the point-net weights (encoder.hip build_enc_image)
    UNetWeights uw{};
    void* ws_enc = nullptr;        // encoder scratch (pre-U-Net planes + U-Net activations), grown on demand
    size_t ws_enc_bytes = 0;
    void* ws_fold = nullptr;       // ONet: folded CBN coefficients of the clouds of the current decode / optimise / mesh call (onet_fold) -
    size_t ws_fold_bytes = 0;      // NOT the encoder scratch: the optimiser reads them for its whole launch while an encoder call may run beside it
    DecConst dc{};
    unsigned long long* d_counters = nullptr;   // IFD_N_COUNTERS diagnostic counters of the last ifd_optimize
    int n_cu = 256;                // compute units of the device (rounds of the persistent optimiser)
    void* ws = nullptr;            // context-owned scratch (kNN lists, encoder activations), grown on demand
    size_t ws_bytes = 0;
    void* adam_tab = nullptr;      // per-step Adam bias corrections of the current optimise call (launch_adam_table)
    size_t adam_bytes = 0;
    // ONet-Opt variant (ifd_onet_create): padded copy of the canonical weights, fragment-ordered decoder layer images
    // (10 forward + 10 transposed), the small decoder parameters, tensor offsets into d_w
    int model = IFD_MODEL_CONVONET;
    float* d_onet_img = nullptr;
    float* d_onet_img_bf = nullptr; // ... the same layers as bf16 pieces (split precision, onet_fragment_image_bf)
    float* d_onet_small = nullptr;
    OnetEncOffsets oe{};
    OnetDecOffsets od{};
    void* ws_mesh = nullptr;       // ONet-Mesh scratch (MISE arrays, triangle soup)
    unsigned long long mesh_points = 0, mesh_rounds = 0;   // grid points evaluated / MISE rounds of the last mesh call
    size_t ws_mesh_bytes = 0;
    int* h_mesh_counts = nullptr;  // pinned: the queue lengths of the last two MISE rounds (ifd_onet_mesh_sample reads them one round late)
    size_t h_mesh_counts_n = 0;    // ints per slot
    hipEvent_t mesh_ev[2] = {nullptr, nullptr};
    // read from the environment once, at creation (read_opt_env): bound of the cross-CU waits of split clouds, test hook
    unsigned int coop_timeout_ticks = 3000000000u;
    int test_drop_member = -1;
    int test_large_groups = 0;     // measurement hook: stream groups of the launch-per-step path (0 = automatic; large_groups_of)
    // clouds of more than 1024 points: side streams + fork / join events of the stream groups (run_large_in_groups), created on first use
    hipStream_t large_side[LARGE_MAX_GROUPS - 1] = {};
    hipEvent_t large_fork = nullptr, large_join[LARGE_MAX_GROUPS - 1] = {};
    int test_no_morton = 0;        // measurement hook: ifd_prepare leaves the optimised points in draw order (the locality A/B through the whole pipeline)
    unsigned long long* d_status_out = nullptr;   // ifd_optimize_status: the status words as taken (atomic exchange) by status_take_kernel
    std::string err;
};

namespace {

// A context is bound to one device (include/ifd.h).  Every entry point that takes a context makes that device current for
// the duration of the call - launches, workspace allocations and the rare synchronisations all go to ctx->device whatever
// the caller's current device is - and restores the caller's device on the way out.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(const ifd_ctx* ctx);
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int fail(ifd_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    }
    return code;
}

DeviceGuard::DeviceGuard(const ifd_ctx* ctx) {
    int cur = -1;
    if (!ctx || hipGetDevice(&cur) != hipSuccess) { ok = ctx != nullptr; return; }
    if (cur != ctx->device) {
        if (hipSetDevice(ctx->device) != hipSuccess) { ok = false; return; }
        prev = cur;
    }
}
#define IFD_ON_CTX_DEVICE(ctx)                                  \
    DeviceGuard ifd_device_guard_(ctx);                          \
    if (!ifd_device_guard_.ok) return fail(ctx, IFD_ERR_HIP, "cannot make the context's device current")

// Build the LDS image of the decoder parameters.
std::vector<float> build_dec_image(const float* w, bool relu_scaled = false) {
    const WeightMap& m = wmap();
    std::vector<float> img(DEC_FLOATS, 0.f);
    auto put_layer = [&](int L, size_t woff, size_t boff) {
        for (int o = 0; o < 32; ++o)
            for (int k = 0; k < 32; ++k) img[DEC_OFF_W + L * W_LAYER + o * W_STRIDE + wperm(k)] = w[woff + o * 32 + k];
        for (int o = 0; o < 32; ++o) img[DEC_OFF_BIAS + L * 32 + o] = w[boff + o];
    };
    for (int i = 0; i < 5; ++i) {
        put_layer(3 * i + 0, m.dec_fc_c_w[i], m.dec_fc_c_b[i]);
        put_layer(3 * i + 1, m.dec_fc0_w[i], m.dec_fc0_b[i]);
        put_layer(3 * i + 2, m.dec_fc1_w[i], m.dec_fc1_b[i]);
    }
    for (int c = 0; c < 32; ++c) {
        for (int a = 0; a < 3; ++a) img[DEC_OFF_WP + c * 4 + a] = w[m.dec_fc_p_w + c * 3 + a];
        img[DEC_OFF_WP + c * 4 + 3] = w[m.dec_fc_p_b + c];
        img[DEC_OFF_WOUT + c] = w[m.dec_out_w + c];
    }
    img[DEC_OFF_BOUT] = w[m.dec_out_b];
    // The bias of fc_c[i] only ever enters through a_i = n_i + fc_c[i](c), and n_i only through a_i: fold it into the bias
    // that produced n_i (fc_p's for i = 0, blocks[i-1].fc_1's otherwise), so that the MFMA chain of fc_c[i] starts from
    // the accumulators of n_i as they are (one vector add per layer and point saved in the tiles).
    for (int c = 0; c < 32; ++c) {
        img[DEC_OFF_WP + c * 4 + 3] += img[DEC_OFF_BIAS + 0 * 32 + c];
        img[DEC_OFF_BIAS + 0 * 32 + c] = 0.f;
        for (int i = 1; i < 5; ++i) {
            img[DEC_OFF_BIAS + (3 * (i - 1) + 2) * 32 + c] += img[DEC_OFF_BIAS + 3 * i * 32 + c];
            img[DEC_OFF_BIAS + 3 * i * 32 + c] = 0.f;
        }
    }
    if (relu_scaled && RELU_K != 0) {      // the layers fed by a ReLU (exact: a power of two; biases stay as they are)
        const float sc = RELU_UP;
        for (int i = 0; i < 5; ++i)
            for (int j = 1; j <= 2; ++j)
                for (int k = 0; k < W_LAYER; ++k) img[DEC_OFF_W + (3 * i + j) * W_LAYER + k] *= sc;
        for (int c = 0; c < 32; ++c) img[DEC_OFF_WOUT + c] *= sc;
    }
    return img;
}

// The same parameters as bf16 pieces in MFMA operand order for the split-precision tiles (ifd_device.h, tile_bf.h).
uint16_t bf16_rne(float f) {                        // round to nearest even, like v_cvt_pk_bf16_f32 (finite inputs)
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf16_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
std::vector<unsigned char> build_dec_image_bf(const float* w) {
    const std::vector<float> f32img = build_dec_image(w);          // source of the folded biases (and of the layer order)
    std::vector<unsigned char> img(BF_IMG_BYTES, 0);
    uint16_t* h = reinterpret_cast<uint16_t*>(img.data());
    for (int L = 0; L < 15; ++L)
        for (int mt = 0; mt < 2; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int m = lane & 15, g = lane >> 4;
                    const float x = f32img[DEC_OFF_W + L * W_LAYER + (16 * mt + m) * W_STRIDE + wperm(bf_chan(g, j))];
                    const uint16_t h1 = bf16_rne(x);
                    const float r1 = x - bf16_to_f32(h1);           // exact
                    const uint16_t h2 = bf16_rne(r1);
                    const float r2 = r1 - bf16_to_f32(h2);          // exact
                    const uint16_t hs[3] = {h1, h2, bf16_rne(r2)};
                    for (int sp = 0; sp < 3; ++sp)
                        h[(size_t)(L * BF_LAYER_BYTES + sp * BF_PIECE_BYTES + (mt * 64 + lane) * BF_ENTRY_BYTES) / 2 + j] = hs[sp];
                }
    float* f = reinterpret_cast<float*>(img.data());
    std::memcpy(f + BF_OFF_BIAS / 4, f32img.data() + DEC_OFF_BIAS, 15 * 32 * sizeof(float));
    std::memcpy(f + BF_OFF_WP / 4, f32img.data() + DEC_OFF_WP, 32 * 4 * sizeof(float));
    std::memcpy(f + BF_OFF_WOUT / 4, f32img.data() + DEC_OFF_WOUT, 32 * sizeof(float));
    f[BF_OFF_BOUT / 4] = f32img[DEC_OFF_BOUT];
    return img;
}

bool bad_bk(int B, int K) { return B < 1 || K < 6 || K > LARGE_MAXK; }

// Bound of the cross-CU waits of split clouds (knn_device.h coop_wait) and the test hook that provokes a time-out.  Environment,
// not ifd_opt_params: neither is part of the path's interface.  Both are read ONCE, when the context is created (round-4 advisor:
// no getenv on the path of every optimise call, and a stray variable must not be able to break a production context):
//   IFD_COOP_TIMEOUT_MS   default 30000 - a wait normally lasts microseconds and a whole launch under a second, but the bound is
//                         wall-clock time (s_memrealtime): a queue that is descheduled (another process time-slicing the GPU, a
//                         profiler or debugger halt) keeps spending it;
//   IFD_TEST_COOP_DROP=<member>  that member of every split cloud never arrives - honoured only together with
//                         IFD_ENABLE_TEST_HOOKS=1 (tests/test_gpu_parity.py::test_split_cloud_wait_is_bounded_and_reported).
struct OptEnv {
    unsigned int coop_timeout_ticks;
    int test_drop_member;
    int test_no_morton;
    int test_large_groups;
};
// the two sticky status words (overflow, time-out), taken and cleared atomically
__global__ void status_take_kernel(unsigned long long* __restrict__ st, unsigned long long* __restrict__ out) {
    if (threadIdx.x < 2) out[threadIdx.x] = atomicExch(st + threadIdx.x, 0ull);
}
OptEnv read_opt_env() {
    OptEnv o;
    double ms = 30000.0;
    if (const char* t = std::getenv("IFD_COOP_TIMEOUT_MS")) { const double x = std::atof(t); if (x > 0.0) ms = x; }
    o.coop_timeout_ticks = (unsigned int)std::min(4.0e9, ms * 1.0e5);          // 100 MHz wall clock (s_memrealtime)
    o.test_drop_member = -1;
    o.test_no_morton = 0;
    o.test_large_groups = 0;
    const char* en = std::getenv("IFD_ENABLE_TEST_HOOKS");
    if (en != nullptr && en[0] == '1')
    {
        if (const char* d = std::getenv("IFD_TEST_COOP_DROP")) o.test_drop_member = std::atoi(d);
        if (const char* d = std::getenv("IFD_TEST_NO_MORTON")) o.test_no_morton = d[0] == '1' ? 1 : 0;
        if (const char* d = std::getenv("IFD_TEST_LARGE_GROUPS")) o.test_large_groups = std::atoi(d);
    }
    return o;
}

// Grow the context workspace.  Growing synchronises the device (hipFree), which only happens when a call
// needs more scratch than any earlier call on this context.
hipError_t ensure_buf(void** buf, size_t* have, size_t bytes) {
    if (bytes <= *have) return hipSuccess;
    if (*buf) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) return e;
        (void)hipFree(*buf);
        *buf = nullptr;
        *have = 0;
    }
    hipError_t e = hipMalloc(buf, bytes);
    if (e == hipSuccess) *have = bytes;
    return e;
}
hipError_t ensure_ws(ifd_ctx* ctx, size_t bytes) { return ensure_buf(&ctx->ws, &ctx->ws_bytes, bytes); }

// Re-pack the U-Net convolutions to [tap][Cin][Cout] (Conv2d weight [Cout][Cin][kh][kw]; ConvTranspose2d weight
// [Cin][Cout][kh][kw]) and record where each tensor starts.
struct UNetPack {
    std::vector<float> data;
    size_t down_w[4][2], down_b[4][2], up_t_w[3], up_t_b[3], up_w[3][2], up_b[3][2], fin_w, fin_b;
    size_t down_u[4][2], up_u[3][2];         // Winograd-domain copies of the 3x3 layers
};
UNetPack pack_unet(const float* w) {
    const WeightMap& m = wmap();
    UNetPack P;
    auto conv = [&](size_t src, int co, int ci, int k) {
        size_t at = P.data.size();
        P.data.resize(at + (size_t)k * k * ci * co);
        for (int o = 0; o < co; ++o)
            for (int i = 0; i < ci; ++i)
                for (int t = 0; t < k * k; ++t) P.data[at + ((size_t)t * ci + i) * co + o] = w[src + ((size_t)o * ci + i) * k * k + t];
        return at;
    };
    // Winograd F(2x2, 3x3) filter transform U = G g G^T (G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]]), in
    // double, rounded once; layout [Cin / 16][xi = 4 i + j][Cout][16]: the slab of one 16-channel chunk and a range of output
    // channels is 16 contiguous pieces, and a lane's four k-steps of one MFMA group are 16 contiguous bytes (unet.hip)
    auto wino = [&](size_t src, int co, int ci) {
        static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        size_t at = P.data.size();
        P.data.resize(at + (size_t)16 * ci * co);
        for (int o = 0; o < co; ++o)
            for (int i = 0; i < ci; ++i) {
                const float* g = w + src + ((size_t)o * ci + i) * 9;
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        double u = 0.0;
                        for (int y = 0; y < 3; ++y)
                            for (int x = 0; x < 3; ++x) u += G[a][y] * (double)g[3 * y + x] * G[b][x];
                        P.data[at + ((((size_t)(i / 16) * 16 + (4 * a + b)) * co + o) * 16) + i % 16] = (float)u;
                    }
            }
        return at;
    };
    auto convt = [&](size_t src, int ci, int co) {
        size_t at = P.data.size();
        P.data.resize(at + (size_t)4 * ci * co);
        for (int i = 0; i < ci; ++i)
            for (int o = 0; o < co; ++o)
                for (int t = 0; t < 4; ++t) P.data[at + ((size_t)t * ci + i) * co + o] = w[src + ((size_t)i * co + o) * 4 + t];
        return at;
    };
    auto vec = [&](size_t src, int n) {
        size_t at = P.data.size();
        P.data.insert(P.data.end(), w + src, w + src + n);
        return at;
    };
    const int ch[4] = {32, 64, 128, 256};
    int cin = 32;
    for (int i = 0; i < 4; ++i) {
        P.down_w[i][0] = conv(m.down_w[i][0], ch[i], cin, 3); P.down_b[i][0] = vec(m.down_b[i][0], ch[i]);
        P.down_w[i][1] = conv(m.down_w[i][1], ch[i], ch[i], 3); P.down_b[i][1] = vec(m.down_b[i][1], ch[i]);
        P.down_u[i][0] = wino(m.down_w[i][0], ch[i], cin);
        P.down_u[i][1] = wino(m.down_w[i][1], ch[i], ch[i]);
        cin = ch[i];
    }
    for (int i = 0; i < 3; ++i) {
        const int co = cin / 2;
        P.up_t_w[i] = convt(m.up_t_w[i], cin, co); P.up_t_b[i] = vec(m.up_t_b[i], co);
        P.up_w[i][0] = conv(m.up_w[i][0], co, 2 * co, 3); P.up_b[i][0] = vec(m.up_b[i][0], co);
        P.up_w[i][1] = conv(m.up_w[i][1], co, co, 3); P.up_b[i][1] = vec(m.up_b[i][1], co);
        P.up_u[i][0] = wino(m.up_w[i][0], co, 2 * co);
        P.up_u[i][1] = wino(m.up_w[i][1], co, co);
        cin = co;
    }
    P.fin_w = conv(m.fin_w, 32, 32, 1); P.fin_b = vec(m.fin_b, 32);
    return P;
}


// ---------------------------------------------------------------------------------------------
// ONet-Opt: canonical weight order (include/ifd.h) = the reference checkpoint's state_dict order without the
// num_batches_tracked scalars: decoder.* then encoder.*
// ---------------------------------------------------------------------------------------------
struct OnetTensor { size_t host_off, dev_off, n; };
struct OnetMap {
    OnetTensor fc_p_w, fc_p_b, cbn[11][6] /* gamma.w, gamma.b, beta.w, beta.b, mean, var */, fc0_w[5], fc0_b[5], fc1_w[5],
        fc1_b[5], out_w, out_b;
    OnetTensor pos_w, pos_b, e_fc0_w[5], e_fc0_b[5], e_fc1_w[5], e_fc1_b[5], e_sc_w[5], fcc_w, fcc_b;
    size_t host_total, dev_total;
};
OnetMap make_onet_map() {
    OnetMap m{};
    size_t ho = 0, dv = 0;
    auto take = [&](size_t n) {
        OnetTensor t{ho, dv, n};
        ho += n;
        dv += (n + 3) & ~(size_t)3;          // device copy: every tensor starts on a 16-byte boundary
        return t;
    };
    auto cbn = [&](int i) {
        m.cbn[i][0] = take((size_t)ONET_H * ONET_C); m.cbn[i][1] = take(ONET_H);
        m.cbn[i][2] = take((size_t)ONET_H * ONET_C); m.cbn[i][3] = take(ONET_H);
        m.cbn[i][4] = take(ONET_H); m.cbn[i][5] = take(ONET_H);
    };
    m.fc_p_w = take(ONET_H * 3); m.fc_p_b = take(ONET_H);
    for (int i = 0; i < 5; ++i) {
        cbn(2 * i); cbn(2 * i + 1);
        m.fc0_w[i] = take((size_t)ONET_H * ONET_H); m.fc0_b[i] = take(ONET_H);
        m.fc1_w[i] = take((size_t)ONET_H * ONET_H); m.fc1_b[i] = take(ONET_H);
    }
    cbn(10);
    m.out_w = take(ONET_H); m.out_b = take(1);
    const size_t H = ONET_ENC_H;
    m.pos_w = take(2 * H * 3); m.pos_b = take(2 * H);
    for (int i = 0; i < 5; ++i) {
        m.e_fc0_w[i] = take(H * 2 * H); m.e_fc0_b[i] = take(H);
        m.e_fc1_w[i] = take(H * H); m.e_fc1_b[i] = take(H);
        m.e_sc_w[i] = take(H * 2 * H);
    }
    m.fcc_w = take((size_t)ONET_C * H); m.fcc_b = take(ONET_C);
    m.host_total = ho; m.dev_total = dv;
    return m;
}
const OnetMap& omap() {
    static const OnetMap m = make_onet_map();
    return m;
}

// Fragment-ordered image of one 256x256 layer for the 16x16x4 MFMA A operand (onet.hip): [tile t][4 k-steps s4]
// [lane][4]; lane (m = l & 15, q = l >> 4) of k-step s = 4 s4 + j multiplies input channel 16 (s >> 2) + 4 q + (s & 3).
void onet_fragment_image(const float* W, bool transposed, float* img) {
    for (int t = 0; t < 16; ++t)
        for (int s4 = 0; s4 < 16; ++s4)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int m = 16 * t + (lane & 15), q = lane >> 4, s = 4 * s4 + j;
                    const int k = 16 * (s >> 2) + 4 * q + (s & 3);
                    img[(((size_t)t * 16 + s4) * 64 + lane) * 4 + j] = transposed ? W[(size_t)k * ONET_H + m] : W[(size_t)m * ONET_H + k];
                }
}

// The same layer as bf16 PIECES for the 16x16x32 bf16 MFMA A operand (onet_kernel.h, split precision): w = w1 + w2 + w3 exactly
// (bf16_rne of what the earlier pieces left).  Chunks of 24 KB = [8 output tiles t8][3 pieces][64 lanes][8 k-slots]: lane (m = l & 15,
// g = l >> 4), slot j of k-step s multiplies input value 8 s + j of the lane that holds it = channel 32 s + 16 (j >> 2) + 4 g + (j & 3);
// output tile t = 8 h + t8, row m.  Chunk order: k-step-major (2 s + h), or output-half-major (8 h + s) for the images the backward
// chain sweeps twice (W0^T).
void onet_fragment_image_bf(const float* W, bool transposed, bool half_major, uint16_t* img) {
    for (int s = 0; s < 8; ++s)
        for (int h = 0; h < 2; ++h) {
            uint16_t* chunk = img + (size_t)(half_major ? 8 * h + s : 2 * s + h) * (8 * 3 * 64 * 8);
            for (int t8 = 0; t8 < 8; ++t8)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int m = 16 * (8 * h + t8) + (lane & 15), g = lane >> 4;
                        const int k = 32 * s + 16 * (j >> 2) + 4 * g + (j & 3);
                        float rest = transposed ? W[(size_t)k * ONET_H + m] : W[(size_t)m * ONET_H + k];
                        for (int pc = 0; pc < 3; ++pc) {
                            const uint16_t b = bf16_rne(rest);
                            chunk[((size_t)(t8 * 3 + pc) * 64 + lane) * 8 + j] = b;
                            uint32_t bits = (uint32_t)b << 16;
                            float piece;
                            std::memcpy(&piece, &bits, 4);
                            rest -= piece;                       // exact: the piece shares the leading bits of what it was rounded from
                        }
                    }
        }
}

}  // namespace

extern "C" {

int ifd_abi_version(void) { return IFD_ABI_VERSION; }

size_t ifd_weight_count(void) { return wmap().total; }

const char* ifd_last_error(const ifd_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

ifd_ctx* ifd_create(const float* weights_host, size_t n_weights, const ifd_config* cfg, int device) {
    g_create_error.clear();
    if (!weights_host || !cfg) { g_create_error = "ifd_create: NULL argument"; return nullptr; }
    if (n_weights != wmap().total) {
        g_create_error = "ifd_create: expected " + std::to_string(wmap().total) + " weights, got " +
                         std::to_string(n_weights);
        return nullptr;
    }
    if (cfg->struct_size != (int32_t)sizeof(ifd_config) || cfg->plane_resolution != RES || cfg->c_dim != CH ||
        cfg->hidden_dim != CH || cfg->n_blocks != NBLK || cfg->unet_depth != 4 || cfg->unet_start_filts != 32) {
        g_create_error = "ifd_create: only the shipped 3-plane config (res 64, c_dim 32, hidden 32, 5 blocks, "
                         "U-Net depth 4 / 32 filters) is supported";
        return nullptr;
    }
    ifd_ctx* ctx = new (std::nothrow) ifd_ctx();
    if (!ctx) { g_create_error = "ifd_create: out of host memory"; return nullptr; }
    ctx->device = device;
    ctx->cfg = *cfg;
    ctx->w.assign(weights_host, weights_host + n_weights);
    ctx->dc.sdiv = (float)(1.0 + (double)cfg->padding + 10e-6);
    ctx->dc.uclamp = (float)(1.0 - 10e-6);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) {
        int n_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0) ctx->n_cu = n_cu;
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_dec_img), DEC_FLOATS * sizeof(float));
    if (e == hipSuccess) {
        std::vector<float> img = build_dec_image(ctx->w.data());
        e = hipMemcpy(ctx->d_dec_img, img.data(), DEC_FLOATS * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_dec_img_opt), DEC_FLOATS * sizeof(float));
        if (e == hipSuccess) {
            std::vector<float> img2 = build_dec_image(ctx->w.data(), true);
            e = hipMemcpy(ctx->d_dec_img_opt, img2.data(), DEC_FLOATS * sizeof(float), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_dec_img_bf), BF_IMG_BYTES);
        if (e == hipSuccess) {
            std::vector<unsigned char> img3 = build_dec_image_bf(ctx->w.data());
            e = hipMemcpy(ctx->d_dec_img_bf, img3.data(), BF_IMG_BYTES, hipMemcpyHostToDevice);
        }
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_counters), N_COUNTERS_ALLOC * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ctx->d_counters, 0, N_COUNTERS_ALLOC * sizeof(unsigned long long));
    if (e == hipSuccess && !ctx->d_status_out) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_status_out), 2 * sizeof(unsigned long long));
    { const OptEnv oe = read_opt_env(); ctx->coop_timeout_ticks = oe.coop_timeout_ticks; ctx->test_drop_member = oe.test_drop_member; ctx->test_no_morton = oe.test_no_morton; ctx->test_large_groups = oe.test_large_groups; }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_w), n_weights * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_w, ctx->w.data(), n_weights * sizeof(float), hipMemcpyHostToDevice);
    {
        const WeightMap& m = wmap();
        EncPointOffsets& o = ctx->eo;
        o.pos_w = (int)m.enc_pos_w; o.pos_b = (int)m.enc_pos_b; o.fcc_w = (int)m.enc_fcc_w; o.fcc_b = (int)m.enc_fcc_b;
        for (int i = 0; i < 5; ++i) {
            o.fc0_w[i] = (int)m.enc_fc0_w[i]; o.fc0_b[i] = (int)m.enc_fc0_b[i];
            o.fc1_w[i] = (int)m.enc_fc1_w[i]; o.fc1_b[i] = (int)m.enc_fc1_b[i];
            o.sc_w[i] = (int)m.enc_sc_w[i];
        }
    }
    if (e == hipSuccess) {
        UNetPack P = pack_unet(ctx->w.data());
        e = hipMalloc(reinterpret_cast<void**>(&ctx->d_unet), P.data.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(ctx->d_unet, P.data.data(), P.data.size() * sizeof(float), hipMemcpyHostToDevice);
        const float* d = ctx->d_unet;
        UNetWeights& u = ctx->uw;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 2; ++j) { u.down_w[i][j] = d + P.down_w[i][j]; u.down_b[i][j] = d + P.down_b[i][j]; u.down_u[i][j] = d + P.down_u[i][j]; }
        for (int i = 0; i < 3; ++i) {
            u.up_t_w[i] = d + P.up_t_w[i]; u.up_t_b[i] = d + P.up_t_b[i];
            for (int j = 0; j < 2; ++j) { u.up_w[i][j] = d + P.up_w[i][j]; u.up_b[i][j] = d + P.up_b[i][j]; u.up_u[i][j] = d + P.up_u[i][j]; }
        }
        u.fin_w = d + P.fin_w; u.fin_b = d + P.fin_b;
    }
    if (e == hipSuccess) {
        std::vector<float> img((size_t)enc_image_floats());
        build_enc_image(ctx->w.data(), ctx->eo, img.data());
        e = hipMalloc(reinterpret_cast<void**>(&ctx->d_enc_img), img.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(ctx->d_enc_img, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = configure_encoder_kernels();
    if (e == hipSuccess) e = configure_unet_kernels();
    if (e == hipSuccess) e = configure_prep_kernels();
    if (e == hipSuccess) e = configure_optimize_kernels();
    if (e == hipSuccess) e = configure_decode_bf_kernels();
    if (e != hipSuccess) {
        g_create_error = std::string("ifd_create: ") + hipGetErrorString(e);
        if (ctx->d_dec_img) (void)hipFree(ctx->d_dec_img);
        if (ctx->d_dec_img_opt) (void)hipFree(ctx->d_dec_img_opt);
        if (ctx->d_dec_img_bf) (void)hipFree(ctx->d_dec_img_bf);
        if (ctx->d_counters) (void)hipFree(ctx->d_counters);
        if (ctx->d_w) (void)hipFree(ctx->d_w);
        if (ctx->d_unet) (void)hipFree(ctx->d_unet);
        if (ctx->d_enc_img) (void)hipFree(ctx->d_enc_img);
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void ifd_destroy(ifd_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx);
    if (ctx->d_dec_img) (void)hipFree(ctx->d_dec_img);
    if (ctx->d_dec_img_opt) (void)hipFree(ctx->d_dec_img_opt);
    if (ctx->d_dec_img_bf) (void)hipFree(ctx->d_dec_img_bf);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->adam_tab) (void)hipFree(ctx->adam_tab);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_status_out) (void)hipFree(ctx->d_status_out);
    if (ctx->d_w) (void)hipFree(ctx->d_w);
    if (ctx->d_unet) (void)hipFree(ctx->d_unet);
    if (ctx->d_enc_img) (void)hipFree(ctx->d_enc_img);
    if (ctx->ws_enc) (void)hipFree(ctx->ws_enc);
    if (ctx->ws_fold) (void)hipFree(ctx->ws_fold);
    if (ctx->d_onet_img) (void)hipFree(ctx->d_onet_img);
    if (ctx->d_onet_img_bf) (void)hipFree(ctx->d_onet_img_bf);
    if (ctx->d_onet_small) (void)hipFree(ctx->d_onet_small);
    if (ctx->ws_mesh) (void)hipFree(ctx->ws_mesh);
    if (ctx->h_mesh_counts) (void)hipHostFree(ctx->h_mesh_counts);
    for (hipEvent_t ev : ctx->mesh_ev) if (ev) (void)hipEventDestroy(ev);
    for (hipStream_t st : ctx->large_side) if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : ctx->large_join) if (ev) (void)hipEventDestroy(ev);
    if (ctx->large_fork) (void)hipEventDestroy(ctx->large_fork);
    delete ctx;
}

int ifd_sor(ifd_ctx* ctx, const float* pc, int B, int K, int k, float alpha, uint8_t* keep_mask, double* value,
            void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (!pc || !keep_mask || B < 1 || K < 2 || K > PREP_MAXK || k < 1 || k > 7 || k >= K)
        return fail(ctx, IFD_ERR_ARG, "ifd_sor: bad argument (2 <= K <= 10000, 1 <= k <= 7)");
    hipError_t e = launch_sor(pc, B, K, k, (double)alpha, keep_mask, value, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_sor launch", e);
}

int ifd_prepare(ifd_ctx* ctx, const float* pc, const uint8_t* keep_mask, int B, int K, const ifd_prep_params* prm,
                const int32_t* sel_idx, const int32_t* init_idx, const float* noise, float* sel, int32_t* t_per_cloud,
                float* init_points, int32_t* n_kept, float* proc, void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (!pc || !prm || prm->struct_size != (int32_t)sizeof(ifd_prep_params) || !sel || !t_per_cloud || !init_points ||
        B < 1 || K < 1 || K > PREP_MAXK || prm->n_sel < 1 || prm->n_sel > 1024 || prm->n_opt < 1)
        return fail(ctx, IFD_ERR_ARG, "ifd_prepare: bad argument (K <= 10000, n_sel <= 1024)");
    PrepArgs a;
    a.cloud_base = (int)prm->cloud_index_base; a.n_sel = prm->n_sel; a.n_opt = prm->n_opt;
    a.padding_scale = prm->padding_scale; a.init_sigma = prm->init_sigma;
    a.seed_lo = (uint32_t)(prm->seed & 0xffffffffu); a.seed_hi = (uint32_t)(prm->seed >> 32);
    a.no_morton = ctx->test_no_morton;
    hipError_t e = launch_prepare(pc, keep_mask, B, K, a, sel_idx, init_idx, noise, sel, t_per_cloud, init_points, n_kept,
                                  proc, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_prepare launch", e);
}

int ifd_encode_points(ifd_ctx* ctx, const float* sel, const int32_t* t_per_cloud, int B, int Tmax, float* planes_pre,
                      float* c_points, void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (ctx->model != IFD_MODEL_CONVONET) return fail(ctx, IFD_ERR_ARG, "ifd_encode_points: not a ConvONet context");
    if (!sel || !planes_pre || B < 1 || Tmax < 1 || Tmax > 1024)
        return fail(ctx, IFD_ERR_ARG, "ifd_encode_points: bad argument (1 <= Tmax <= 1024)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = launch_encode_points(ctx->d_w, ctx->eo, ctx->d_enc_img, sel, t_per_cloud, B, Tmax, planes_pre, c_points, ctx->dc, s);
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_encode_points", e);
}

int ifd_unet(ifd_ctx* ctx, const float* planes_pre, int B, float* planes, void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (ctx->model != IFD_MODEL_CONVONET) return fail(ctx, IFD_ERR_ARG, "ifd_unet: not a ConvONet context");
    if (!planes_pre || !planes || B < 1) return fail(ctx, IFD_ERR_ARG, "ifd_unet: bad argument");
    hipError_t e = ensure_buf(&ctx->ws_enc, &ctx->ws_enc_bytes, unet_workspace_floats(3 * B) * sizeof(float));
    if (e != hipSuccess) return fail(ctx, IFD_ERR_NOMEM, "ifd_unet workspace", e);
    e = launch_unet(ctx->uw, planes_pre, planes, static_cast<float*>(ctx->ws_enc), 3 * B, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_unet launch", e);
}

int ifd_encode_planes(ifd_ctx* ctx, const float* sel, const int32_t* t_per_cloud, int B, int Tmax, float* planes,
                      void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (ctx->model != IFD_MODEL_CONVONET) return fail(ctx, IFD_ERR_ARG, "ifd_encode_planes: not a ConvONet context");
    if (!sel || !planes || B < 1 || Tmax < 1 || Tmax > 1024)
        return fail(ctx, IFD_ERR_ARG, "ifd_encode_planes: bad argument (1 <= Tmax <= 1024)");
    // scratch = [pre-U-Net planes | U-Net activations]
    const size_t pre_floats = (size_t)B * CLOUD_PLANE_FLOATS;
    hipError_t e = ensure_buf(&ctx->ws_enc, &ctx->ws_enc_bytes, (pre_floats + unet_workspace_floats(3 * B)) * sizeof(float));
    if (e != hipSuccess) return fail(ctx, IFD_ERR_NOMEM, "ifd_encode_planes workspace", e);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* pre = static_cast<float*>(ctx->ws_enc);
    e = launch_encode_points(ctx->d_w, ctx->eo, ctx->d_enc_img, sel, t_per_cloud, B, Tmax, pre, nullptr, ctx->dc, s);
    if (e == hipSuccess) e = launch_unet(ctx->uw, pre, planes, pre + pre_floats, 3 * B, s);
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_encode_planes", e);
}

namespace {

// Clouds of more than 1024 points take two launches per Adam step (occupancy gradient; then neighbour lists + repulsion + Adam), and a
// launch ends with its slowest cloud: the list-step launch of 256 clouds on 256 CUs takes 187 us where ONE cloud's averages 79
// (profiles/r06_large_k_launch_times.txt) - most CUs idle while the slowest clouds finish.  The batch therefore goes as G contiguous
// groups on G streams (the caller's + side streams of the context), the launches enqueued step by step across the groups.  The steps
// of a group depend on its own clouds only, and the groups drift apart: the CUs one group's list step leaves idle run another group's
// occupancy workgroups.  Same kernels, same arithmetic per cloud: bit-identical to one group (tests).  Measured, K = 2048, 501 steps,
// time per point over the persistent kernel's (profiles/r06_large_k_groups.txt): 256 clouds 1.51 -> 1.34 (G = 4), 512 1.41 -> 1.24
// (G = 2), 1024 1.24 (2) / 1.20 (4), a file of 2304 1.30 -> 1.23 (4); groups of about one cloud per CU do best, more than four
// never helped, and chaining the occupancy launches in a ring through events cost more than it ordered (1.77).
int large_groups_of(const ifd_ctx* ctx, int B, bool lists) {
    if (ctx->test_large_groups > 0) return std::min(std::min(ctx->test_large_groups, LARGE_MAX_GROUPS), B);
    if (B < 32 || !lists) return 1;          // (the brute-force step kernels cost every cloud the same: nothing to fill)
    if (B <= ctx->n_cu) return LARGE_MAX_GROUPS;
    return std::max(2, std::min(LARGE_MAX_GROUPS, (B + ctx->n_cu / 2) / ctx->n_cu));
}

hipError_t large_optimize_in_groups(ifd_ctx* ctx, const float* dec_img, const float* planes, float* p, float* m, float* v, float* loss,
                                    const int32_t* lbpc, int B, int K, const OptArgs& a, hipStream_t s) {
    const bool own = m == nullptr;
    const int G = large_groups_of(ctx, B, K <= LARGE_LDS_MAXK && a.knn_scan_every_step == 0);
    const int per = (B + G - 1) / G;
    const size_t slice = (large_ws_bytes(per, K, own) + 255) & ~(size_t)255;
    hipError_t e = ensure_ws(ctx, slice * G);
    if (e != hipSuccess) return e;
    if (G > 1) {
        if (!ctx->large_fork && (e = hipEventCreateWithFlags(&ctx->large_fork, hipEventDisableTiming)) != hipSuccess) return e;
        for (int g = 1; g < G; ++g) {
            if (!ctx->large_side[g - 1] && (e = hipStreamCreateWithFlags(&ctx->large_side[g - 1], hipStreamNonBlocking)) != hipSuccess) return e;
            if (!ctx->large_join[g - 1] && (e = hipEventCreateWithFlags(&ctx->large_join[g - 1], hipEventDisableTiming)) != hipSuccess) return e;
        }
        if ((e = hipEventRecord(ctx->large_fork, s)) != hipSuccess) return e;
    }
    struct Group { int g0, Bg, parts; char* G; float *p, *m, *v, *loss; const int32_t* lb; const float* planes; void *f_ws, *list_ws; hipStream_t s; };
    Group gr[LARGE_MAX_GROUPS];
    int n_groups = 0;
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t x) { if (x != hipSuccess && first == hipSuccess) first = x; };
    const int ntiles = (K + 31) / 32, n_cu = std::max(8, ctx->n_cu);
    for (int g = 0; g < G; ++g) {
        Group& q = gr[g];
        q.g0 = g * per; q.Bg = std::min(per, B - q.g0);
        if (q.Bg <= 0) break;
        ++n_groups;
        const size_t o3 = (size_t)q.g0 * K * 3;
        q.s = g == 0 ? s : ctx->large_side[g - 1];
        q.G = static_cast<char*>(ctx->ws) + slice * g;
        q.p = p + o3; q.planes = planes + (size_t)q.g0 * CLOUD_PLANE_FLOATS;
        q.loss = loss ? loss + 2 * (size_t)q.g0 : nullptr; q.lb = lbpc ? lbpc + q.g0 : nullptr;
        // one workgroup fills a CU; with fewer clouds than CUs a cloud's tiles are shared out over `parts` workgroups
        q.parts = q.Bg >= n_cu ? 1 : std::min((ntiles + 7) / 8, std::max(1, n_cu / q.Bg));
        if (g > 0) note(hipStreamWaitEvent(q.s, ctx->large_fork, 0));
        note(large_f_prepare(q.G, q.Bg, K, own, &q.f_ws, q.s));
        q.list_ws = large_list_ws(q.G, q.Bg, K, own);
        if (own) {                                        // own moments behind G (large_ws_bytes), zeroed
            q.m = reinterpret_cast<float*>(q.G + (size_t)q.Bg * K * 16);
            q.v = q.m + (size_t)q.Bg * K * 3;
            note(hipMemsetAsync(q.m, 0, (size_t)q.Bg * K * 3 * 4 * 2, q.s));
        } else {
            q.m = m + o3; q.v = v + o3;
        }
    }
    for (int step = 0; step < a.steps && first == hipSuccess; ++step) {
        const bool last = step == a.steps - 1;
        for (int g = 0; g < n_groups; ++g) {
            Group& q = gr[g];
            note(launch_large_occupancy(a.precision, dec_img, q.planes, q.p, q.Bg, q.parts, K, q.lb, a.loss_batch, a.threshold,
                                        (last && q.loss != nullptr) ? 1 : 0, q.G, a.dc, q.s));
            note(launch_large_step(q.p, q.m, q.v, q.G, q.Bg, K, static_cast<const float*>(ctx->adam_tab), step, q.lb, a, last ? q.loss : nullptr,
                                   q.f_ws, q.list_ws, ctx->d_counters, q.s));
        }
    }
    for (int g = 0; g < n_groups; ++g) {
        Group& q = gr[g];
        if (a.normalize && first == hipSuccess) note(launch_large_normalize(q.p, q.Bg, K, q.s));
        if (g > 0) {                                      // (joined even after a failed launch: the caller's stream must not run ahead of a side stream)
            hipError_t x = hipEventRecord(ctx->large_join[g - 1], q.s);
            if (x == hipSuccess) x = hipStreamWaitEvent(s, ctx->large_join[g - 1], 0);
            note(x);
        }
    }
    return first;
}

}  // namespace

int ifd_decode_ex(ifd_ctx* ctx, const float* planes, const float* p, int B, int K, int precision, float* logits, float* dlogit_dp,
                  void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (ctx->model != IFD_MODEL_CONVONET) return fail(ctx, IFD_ERR_ARG, "ifd_decode: not a ConvONet context (use ifd_onet_decode)");
    if (!planes || !p || !logits || B < 1 || K < 1) return fail(ctx, IFD_ERR_ARG, "ifd_decode: bad argument");
    if (precision < 0 || precision > 2) return fail(ctx, IFD_ERR_ARG, "ifd_decode_ex: precision must be 0 (f32 MFMA), 1 (bf16x6) or 2 (bf16x3)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = precision == 0 ? launch_decode(ctx->d_dec_img, planes, p, B, K, logits, dlogit_dp, ctx->dc, s)
                                  : launch_decode_bf(precision, ctx->d_dec_img_bf, planes, p, B, K, logits, dlogit_dp, ctx->dc, ctx->n_cu, s);
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_decode launch", e);
}

int ifd_decode(ifd_ctx* ctx, const float* planes, const float* p, int B, int K, float* logits, float* dlogit_dp,
               void* stream) {
    return ifd_decode_ex(ctx, planes, p, B, K, 0, logits, dlogit_dp, stream);
}

int ifd_repulsion(ifd_ctx* ctx, const float* p, int B, int K, float* loss, float* grad, int32_t* knn_idx,
                  void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (!p || !loss || bad_bk(B, K)) return fail(ctx, IFD_ERR_ARG, "ifd_repulsion: bad argument (6 <= K <= 10000)");
    hipError_t e = hipSuccess;
    if (K <= MAXK) {
        e = launch_repulsion(p, B, K, loss, grad, knn_idx, 0.07f, 0.03f, 1e-12f, static_cast<hipStream_t>(stream));
    } else {
        e = ensure_ws(ctx, large_f_bytes(B, K));          // (clouds beyond 4096 points: accumulators in the context's workspace)
        if (e == hipSuccess)
            e = launch_large_repulsion(p, B, K, loss, grad, knn_idx, 0.07f, 0.03f, 1e-12f, large_f_bytes(B, K) ? ctx->ws : nullptr,
                                       static_cast<hipStream_t>(stream));
    }
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_repulsion launch", e);
}

int ifd_optimize(ifd_ctx* ctx, const float* planes, float* p, int B, int K, const ifd_opt_params* prm,
                 const int32_t* loss_batch_per_cloud, float* m, float* v, float* loss, void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (ctx->model != IFD_MODEL_CONVONET) return fail(ctx, IFD_ERR_ARG, "ifd_optimize: not a ConvONet context (use ifd_onet_optimize)");
    if (prm && prm->struct_size != (int32_t)sizeof(ifd_opt_params))
        return fail(ctx, IFD_ERR_ARG, "ifd_optimize: ifd_opt_params.struct_size does not match this library (caller built against another ifd.h?)");
    if (!planes || !p || !prm || bad_bk(B, K))
        return fail(ctx, IFD_ERR_ARG, "ifd_optimize: bad argument (6 <= K <= 10000)");
    if ((m == nullptr) != (v == nullptr)) return fail(ctx, IFD_ERR_ARG, "ifd_optimize: pass both m and v or neither");
    if (prm->steps < 0 || prm->t0 < 0 || prm->loss_batch < 1 || (prm->t0 > 0 && !m))
        return fail(ctx, IFD_ERR_ARG, "ifd_optimize: bad steps/t0/loss_batch (t0 > 0 needs m and v)");
    OptArgs a{};
    a.steps = prm->steps; a.t0 = prm->t0; a.loss_batch = prm->loss_batch; a.normalize = prm->normalize;
    a.knn_scan_every_step = prm->knn_reference_form ? 2 : (prm->knn_scan_every_step ? 1 : 0);
    a.planes_shared = prm->planes_shared;
    a.lr = prm->lr; a.rep_weight = prm->rep_weight; a.threshold = prm->threshold;
    a.rep_radius = prm->rep_radius; a.rep_h = prm->rep_h; a.rep_eps = prm->rep_eps;
    a.dc = ctx->dc;
    a.coop_timeout_ticks = ctx->coop_timeout_ticks;
    a.test_drop_member = ctx->test_drop_member;
    if (prm->split != 0 && prm->split != 1 && prm->split != 2 && prm->split != 4)
        return fail(ctx, IFD_ERR_ARG, "ifd_optimize: split must be 0 (automatic), 1, 2 or 4");
    if (prm->precision < 0 || prm->precision > 2)
        return fail(ctx, IFD_ERR_ARG, "ifd_optimize: precision must be 0 (f32 MFMA), 1 (bf16x6) or 2 (bf16x3)");
    a.precision = prm->precision;
    const bool large = K > MAXK;              // more points than one CU's LDS holds: two launches per step (optimize.hip)
    hipError_t e = large ? hipSuccess : ensure_ws(ctx, optimize_ws_bytes(B));          // (large: per stream group, run_large_in_groups)
    if (e == hipSuccess) e = ensure_buf(&ctx->adam_tab, &ctx->adam_bytes, (size_t)(prm->steps > 0 ? prm->steps : 1) * 2 * sizeof(float));
    if (e != hipSuccess) return fail(ctx, IFD_ERR_NOMEM, "ifd_optimize workspace", e);
    e = hipMemsetAsync(ctx->d_counters, 0, N_COUNTERS_DEV * sizeof(unsigned long long), static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_counters + STATUS_TIMEOUT_CUR, 0, sizeof(unsigned long long), static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = launch_adam_table(static_cast<float*>(ctx->adam_tab), a.t0, a.steps, a.lr, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(ctx, IFD_ERR_HIP, "ifd_optimize memset / Adam table", e);
    if (large) {
        e = large_optimize_in_groups(ctx, a.precision != 0 ? ctx->d_dec_img_bf : ctx->d_dec_img_opt, planes, p, m, v, loss, loss_batch_per_cloud,
                                     B, K, a, static_cast<hipStream_t>(stream));
        return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_optimize launch (large clouds)", e);
    }
    e = launch_optimize(a.precision != 0 ? ctx->d_dec_img_bf : ctx->d_dec_img_opt, planes, p, m, v, loss, loss_batch_per_cloud, ctx->ws, ctx->d_counters,
                        static_cast<const float*>(ctx->adam_tab), B, K, a, prm->split, ctx->n_cu, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_optimize launch", e);
}

int ifd_get_counters(ifd_ctx* ctx, uint64_t* out_host, int n) {
    if (!ctx || !out_host || n < 1) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    unsigned long long tmp[N_COUNTERS_ALLOC] = {0};
    hipError_t e = hipMemcpy(tmp, ctx->d_counters, sizeof(tmp), hipMemcpyDeviceToHost);   // synchronises
    if (e != hipSuccess) return fail(ctx, IFD_ERR_HIP, "ifd_get_counters", e);
    if (ctx->model == IFD_MODEL_ONET) {   // host-side tallies of the last ifd_onet_mesh_sample
        tmp[8] = ctx->mesh_points;
        tmp[9] = ctx->mesh_rounds;
    }
    // slots >= IFD_N_COUNTERS: wave trace (diagnostic -DIFD_TRACE builds), the two status words (read-only here), tile trace
    for (int i = 0; i < n; ++i) out_host[i] = i < N_COUNTERS_ALLOC ? tmp[i] : 0;
    return IFD_OK;
}

int ifd_optimize_status(ifd_ctx* ctx, void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // read-and-reset in one atomic exchange per word: an event raised by a launch on another stream between a copy and a
    // separate memset would have been cleared unreported (round-4 advisor)
    unsigned long long st[2] = {0, 0};
    hipLaunchKernelGGL(status_take_kernel, dim3(1), dim3(64), 0, s, ctx->d_counters + STATUS_OVERFLOW, ctx->d_status_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(st, ctx->d_status_out, sizeof(st), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(ctx, IFD_ERR_HIP, "ifd_optimize_status", e);
    if (st[0] == 0 && st[1] == 0) return IFD_OK;
    char msg[256];
    if (st[1] != 0) {
        std::snprintf(msg, sizeof(msg), "split clouds: %llu cross-CU wait(s) gave up (a member workgroup never arrived: CUs masked or "
                      "held by another process?); the results of that launch are invalid - rerun with ifd_opt_params.split = 1", st[1]);
        return fail(ctx, IFD_ERR_TIMEOUT, msg);
    }
    std::snprintf(msg, sizeof(msg), "repulsion gradient: the fixed-point sums of %llu point-step(s) reached half their range (|sum| >= 128; "
                  "non-reference rep_radius / rep_h?); the results of that launch are invalid", st[0]);
    return fail(ctx, IFD_ERR_OVERFLOW, msg);
}

int ifd_normalize_unit_sphere(ifd_ctx* ctx, float* p, int B, int K, void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (!p || B < 1 || K < 1 || K > LARGE_MAXK) return fail(ctx, IFD_ERR_ARG, "ifd_normalize_unit_sphere: bad argument (1 <= K <= 10000)");
    hipError_t e = K <= MAXK ? launch_normalize(p, B, K, static_cast<hipStream_t>(stream))
                             : launch_large_normalize(p, B, K, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_normalize launch", e);
}

// ---------------------------------------------------------------------------------------------
// ONet-Opt variant
// ---------------------------------------------------------------------------------------------
size_t ifd_onet_weight_count(void) { return omap().host_total; }

ifd_ctx* ifd_onet_create(const float* weights_host, size_t n_weights, int device) {
    g_create_error.clear();
    const OnetMap& m = omap();
    if (!weights_host) { g_create_error = "ifd_onet_create: NULL argument"; return nullptr; }
    if (n_weights != m.host_total) {
        g_create_error = "ifd_onet_create: expected " + std::to_string(m.host_total) + " weights, got " + std::to_string(n_weights);
        return nullptr;
    }
    ifd_ctx* ctx = new (std::nothrow) ifd_ctx();
    if (!ctx) { g_create_error = "ifd_onet_create: out of host memory"; return nullptr; }
    ctx->device = device;
    {
        int n_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0) ctx->n_cu = n_cu;
    }
    ctx->model = IFD_MODEL_ONET;
    ctx->w.assign(weights_host, weights_host + n_weights);
    const float* w = ctx->w.data();
    // padded device copy
    std::vector<float> dev(m.dev_total, 0.f);
    auto put = [&](const OnetTensor& t) { std::memcpy(dev.data() + t.dev_off, w + t.host_off, t.n * sizeof(float)); return (int)t.dev_off; };
    put(m.fc_p_w); put(m.fc_p_b);
    OnetDecOffsets& od = ctx->od;
    for (int i = 0; i < 11; ++i) {
        od.cbn_gamma_w[i] = put(m.cbn[i][0]); od.cbn_gamma_b[i] = put(m.cbn[i][1]);
        od.cbn_beta_w[i] = put(m.cbn[i][2]); od.cbn_beta_b[i] = put(m.cbn[i][3]);
        od.cbn_mean[i] = put(m.cbn[i][4]); od.cbn_var[i] = put(m.cbn[i][5]);
    }
    for (int i = 0; i < 5; ++i) { put(m.fc0_w[i]); od.fc0_b[i] = put(m.fc0_b[i]); put(m.fc1_w[i]); put(m.fc1_b[i]); }
    put(m.out_w); put(m.out_b);
    OnetEncOffsets& oe = ctx->oe;
    oe.pos_w = put(m.pos_w); oe.pos_b = put(m.pos_b);
    for (int i = 0; i < 5; ++i) {
        oe.fc0_w[i] = put(m.e_fc0_w[i]); oe.fc0_b[i] = put(m.e_fc0_b[i]);
        oe.fc1_w[i] = put(m.e_fc1_w[i]); oe.fc1_b[i] = put(m.e_fc1_b[i]);
        oe.sc_w[i] = put(m.e_sc_w[i]);
    }
    oe.fcc_w = put(m.fcc_w); oe.fcc_b = put(m.fcc_b);
    // decoder layer images: forward fc_0 / fc_1 of blocks 0..4, then the transposes in backward order
    const size_t L = (size_t)ONET_H * ONET_H;
    std::vector<float> img(20 * L);
    for (int i = 0; i < 5; ++i) {
        onet_fragment_image(w + m.fc0_w[i].host_off, false, img.data() + (size_t)(2 * i) * L);
        onet_fragment_image(w + m.fc1_w[i].host_off, false, img.data() + (size_t)(2 * i + 1) * L);
        onet_fragment_image(w + m.fc1_w[i].host_off, true, img.data() + (size_t)(10 + 2 * (4 - i)) * L);
        onet_fragment_image(w + m.fc0_w[i].host_off, true, img.data() + (size_t)(11 + 2 * (4 - i)) * L);
    }
    // the bf16 piece images of the same twenty layers (W0^T output-half-major), packed into a float vector for the upload helper
    const size_t LB = (size_t)16 * (8 * 3 * 64 * 8);              // uint16 per layer image
    std::vector<float> img_bf(20 * LB / 2);
    {
        uint16_t* ib = reinterpret_cast<uint16_t*>(img_bf.data());
        for (int i = 0; i < 5; ++i) {
            onet_fragment_image_bf(w + m.fc0_w[i].host_off, false, false, ib + (size_t)(2 * i) * LB);
            onet_fragment_image_bf(w + m.fc1_w[i].host_off, false, false, ib + (size_t)(2 * i + 1) * LB);
            onet_fragment_image_bf(w + m.fc1_w[i].host_off, true, false, ib + (size_t)(10 + 2 * (4 - i)) * LB);
            onet_fragment_image_bf(w + m.fc0_w[i].host_off, true, true, ib + (size_t)(11 + 2 * (4 - i)) * LB);
        }
    }
    // small parameters, in the LDS order of onet.hip: fc_p [256][4] | fc_1 biases [5][256] | fc_out w [256] | b
    std::vector<float> small((size_t)onet_small_floats(), 0.f);
    for (int c = 0; c < ONET_H; ++c) {
        for (int a = 0; a < 3; ++a) small[c * 4 + a] = w[m.fc_p_w.host_off + c * 3 + a];
        small[c * 4 + 3] = w[m.fc_p_b.host_off + c];
    }
    for (int i = 0; i < 5; ++i)
        for (int c = 0; c < ONET_H; ++c) small[ONET_H * 4 + i * ONET_H + c] = w[m.fc1_b[i].host_off + c];
    for (int c = 0; c < ONET_H; ++c) small[ONET_H * 4 + 5 * ONET_H + c] = w[m.out_w.host_off + c];
    small[ONET_H * 4 + 5 * ONET_H + ONET_H] = w[m.out_b.host_off];

    hipError_t e = hipSetDevice(device);
    auto upload = [&](float** dst, const std::vector<float>& src) {
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(dst), src.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice);
    };
    upload(&ctx->d_w, dev);
    upload(&ctx->d_onet_img, img);
    upload(&ctx->d_onet_img_bf, img_bf);
    upload(&ctx->d_onet_small, small);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_counters), N_COUNTERS_ALLOC * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(ctx->d_counters, 0, N_COUNTERS_ALLOC * sizeof(unsigned long long));
    if (e == hipSuccess && !ctx->d_status_out) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_status_out), 2 * sizeof(unsigned long long));
    { const OptEnv oe = read_opt_env(); ctx->coop_timeout_ticks = oe.coop_timeout_ticks; ctx->test_drop_member = oe.test_drop_member; ctx->test_no_morton = oe.test_no_morton; }
    if (e == hipSuccess) e = configure_prep_kernels();
    if (e == hipSuccess) e = configure_optimize_kernels();
    if (e == hipSuccess) e = configure_onet_kernels();
    if (e == hipSuccess) e = configure_onet_bf_kernels();
    if (e == hipSuccess) e = mc_upload_table();
    if (e != hipSuccess) {
        g_create_error = std::string("ifd_onet_create: ") + hipGetErrorString(e);
        ifd_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

int ifd_onet_encode(ifd_ctx* ctx, const float* sel, const int32_t* t_per_cloud, int B, int Tmax, float* c, void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (ctx->model != IFD_MODEL_ONET) return fail(ctx, IFD_ERR_ARG, "ifd_onet_encode: not an ONet context");
    if (!sel || !c || B < 1 || Tmax < 1 || Tmax > 1024) return fail(ctx, IFD_ERR_ARG, "ifd_onet_encode: bad argument (1 <= Tmax <= 1024)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int chunk = 256;                                     // clouds per pass: ~0.8 GB of activations at T = 300
    hipError_t e = ensure_buf(&ctx->ws_enc, &ctx->ws_enc_bytes, onet_encode_ws_floats(B < chunk ? B : chunk, Tmax) * sizeof(float));
    if (e != hipSuccess) return fail(ctx, IFD_ERR_NOMEM, "ifd_onet_encode workspace", e);
    for (int b0 = 0; b0 < B && e == hipSuccess; b0 += chunk) {
        const int nb = B - b0 < chunk ? B - b0 : chunk;
        e = launch_onet_encode(ctx->d_w, ctx->oe, sel + (size_t)b0 * Tmax * 3, t_per_cloud ? t_per_cloud + b0 : nullptr, nb, Tmax,
                               static_cast<float*>(ctx->ws_enc), c + (size_t)b0 * ONET_C, s);
    }
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_onet_encode", e);
}

namespace {
// CBN fold of B clouds into the context's fold buffer: returns the device pointer of ab [B][11][2][256].  The buffer belongs to the
// decoder-side calls (ifd_onet_decode / ifd_onet_optimize / ifd_onet_mesh_sample, stream-ordered with each other like every call of
// one kind: include/ifd.h); the encoder calls never touch it, so ifd_onet_encode of the NEXT pass may run on a second stream while
// an optimiser launch still reads its coefficients (round-5 advisor: it used to live at offset 0 of the encoder scratch).
hipError_t onet_fold(ifd_ctx* ctx, const float* c, int B, hipStream_t s, float** ab_out) {
    const size_t per = (size_t)2 * ONET_NCBN * ONET_H;
    hipError_t e = ensure_buf(&ctx->ws_fold, &ctx->ws_fold_bytes, 2 * per * B * sizeof(float));
    if (e != hipSuccess) return e;
    float* gb = static_cast<float*>(ctx->ws_fold);
    float* ab = gb + per * B;
    *ab_out = ab;
    return launch_onet_cbn(ctx->d_w, ctx->od, c, B, gb, ab, s);
}
}  // namespace

int ifd_onet_decode_ex(ifd_ctx* ctx, const float* c, const float* p, int B, int K, int precision, float* logits, float* dlogit_dp,
                       void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (ctx->model != IFD_MODEL_ONET) return fail(ctx, IFD_ERR_ARG, "ifd_onet_decode: not an ONet context");
    if (!c || !p || !logits || B < 1 || K < 1) return fail(ctx, IFD_ERR_ARG, "ifd_onet_decode: bad argument");
    if (precision < 0 || precision > 2) return fail(ctx, IFD_ERR_ARG, "ifd_onet_decode_ex: precision must be 0 (f32 MFMA), 1 (bf16x6) or 2 (bf16x3)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* ab = nullptr;
    hipError_t e = onet_fold(ctx, c, B, s, &ab);
    if (e == hipSuccess)
        e = precision == 0 ? launch_onet_decode(ctx->d_onet_img, ctx->d_onet_small, ab, p, B, K, logits, dlogit_dp, s)
                           : launch_onet_decode_bf(precision, ctx->d_onet_img_bf, ctx->d_onet_small, ab, p, B, K, logits, dlogit_dp, s);
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_onet_decode", e);
}

int ifd_onet_decode(ifd_ctx* ctx, const float* c, const float* p, int B, int K, float* logits, float* dlogit_dp, void* stream) {
    return ifd_onet_decode_ex(ctx, c, p, B, K, 0, logits, dlogit_dp, stream);
}

int ifd_onet_optimize(ifd_ctx* ctx, const float* c, float* p, int B, int K, const ifd_opt_params* prm,
                      const int32_t* loss_batch_per_cloud, float* m, float* v, float* loss, void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (ctx->model != IFD_MODEL_ONET) return fail(ctx, IFD_ERR_ARG, "ifd_onet_optimize: not an ONet context");
    if (prm && prm->struct_size != (int32_t)sizeof(ifd_opt_params))
        return fail(ctx, IFD_ERR_ARG, "ifd_onet_optimize: ifd_opt_params.struct_size does not match this library (caller built against another ifd.h?)");
    if (!c || !p || !prm || bad_bk(B, K))
        return fail(ctx, IFD_ERR_ARG, "ifd_onet_optimize: bad argument (6 <= K <= 10000)");
    if ((m == nullptr) != (v == nullptr)) return fail(ctx, IFD_ERR_ARG, "ifd_onet_optimize: pass both m and v or neither");
    if (prm->steps < 0 || prm->t0 < 0 || prm->loss_batch < 1 || (prm->t0 > 0 && !m))
        return fail(ctx, IFD_ERR_ARG, "ifd_onet_optimize: bad steps/t0/loss_batch (t0 > 0 needs m and v)");
    if (prm->precision < 0 || prm->precision > 2)
        return fail(ctx, IFD_ERR_ARG, "ifd_onet_optimize: precision must be 0 (f32 MFMA), 1 (bf16x6) or 2 (bf16x3)");
    OptArgs a{};
    a.steps = prm->steps; a.t0 = prm->t0; a.loss_batch = prm->loss_batch; a.normalize = prm->normalize;
    a.knn_scan_every_step = prm->knn_reference_form ? 2 : (prm->knn_scan_every_step ? 1 : 0);
    a.lr = prm->lr; a.rep_weight = prm->rep_weight; a.threshold = prm->threshold;
    a.rep_radius = prm->rep_radius; a.rep_h = prm->rep_h; a.rep_eps = prm->rep_eps;
    a.coop_timeout_ticks = ctx->coop_timeout_ticks;
    a.test_drop_member = ctx->test_drop_member;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool large = K > MAXK;              // more points than one CU's LDS holds: two launches per step (onet.hip)
    hipError_t e = ensure_ws(ctx, large ? large_ws_bytes(B, K, m == nullptr) : knn_list_bytes(B));
    if (e == hipSuccess) e = ensure_buf(&ctx->adam_tab, &ctx->adam_bytes, (size_t)(prm->steps > 0 ? prm->steps : 1) * 2 * sizeof(float));
    if (e != hipSuccess) return fail(ctx, IFD_ERR_NOMEM, "ifd_onet_optimize workspace", e);
    float* ab = nullptr;
    e = onet_fold(ctx, c, B, s, &ab);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_counters, 0, N_COUNTERS_DEV * sizeof(unsigned long long), s);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_counters + STATUS_TIMEOUT_CUR, 0, sizeof(unsigned long long), s);
    if (e == hipSuccess) e = launch_adam_table(static_cast<float*>(ctx->adam_tab), a.t0, a.steps, a.lr, s);
    a.precision = prm->precision;
    if (e == hipSuccess && large)
        e = launch_onet_large_optimize(prm->precision != 0 ? ctx->d_onet_img_bf : ctx->d_onet_img, ctx->d_onet_small, ab, p, m, v, loss,
                                       loss_batch_per_cloud, ctx->ws, ctx->d_counters, static_cast<const float*>(ctx->adam_tab), B, K, a, s);
    else if (e == hipSuccess && prm->precision != 0)
        e = launch_onet_optimize_bf(prm->precision, ctx->d_onet_img_bf, ctx->d_onet_small, ab, p, m, v, loss, loss_batch_per_cloud,
                                    static_cast<uint16_t*>(ctx->ws), ctx->d_counters, static_cast<const float*>(ctx->adam_tab), B, K,
                                    a, s);
    else if (e == hipSuccess)
        e = launch_onet_optimize(ctx->d_onet_img, ctx->d_onet_small, ab, p, m, v, loss, loss_batch_per_cloud,
                                 static_cast<uint16_t*>(ctx->ws), ctx->d_counters, static_cast<const float*>(ctx->adam_tab), B, K,
                                 a, s);
    return e == hipSuccess ? IFD_OK : fail(ctx, IFD_ERR_HIP, "ifd_onet_optimize", e);
}

int ifd_mc_table(int8_t* tri, uint8_t* ntri) {
    if (!tri || !ntri) return IFD_ERR_ARG;
    mc_host_table(reinterpret_cast<int8_t(*)[16]>(tri), ntri);
    return IFD_OK;
}

int ifd_onet_mesh_sample(ifd_ctx* ctx, const float* c, int B, const ifd_mesh_params* prm, float* points,
                         int32_t* n_triangles, float* grid, float* triangles, void* stream) {
    if (!ctx) return IFD_ERR_ARG;
    IFD_ON_CTX_DEVICE(ctx);
    if (ctx->model != IFD_MODEL_ONET) return fail(ctx, IFD_ERR_ARG, "ifd_onet_mesh_sample: not an ONet context");
    if (!c || !prm || prm->struct_size != (int32_t)sizeof(ifd_mesh_params) || !points || !n_triangles || B < 1)
        return fail(ctx, IFD_ERR_ARG, "ifd_onet_mesh_sample: bad argument");
    const int depth = prm->upsampling_steps, res0 = prm->resolution0;
    if (depth < 0 || depth > 2 || res0 < 2 || (res0 << depth) > 128 || prm->n_sample < 1 || prm->max_triangles < 1 ||
        !(prm->threshold > 0.0 && prm->threshold < 1.0) || prm->precision < 0 || prm->precision > 2)
        return fail(ctx, IFD_ERR_ARG, "ifd_onet_mesh_sample: resolution0 << upsampling_steps <= 128, steps <= 2, 0 < threshold < 1, precision 0 ... 2");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MiseGrid g{};
    g.res0 = res0; g.depth = depth; g.P = (res0 << depth) + 1; g.P3 = g.P * g.P * g.P;
    g.cap = g.P3;                                   // every grid point can be queued once
    g.pend_stride = ((size_t)g.P3 + 3) & ~(size_t)3;
    g.sub_total = 0;
    for (int l = 0; l < 4; ++l) {
        g.sub_off[l] = g.sub_total;
        if (l < depth) { const int nv = res0 << l; g.sub_total += nv * nv * nv; }
    }
    g.sub_total = (g.sub_total + 3) & ~3;
    if (g.sub_total == 0) g.sub_total = 4;
    g.threshold = std::log(prm->threshold) - std::log(1.0 - prm->threshold);        // generation.py:97
    const float box = 1.0f + prm->padding;
    const int NC = g.P + 1, ncube = NC * NC * NC, capT = prm->max_triangles;
    // per-cloud scratch layout (bytes, every block 16-byte aligned)
    auto al = [](size_t n) { return (n + 15) & ~(size_t)15; };
    const size_t o_val = 0, o_known = o_val + al((size_t)g.P3 * 4), o_pend = o_known + al(g.P3), o_sub = o_pend + al(g.pend_stride),
                 o_mix = o_sub + al(g.sub_total), o_list = o_mix + al(g.sub_total), o_cube = o_list + al((size_t)g.cap * 4),
                 o_tris = o_cube + al((size_t)ncube * 4), o_area = o_tris + al((size_t)capT * 36), per = o_area + al((size_t)capT * 8);
    const size_t fit = ((size_t)6 << 30) / per;                 // ~6 GB of scratch per pass
    const int chunk = fit < 1 ? 1 : fit > (size_t)B ? B : (int)fit;
    // layout: arrays are [chunk][stride] each (struct-of-arrays), plus count / ntri / ab folded separately
    const size_t total = per * chunk + al((size_t)chunk * 4) * 3 + al((size_t)(chunk + 1) * 4);
    hipError_t e = ensure_buf(&ctx->ws_mesh, &ctx->ws_mesh_bytes, total);
    if (e != hipSuccess) return fail(ctx, IFD_ERR_NOMEM, "ifd_onet_mesh_sample workspace", e);
    if (ctx->h_mesh_counts_n < (size_t)chunk) {          // pinned landing slots of the queue lengths + their events (kept by the context)
        if (ctx->h_mesh_counts) (void)hipHostFree(ctx->h_mesh_counts);
        ctx->h_mesh_counts = nullptr;
        ctx->h_mesh_counts_n = 0;
        e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_mesh_counts), (size_t)2 * chunk * sizeof(int), hipHostMallocDefault);
        if (e != hipSuccess) return fail(ctx, IFD_ERR_NOMEM, "ifd_onet_mesh_sample pinned counts", e);
        ctx->h_mesh_counts_n = (size_t)chunk;
    }
    for (hipEvent_t& ev : ctx->mesh_ev)
        if (!ev && (e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess)
            return fail(ctx, IFD_ERR_HIP, "ifd_onet_mesh_sample events", e);
    char* base = static_cast<char*>(ctx->ws_mesh);
    auto blk = [&](size_t off_per_cloud) { return base + off_per_cloud * chunk; };
    g.val = reinterpret_cast<float*>(blk(o_val));
    g.known = reinterpret_cast<uint8_t*>(blk(o_known));
    g.pend = reinterpret_cast<uint8_t*>(blk(o_pend));
    g.sub = reinterpret_cast<uint8_t*>(blk(o_sub));
    g.mix = reinterpret_cast<uint8_t*>(blk(o_mix));
    g.list = reinterpret_cast<int*>(blk(o_list));
    int* cube_offs = reinterpret_cast<int*>(blk(o_cube));
    float* tris = reinterpret_cast<float*>(blk(o_tris));
    double* area = reinterpret_cast<double*>(blk(o_area));
    g.count = reinterpret_cast<int*>(base + per * chunk);
    int* ntri = reinterpret_cast<int*>(base + per * chunk + al((size_t)chunk * 4));
    g.prev = reinterpret_cast<int*>(base + per * chunk + al((size_t)chunk * 4) * 2);
    g.plan = reinterpret_cast<int*>(base + per * chunk + al((size_t)chunk * 4) * 3);
    ctx->mesh_points = 0;
    ctx->mesh_rounds = 0;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = B - b0 < chunk ? B - b0 : chunk;
        float* ab = nullptr;
        e = onet_fold(ctx, c + (size_t)b0 * ONET_C, nb, s, &ab);
        if (e == hipSuccess) e = launch_mise_init(g, nb, s);
        if (e != hipSuccess) return fail(ctx, IFD_ERR_HIP, "ifd_onet_mesh_sample init", e);
        // The MISE loop is driven from the device: queue lengths, the split of a round's decoder passes over the CUs and the
        // clouds that are finished are all decided there (onet.hip grid_plan_kernel / onet_grid_eval_kernel, mesh.hip
        // mise_begin_kernel).  The host only has to learn WHEN every queue has run empty, and it does so one round late: round r
        // is enqueued before the queue lengths round r - 1 produced have been looked at (pinned copy + event per round, two
        // slots), so the GPU never waits for the host; the one round enqueued past the end finds empty queues and is a handful of
        // empty launches.
        const int n0 = (res0 + 1) * (res0 + 1) * (res0 + 1);
        ctx->mesh_points += (unsigned long long)n0 * nb;
        bool done = false;
        int round = 0;
        for (; round < 64 && !done; ++round) {
            int* slot = ctx->h_mesh_counts + (size_t)(round & 1) * ctx->h_mesh_counts_n;
            e = prm->precision != 0 ? launch_onet_grid_eval_bf(prm->precision, ctx->d_onet_img_bf, ctx->d_onet_small, ab, g, nb, ctx->n_cu, box, s)
                                    : launch_onet_grid_eval(ctx->d_onet_img, ctx->d_onet_small, ab, g, nb, ctx->n_cu, box, s);
            if (e == hipSuccess) e = launch_mise_update(g, nb, s);
            if (e == hipSuccess) e = hipMemcpyAsync(slot, g.count, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipEventRecord(ctx->mesh_ev[round & 1], s);
            if (e == hipSuccess && round >= 1) {
                e = hipEventSynchronize(ctx->mesh_ev[(round - 1) & 1]);
                const int* c_prev = ctx->h_mesh_counts + (size_t)((round - 1) & 1) * ctx->h_mesh_counts_n;   // queued BY round - 1 = evaluated IN this round
                int max_count = 0;
                for (int b = 0; b < nb; ++b) { max_count = c_prev[b] > max_count ? c_prev[b] : max_count; ctx->mesh_points += c_prev[b]; }
                if (max_count > g.cap) return fail(ctx, IFD_ERR_HIP, "ifd_onet_mesh_sample: point queue overflow");
                if (max_count == 0) done = true;           // this round (already enqueued) had nothing to do: the grid is complete
                else ++ctx->mesh_rounds;
            } else if (e == hipSuccess) {
                ++ctx->mesh_rounds;                        // round 0 always has the coarse lattice to evaluate
            }
            if (e != hipSuccess) return fail(ctx, IFD_ERR_HIP, "ifd_onet_mesh_sample round", e);
        }
        // the last enqueued round's copy must have landed before its slot is reused by the next chunk
        e = hipEventSynchronize(ctx->mesh_ev[(round - 1) & 1]);
        if (e != hipSuccess) return fail(ctx, IFD_ERR_HIP, "ifd_onet_mesh_sample round", e);
        e = launch_mise_fill(g, nb, s);
        if (e == hipSuccess && grid)
            e = hipMemcpyAsync(grid + (size_t)b0 * g.P3, g.val, (size_t)nb * g.P3 * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = launch_marching_cubes(g.val, nb, g.P, g.threshold, box, cube_offs, ntri, capT, tris, area, s);
        if (e == hipSuccess)
            e = launch_sample_surface(tris, area, ntri, nb, capT, prm->n_sample, prm->seed, (int)(prm->cloud_index_base + b0),
                                      points + (size_t)b0 * prm->n_sample * 3, s);
        if (e == hipSuccess) e = hipMemcpyAsync(n_triangles + b0, ntri, (size_t)nb * sizeof(int), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess && triangles)
            e = hipMemcpyAsync(triangles + (size_t)b0 * capT * 9, tris, (size_t)nb * capT * 9 * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return fail(ctx, IFD_ERR_HIP, "ifd_onet_mesh_sample", e);
    }
    return IFD_OK;
}

}  // extern "C"
