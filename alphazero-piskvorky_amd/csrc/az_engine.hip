// az_engine.hip -- host side of the C-ABI (include/az_engine.h): device memory, weight packing into
// MFMA fragment order, the lock-step episode loop (root eval -> S x {net, expand/backup/select} -> move),
// record export.  gfx950 only; there is no CPU implementation of the path in this library.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define AZ_ENGINE_TU 1
#include "../../include/az_engine.h"
#include "az_launch.h"

#include "az_host.h"

static thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct PackedNet {
    bool loaded = false;
    DevBuf c1, c2, c3, hd, pf, vf, c1b, c2b, c3b, hdb, pfb, vfb, v2w, v2b;
    DevBuf rblk[6], rblkb[6];          // ResidualBlock variant: the six 64->64 convs (c1/c1b hold the stem)
    DevBuf c1x[2], c2x[2], c3x[2], hdx[2];     // the convs (c1x: conv1 / stem) and head convs split into 16-bit MFMA fragments (az_net_emul.h): [0] bf16x3, [1] f16x2
    DevBuf rblkx[2][6];                // ... and the six 64 -> 64 convs of the ResidualBlock variant
    bool f16_ok = true;                // every weight of the emulated layers is inside float16's range (AZ_TRUNK_F16X2)
    // The split fragments are packed and uploaded when a scheme is first used (az_set_trunk_mode, or az_load_weights* while
    // a scheme is selected), not on every weight load: the training loop reloads weights every episode and mostly stays on
    // the float32 trunk.  Until then the float32 weights of the emulated layers wait here.
    bool emul_ready[2] = {false, false};
    std::vector<float> hw_first, hw_conv[6], hw_pol, hw_val;
    NetWeights w{};
    ResWeights rw{};
};

// One lane = one HIP stream with its own game slots and search state.  The lanes of an engine share the weights, the
// episode's record/tape buffers and ONE queue of game ids (a device counter every lane's refill claims from), and are
// driven by one host thread each: a lane's latency-bound tree / FC kernels run underneath another lane's conv trunk.
struct Lane {
    int index = 0;
    hipStream_t stream = nullptr;
    DevState d{};                  // the games' view (B = slots of this lane)
    DevState dv{};                 // the net kernels' view (B = evaluation items = slots x leaves per batch)
    DevBuf board, s_game, s_ply, s_player, s_last, s_status, s_net, edges, rows_used, cnt, active_dev, carried;   // per slot
    DevBuf path, depth, leaf_kind, leaf, leaf_last, logits, vhid, pol_feat, dbg, scratch, it_status, it_net, leaf_sym;       // per evaluation item
    // one ply (k_begin, (S+1) x {trunk, fc, step}, k_move) captured once as a hipGraph and replayed every ply:
    // 3(S+1)+2 launches (6(S+1)+2 with the split trunk) become one submission.  Indexed [split trunk][arena].
    struct PlyGraph {
        hipGraphExec_t exec = nullptr;
        LaunchCtx key{};           // kernel arguments baked into the nodes; any change re-captures
    } graph[2][2];
    std::vector<hipEvent_t> ev;    // profiling events
    // progress of the open episode
    int active = 0;                // active slots after the last refill
    int cur = 0;                   // slots the next ply is launched over: all of them, or (after a compacting refill) the active ones
    int plies_played = 0;          // plies this lane has played in the open episode
    int64_t steps = 0, trunk_launches = 0, plies = 0;
    double trunk_ms = 0.0, nn_ms = 0.0, step_ms = 0.0;
    double tape_wait_s = 0.0;      // host time this lane's thread was blocked on a tape wave (production or its upload)
    int rc = AZ_OK;                // result of the last threaded call on this lane
    std::string err;
};

struct az_engine {
    az_config cfg{};
    int n = 0, nn = 0, RW = 0, R = 0, PATH = 0, num_cus = 256;
    std::string err;
    hipStream_t stream = nullptr;  // lane 0's stream: uploads and copies of the shared buffers
    const SizeOps *ops = nullptr;
    std::vector<Lane> lanes;
    // shared by the lanes: tables, the game-id queue, the episode's tapes and records
    DevBuf T_table, log_table, sqrt_table, noise_off, next_game;
    DevBuf noise, u, rec_planes, rec_last, rec_action, rec_mover, rec_pi, rec_visits, g_nply, g_result, src_index;
    int split_max = 64;            // use the tile-split (low-latency) trunk when at most this many slots of a lane are active (measured: 32 / 64 / 128 -> 102.0 / 102.3 / 101.0 games per second on the 1024-game episode; the 51-game arena 3.08 / 1.64 / 1.64 s)
    int episode_games = 0;
    int64_t tape_len = 0;          // doubles per game in the noise tape
    bool have_episode = false;
    int reuse = 0;
    int vl = 1;                    // leaves per game and evaluation batch (az_set_virtual_loss); 1 = the reference's sequential loop
    bool vl_kernel = false;        // the batched tree kernel is in use (vl > 1, or AZ_VL_FORCE=1 to run it with batches of one)
    bool persist_allowed = true;   // AZ_PERSIST=0: never use the persistent search kernel
    int leaf_symmetry = 0;         // az_set_leaf_symmetry: every evaluation shows the net a pseudo-random dihedral symmetry of the position
    int trunk_mode = AZ_TRUNK_F32; // az_set_trunk_mode: AZ_TRUNK_BF16X3 / AZ_TRUNK_F16X2 = the convs on the 16-bit MFMAs with split operands
    int persist_gp = 0;            // games per workgroup of the persistent search kernel for the open episode, 0 = lock-step pipeline
    DevBuf cache;                  // evaluation cache shared by the lanes (az_set_eval_cache)
    unsigned cache_mask = 0, cache_gen = 1;
    std::vector<int> h_nply, h_result;
    PackedNet net[2];
    az_counters last{};
    // running episode (az_selfplay_begin .. az_selfplay_end)
    struct Run {
        bool open = false;
        int num_games = 0, max_plies = 0;
        bool add_noise = true, arena = false, preset = false, profile = true;
        az_counters c{};
        int64_t reused_roots = 0;
    } run;
    bool profile = false;          // HIP events around every trunk / FC launch (az_set_profiling); lanes then play one after another
    bool use_graph = true;         // AZ_GRAPH=0: launch kernel by kernel
    bool compact = true;           // AZ_COMPACT=0: never move the active slots to the front between plies (k_refill)
    void *comm = nullptr;          // RCCL communicator of az_dist_init (ncclComm_t), one rank per engine
    int dist_rank = 0, dist_world = 1;
    DevBuf dist_buf, dist_pack;    // small device scratch of the collectives; this rank's packed records
    TapeProducer *tapes = nullptr; // running while a self-play episode with engine-generated tapes is open
    bool stream_tapes = true;      // AZ_TAPE_STREAM=0: generate every tape before the first ply
    int tape_threads = 4;          // AZ_TAPE_THREADS: host threads of the tape producer
    int host_cpus = 1;             // CPUs this process may run on (affinity mask and cgroup quota), what the thread counts are sized from
};

static void stop_tapes(az_engine *e)
{
    if (e->tapes) { e->tapes->shutdown(); delete e->tapes; e->tapes = nullptr; }
}

static LaunchCtx ctx_of_impl(const az_engine *e, const Lane &L)
{
    LaunchCtx c{};
    c.stream = L.stream;
    c.d = L.d;
    c.dv = L.dv;
    for (int i = 0; i < 2; i++) { c.w[i] = e->net[i].w; c.rw[i] = e->net[i].rw; }
    c.model = e->cfg.model;
    c.synthetic = e->cfg.eval_kind == AZ_EVAL_SYNTHETIC;
    c.persist_gp = e->persist_gp;
    c.vl_kernel = e->vl_kernel ? 1 : 0;
    c.emul = e->trunk_mode;        // AZ_TRUNK_F32 = 0, AZ_TRUNK_BF16X3 = EMUL_BF16X3, AZ_TRUNK_F16X2 = EMUL_F16X2
    c.feat = (float *)L.pol_feat.p;
    c.dbg = (unsigned long long *)L.dbg.p;
    c.scratch = (float *)L.scratch.p;
    return c;
}

// ------------------------------------------------------------------------------------------------
static int fail(az_engine *e, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_create_error = buf;
    return code;
}
#define HIPCHECK(e, call)                                                                          \
    do {                                                                                           \
        hipError_t _r = (call);                                                                    \
        if (_r != hipSuccess)                                                                      \
            return fail(e, AZ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

// The HIP current device is thread-local state the caller's framework (PyTorch) shares with this library: every entry
// point selects the engine's device for its own duration and puts the caller's device back on return.
struct DeviceGuard {
    int prev = -1;
    hipError_t rc;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        rc = prev == dev ? hipSuccess : hipSetDevice(dev);
        if (prev == dev) prev = -1;          // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define DEVICE_GUARD(e)                                                                            \
    DeviceGuard _guard((e)->cfg.device);                                                           \
    if (_guard.rc != hipSuccess) return fail(e, AZ_ERR_HIP, "hipSetDevice(%d) failed: %s", (e)->cfg.device, hipGetErrorString(_guard.rc))

static int dev_alloc(az_engine *e, DevBuf &b, size_t bytes, bool zero = true)
{
    if (b.p && b.bytes >= bytes) {
        if (zero) HIPCHECK(e, hipMemsetAsync(b.p, 0, bytes, e->stream));
        return AZ_OK;
    }
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    if (bytes == 0) bytes = 16;
    HIPCHECK(e, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    if (zero) HIPCHECK(e, hipMemsetAsync(b.p, 0, bytes, e->stream));
    return AZ_OK;
}
static void dev_free(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}
// Blocking copies go through the engine's own (non-blocking) stream: nothing in the engine touches the legacy null
// stream, whose implicit synchronisation would couple the engines of a process to each other and to PyTorch's work.
static hipError_t az_memcpy(hipStream_t s, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    hipError_t rc = hipMemcpyAsync(dst, src, bytes, kind, s);
    return rc == hipSuccess ? hipStreamSynchronize(s) : rc;
}
static hipError_t az_memcpy2d(hipStream_t s, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width,
                              size_t height, hipMemcpyKind kind)
{
    hipError_t rc = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, s);
    return rc == hipSuccess ? hipStreamSynchronize(s) : rc;
}
static int upload(az_engine *e, DevBuf &b, const void *src, size_t bytes)
{
    int rc = dev_alloc(e, b, bytes, false);
    if (rc) return rc;
    HIPCHECK(e, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    return AZ_OK;
}

// ------------------------------------------------------------------------------------------------
// weight packing: B-operand fragments of v_mfma_f32_16x16x4_f32.  Lane l supplies B[k = l>>4][j = l&15]
// of k-step s; four consecutive k-steps are stored together so one dwordx4 load feeds four MFMAs:
//   packed[((ntile*KS4 + s/4)*64 + lane)*4 + s%4]
// ------------------------------------------------------------------------------------------------
static std::vector<float> pack_conv(const float *w, int cout, int cin)
{
    // torch [co][ci][ky][kx]; k-step s = tap*(cin/4) + s', ci = 4*s' + (lane>>4)
    const int kst = cin / 4, ks = 9 * kst, ks4 = (ks + 3) / 4, nt = cout / 16;
    std::vector<float> out((size_t)nt * ks4 * 64 * 4, 0.0f);
    for (int t = 0; t < nt; t++)
        for (int s = 0; s < ks; s++)
            for (int lane = 0; lane < 64; lane++) {
                int tap = s / kst, sp = s % kst;
                int ci = 4 * sp + (lane >> 4), co = t * 16 + (lane & 15);
                out[(((size_t)t * ks4 + s / 4) * 64 + lane) * 4 + (s % 4)] = w[((size_t)co * cin + ci) * 9 + tap];
            }
    return out;
}
// fp32-emulating trunks (az_net_emul.h): A-operand fragments of the 16-bit MFMAs with every weight split into parts.
//   scheme 1 (AZ_TRUNK_BF16X3): w = hi + mid + lo, three bfloat16 parts, each the round-to-nearest-even bf16 of what the
//                               previous parts left over;
//   scheme 2 (AZ_TRUNK_F16X2):  w = hi + lo / 2048, two float16 parts, lo stored scaled by 2^11.
// Conv layers: lane l supplies A[row = l & 15][k = 8 (l >> 4) + j]; K-block kb = one tap x 32 input channels:
//   packed[(((ntile * KB + kb) * NS + part) * 64 + lane) * 8 + j]
static inline uint16_t bf16_rne(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_val(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f16_rne(float x)
{
    const _Float16 h = (_Float16)x;             // round to nearest even
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static inline float f16_val(uint16_t u)
{
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}
static inline int emul_parts(int scheme) { return scheme == AZ_TRUNK_F16X2 ? 2 : 3; }
// the parts of one weight; false when the value is outside the scheme's range
static inline bool emul_split(int scheme, float x, uint16_t *parts)
{
    if (scheme == AZ_TRUNK_F16X2) {
        if (!(std::fabs(x) < 65504.0f)) { parts[0] = parts[1] = 0; return false; }
        parts[0] = f16_rne(x);
        parts[1] = f16_rne((x - f16_val(parts[0])) * 2048.0f);
        return true;
    }
    parts[0] = bf16_rne(x);
    const float r1 = x - bf16_val(parts[0]);
    parts[1] = bf16_rne(r1);
    parts[2] = bf16_rne(r1 - bf16_val(parts[1]));
    return true;
}
// the first conv (cin = 4, K = 36 padded to two K-blocks of 32): k = tap * 4 + plane
static std::vector<uint16_t> pack_first_emul(int scheme, const float *w, int cout, bool *ok)
{
    const int ns = emul_parts(scheme), nt = cout / 16;
    std::vector<uint16_t> out((size_t)nt * 2 * ns * 64 * 8, 0);
    for (int t = 0; t < nt; t++)
        for (int kb = 0; kb < 2; kb++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int k = kb * 32 + 8 * (lane >> 4) + j, co = t * 16 + (lane & 15);
                    if (k >= 36) continue;
                    uint16_t parts[3];
                    if (!emul_split(scheme, w[((size_t)co * 4 + (k & 3)) * 9 + (k >> 2)], parts)) *ok = false;
                    const size_t base = (((size_t)t * 2 + kb) * ns * 64 + lane) * 8 + j;
                    for (int s2 = 0; s2 < ns; s2++) out[base + (size_t)s2 * 64 * 8] = parts[s2];
                }
    return out;
}
static std::vector<uint16_t> pack_conv_emul(int scheme, const float *w, int cout, int cin, bool *ok)
{
    const int ns = emul_parts(scheme), kbt = cin / 32, kb_n = 9 * kbt, nt = cout / 16;
    std::vector<uint16_t> out((size_t)nt * kb_n * ns * 64 * 8, 0);
    for (int t = 0; t < nt; t++)
        for (int kb = 0; kb < kb_n; kb++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int tap = kb / kbt, ci = (kb % kbt) * 32 + 8 * (lane >> 4) + j, co = t * 16 + (lane & 15);
                    uint16_t parts[3];
                    if (!emul_split(scheme, w[((size_t)co * cin + ci) * 9 + tap], parts)) *ok = false;
                    const size_t base = (((size_t)t * kb_n + kb) * ns * 64 + lane) * 8 + j;
                    for (int s2 = 0; s2 < ns; s2++) out[base + (size_t)s2 * 64 * 8] = parts[s2];
                }
    return out;
}
// The 1x1 head convs for the fused epilogue of the last emulated conv: A-operand fragments of the 16x16x16 MFMA, one per
// 16-channel tile of that conv's output; lane l supplies A[row = l & 15 = head channel][k = 4 (l >> 4) + j] = w[row][16 tile + k]:
//   packed[((tile * NS + part) * 64 + lane) * 4 + j]
static std::vector<uint16_t> pack_heads_emul(int scheme, const float *pw, int pc, const float *vw, int vc, int cin, bool *ok)
{
    const int ns = emul_parts(scheme), nt = cin / 16;
    std::vector<uint16_t> out((size_t)nt * ns * 64 * 4, 0);
    for (int t = 0; t < nt; t++)
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 4; j++) {
                const int row = lane & 15, ci = 16 * t + 4 * (lane >> 4) + j;
                const float x = row < pc ? pw[row * cin + ci] : (row < pc + vc ? vw[(row - pc) * cin + ci] : 0.0f);
                uint16_t parts[3];
                if (!emul_split(scheme, x, parts)) *ok = false;
                const size_t base = ((size_t)t * ns * 64 + lane) * 4 + j;
                for (int s2 = 0; s2 < ns; s2++) out[base + (size_t)s2 * 64 * 4] = parts[s2];
            }
    return out;
}
static std::vector<float> pack_heads(const float *pw, int pc, const float *vw, int vc, int cin)
{
    // policy_conv [pc][cin], value_conv [vc][cin] -> rows of one 16-row tile, cin/4 k-steps
    const int ks = cin / 4;
    std::vector<float> out((size_t)(ks / 4) * 64 * 4, 0.0f);
    for (int s = 0; s < ks; s++)
        for (int lane = 0; lane < 64; lane++) {
            int ci = 4 * s + (lane >> 4), j = lane & 15;
            float v = j < pc ? pw[j * cin + ci] : (j < pc + vc ? vw[(j - pc) * cin + ci] : 0.0f);
            out[(((size_t)s / 4) * 64 + lane) * 4 + (s % 4)] = v;
        }
    return out;
}
static std::vector<float> pack_fc(const float *w, int nout, int kin)
{
    // torch Linear [out][in]; tile t covers outputs 16t..16t+15; k-step s covers k = 4s..4s+3; group = 4 k-steps = 16 inputs.
    // Canonical order (az_net.h fc_chain_groups): 4 chains of qg groups each, zero-padded: [tile][4 qg][lane][4]
    const int nt = (nout + 15) / 16, ks = (kin + 3) / 4, qg = (((kin + 15) / 16) + 3) / 4, ks4 = 4 * qg;
    std::vector<float> out((size_t)nt * ks4 * 64 * 4, 0.0f);
    for (int t = 0; t < nt; t++)
        for (int s = 0; s < ks; s++)
            for (int lane = 0; lane < 64; lane++) {
                int k = 4 * s + (lane >> 4), j = t * 16 + (lane & 15);
                float v = (j < nout && k < kin) ? w[(size_t)j * kin + k] : 0.0f;
                out[(((size_t)t * ks4 + s / 4) * 64 + lane) * 4 + (s % 4)] = v;
            }
    return out;
}

// ------------------------------------------------------------------------------------------------
// kernel dispatch by board size: one translation unit per size (az_kernels.hip), selected once at az_create
// ------------------------------------------------------------------------------------------------
const SizeOps *az_size_ops(int n)
{
    switch (n) {
#define AZ_CASE(k) case k: return az_size_ops_##k ? az_size_ops_##k() : nullptr;
    AZ_CASE(3) AZ_CASE(4) AZ_CASE(5) AZ_CASE(6) AZ_CASE(7) AZ_CASE(8) AZ_CASE(9) AZ_CASE(10) AZ_CASE(11) AZ_CASE(12)
    AZ_CASE(13) AZ_CASE(14) AZ_CASE(15)
#undef AZ_CASE
    default: return nullptr;
    }
}


// ------------------------------------------------------------------------------------------------
// packed records + examples (self_play.py:94-108 augmentation fused into the encode)
// record layout: [8 x u64 planes (mover, opponent)] [pi f32 x nn] [last i16] [mover u8] [z i8], padded to 8 B
// ------------------------------------------------------------------------------------------------
static inline int64_t record_bytes(int nn) { return ((64 + 4 * (int64_t)nn + 4) + 7) / 8 * 8; }

__global__ void k_pack(DevState d, const int *src_index, int64_t records, int nn, int64_t rb, unsigned char *out)
{
    int64_t r = blockIdx.x;
    if (r >= records) return;
    int si = src_index[r];
    int g = si / nn;
    unsigned char *o = out + r * rb;
    u64 *op = reinterpret_cast<u64 *>(o);
    float *opi = reinterpret_cast<float *>(o + 64);
    if (threadIdx.x < 8) op[threadIdx.x] = d.rec_planes[(size_t)si * 8 + threadIdx.x];
    for (int j = threadIdx.x; j < nn; j += blockDim.x) opi[j] = d.rec_pi[(size_t)si * nn + j];
    if (threadIdx.x == 0) {
        short *ol = reinterpret_cast<short *>(o + 64 + 4 * nn);
        ol[0] = d.rec_last[si];
        int mover = d.rec_mover[si], res = d.g_result[g];
        o[64 + 4 * nn + 2] = (unsigned char)mover;
        // self_play.py:71: 0 if draw, +1 if the mover won, -1 otherwise; 99 marks a game cut by max_plies
        reinterpret_cast<signed char *>(o)[64 + 4 * nn + 3] = (signed char)(res == 0 ? 99 : (res == 3 ? 0 : (mover == res ? 1 : -1)));
    }
}

__global__ void k_examples(const unsigned char *packed, int64_t records, int n, int64_t rb, int aug, float *states,
                           float *pis, float *zs)
{
    const int nn = n * n;
    int64_t r = blockIdx.x;
    if (r >= records) return;
    const unsigned char *o = packed + r * rb;
    const u64 *pl = reinterpret_cast<const u64 *>(o);
    const float *pi = reinterpret_cast<const float *>(o + 64);
    const int last = reinterpret_cast<const short *>(o + 64 + 4 * nn)[0];
    const int z = reinterpret_cast<const signed char *>(o)[64 + 4 * nn + 3];
    for (int k = 0; k < aug; k++) {
        float *so = states + ((size_t)r * aug + k) * 4 * nn;
        float *po = pis + ((size_t)r * aug + k) * nn;
        for (int c = threadIdx.x; c < nn; c += blockDim.x) {
            int i = c / n, j = c - i * n;
            int src = sym_src(k, i, j, n);
            so[c] = ((pl[src >> 6] >> (src & 63)) & 1ull) ? 1.0f : 0.0f;            // games.py:117-124
            so[nn + c] = ((pl[4 + (src >> 6)] >> (src & 63)) & 1ull) ? 1.0f : 0.0f;
            so[2 * nn + c] = (src == last) ? 1.0f : 0.0f;                              // games.py:126-128
            so[3 * nn + c] = 0.0f;
            // reference mode (aug == 4): pi is rotated exactly once for every k (self_play.py:105)
            int psrc = aug == AZ_AUG_REFERENCE4 ? sym_src(1, i, j, n) : src;
            po[c] = pi[psrc];
        }
        if (threadIdx.x == 0) zs[(size_t)r * aug + k] = (float)z;
    }
}

// Training batch straight from a device-resident ring of packed records: example i = symmetry sym[i] of record
// idx[i] (replay_buffer.py:26-39 sample_batch + controller.py:23-31 collate, without the host round trip).
__global__ void k_examples_gather(const unsigned char *packed, const long long *idx, const int *sym, int count, int n,
                                  int64_t rb, int reference_pi, float *states, float *pis, float *zs)
{
    const int nn = n * n;
    const int i = blockIdx.x;
    if (i >= count) return;
    const unsigned char *o = packed + idx[i] * rb;
    const u64 *pl = reinterpret_cast<const u64 *>(o);
    const float *pi = reinterpret_cast<const float *>(o + 64);
    const int last = reinterpret_cast<const short *>(o + 64 + 4 * nn)[0];
    const int z = reinterpret_cast<const signed char *>(o)[64 + 4 * nn + 3];
    const int k = sym[i];
    float *so = states + (size_t)i * 4 * nn;
    float *po = pis + (size_t)i * nn;
    for (int c = threadIdx.x; c < nn; c += blockDim.x) {
        int r = c / n, q = c - r * n;
        int src = sym_src(k, r, q, n);
        so[c] = ((pl[src >> 6] >> (src & 63)) & 1ull) ? 1.0f : 0.0f;
        so[nn + c] = ((pl[4 + (src >> 6)] >> (src & 63)) & 1ull) ? 1.0f : 0.0f;
        so[2 * nn + c] = (src == last) ? 1.0f : 0.0f;
        so[3 * nn + c] = 0.0f;
        po[c] = pi[reference_pi ? sym_src(1, r, q, n) : src];
    }
    if (threadIdx.x == 0) zs[i] = (float)z;
}

// ------------------------------------------------------------------------------------------------
// lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" const char *az_last_error(const az_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

static int setup_tables(az_engine *e)
{
    const int S = e->cfg.num_simulations, nn = e->nn;
    std::vector<float> lt(S + 2);
    for (int i = 0; i <= S + 1; i++) lt[i] = e->cfg.log_table && i <= S ? e->cfg.log_table[i] : logf((float)i + 1e-8f);
    std::vector<double> st(S + 3);
    for (int i = 0; i <= S + 2; i++) st[i] = std::sqrt((double)i + 1e-8);
    std::vector<int> no(nn + 2);
    int off = 0;
    for (int m = 0; m <= nn; m++) { no[m] = off; off += nn - m; }
    e->tape_len = off;
    int rc;
    if ((rc = upload(e, e->log_table, lt.data(), lt.size() * sizeof(float)))) return rc;
    if ((rc = upload(e, e->sqrt_table, st.data(), st.size() * sizeof(double)))) return rc;
    if ((rc = upload(e, e->noise_off, no.data(), no.size() * sizeof(int)))) return rc;
    return AZ_OK;
}

// lanes of an engine when the caller leaves it to the library: small boards are launch-bound (one lane); otherwise one
// lane per 128 slots up to four (measured at 15x15 / 1024 slots: 1/2/3/4/6/8 lanes -> 2.38/2.47/2.53/2.58/1.95/2.11 M exp/s)
static int auto_lanes(int n, int slots)
{
    if (n <= 5) return 1;
    const int k = slots / 128;
    return k < 1 ? 1 : (k > 4 ? 4 : k);
}

// CPUs this process may run on: the affinity mask, capped by the cgroup CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us);
// std::thread::hardware_concurrency() knows about neither on every libstdc++
static int usable_cpus()
{
    int n = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    double quota = 0.0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64] = {0};
        long long period = 0;
        if (fscanf(f, "%63s %lld", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) quota = atof(a) / (double)period;
        fclose(f);
    } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        long long q = -1, period = 0;
        if (fscanf(f1, "%lld", &q) != 1) q = -1;
        fclose(f1);
        if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(f2, "%lld", &period) != 1) period = 0;
            fclose(f2);
        }
        if (q > 0 && period > 0) quota = (double)q / (double)period;
    }
    if (quota >= 1.0 && quota < (double)n) n = (int)quota;
    return n;
}

// every lane's DevState carries the shared pointers and scalars
template <class F>
static void each_state(az_engine *e, F f)
{
    for (Lane &L : e->lanes) {
        f(L.d);
        L.dv = L.d;                                  // the net kernels see evaluation items instead of slots
        L.dv.B = L.d.B * L.d.L;
        L.dv.s_status = L.d.it_status;
        L.dv.s_net = L.d.it_net;
    }
}

// (re)allocates the per-item buffers of a lane for `leaves` leaves per game and batch
static int alloc_items(az_engine *e, Lane &L, int leaves)
{
    const size_t NI = (size_t)L.d.B * leaves;
    int rc = AZ_OK;
#define ALLOC(buf, bytes) if (!rc) rc = dev_alloc(e, L.buf, (bytes))
    ALLOC(path, (NI * (size_t)e->PATH + 64) * 4);   // +64: k_step's speculative path[lane] read of the last slot
    ALLOC(depth, NI * 4);
    ALLOC(leaf_kind, NI * 4); ALLOC(leaf, NI * 8 * sizeof(u64)); ALLOC(leaf_last, NI * 4); ALLOC(leaf_sym, NI * 4);
    ALLOC(logits, NI * (size_t)e->RW * 4); ALLOC(vhid, NI * 64 * 4);
    ALLOC(pol_feat, NI * (size_t)((((e->cfg.model == AZ_MODEL_RESNET ? 3 : 6) * e->nn + 3) / 4) * 4) * 4);   // feature rows [NI][FROW], zero tail stays zero
#ifdef AZ_STAMPS
    ALLOC(dbg, (NI * 16 + 4096 * 32 + NI * 32) * sizeof(unsigned long long));     // trunk stamps | k_fc stamps | conv3 per-wave stamps
#endif
    if (e->split_max > 0)    // zeroed once: the padding ring of the packed images is never written afterwards
        ALLOC(scratch, (size_t)e->ops->split_scratch_floats((int)NI, e->cfg.model) * sizeof(float));
    if (leaves > 1) { ALLOC(it_status, NI * 4); ALLOC(it_net, NI * 4); }
#undef ALLOC
    if (rc) return rc;
    DevState &d = L.d;
    d.L = leaves;
    d.path = (unsigned *)L.path.p; d.depth = (int *)L.depth.p; d.leaf_kind = (int *)L.leaf_kind.p; d.leaf = (u64 *)L.leaf.p;
    d.leaf_last = (int *)L.leaf_last.p; d.logits = (float *)L.logits.p; d.vhid = (float *)L.vhid.p;
    d.leaf_sym = e->leaf_symmetry ? (int *)L.leaf_sym.p : nullptr;
    d.it_status = leaves > 1 ? (int *)L.it_status.p : d.s_status;
    d.it_net = leaves > 1 ? (int *)L.it_net.p : d.s_net;
    return AZ_OK;
}

extern "C" int az_create(const az_config *cfg, az_engine **out)
{
    if (!cfg || !out) return fail(nullptr, AZ_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->board_size < 3 || cfg->board_size > 15)
        return fail(nullptr, AZ_ERR_INVALID, "board_size must be 3..15 (got %d)", cfg->board_size);
    if (cfg->win_length < 2 || cfg->win_length > cfg->board_size) return fail(nullptr, AZ_ERR_INVALID, "bad win_length");
    if (cfg->num_simulations < 1 || cfg->num_simulations > 1024) return fail(nullptr, AZ_ERR_INVALID, "num_simulations must be 1..1024");
    if (cfg->slots < 1 || cfg->slots > 65536) return fail(nullptr, AZ_ERR_INVALID, "slots must be 1..65536");
    if (cfg->model != AZ_MODEL_PLAIN && cfg->model != AZ_MODEL_RESNET) return fail(nullptr, AZ_ERR_INVALID, "unknown model kind %d", cfg->model);
    if (cfg->engines < 0 || cfg->engines > 16) return fail(nullptr, AZ_ERR_INVALID, "engines must be 0 (auto) .. 16");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, AZ_ERR_NO_DEVICE, "no HIP device: this engine has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, AZ_ERR_INVALID, "device %d out of range (%d devices)", cfg->device, ndev);
    az_engine *e = new az_engine();
    e->cfg = *cfg;
    e->n = cfg->board_size;
    e->nn = e->n * e->n;
    e->RW = (e->nn + 63) / 64 * 64;
    e->R = cfg->num_simulations + 1;
    e->PATH = e->nn + 1;
    e->ops = az_size_ops(e->n);
    if (!e->ops) {
        delete e;
        return fail(nullptr, AZ_ERR_INVALID, "board size %d is not built into this library", cfg->board_size);
    }
    int K = cfg->engines > 0 ? cfg->engines : auto_lanes(e->n, cfg->slots);
    if (K > cfg->slots) K = cfg->slots;
    const int per = (cfg->slots + K - 1) / K;
    K = (cfg->slots + per - 1) / per;              // no empty lane
    e->cfg.engines = K;
    e->lanes.resize(K);
    DeviceGuard guard(cfg->device);          // the caller's current device is restored on return
    hipError_t hr = guard.rc;
    // Own hardware queue per lane: the runtime multiplexes the streams of one priority level over a small pool of
    // hardware queues, so next to a framework that has already created streams (PyTorch's context) two lanes can
    // land on one queue and stop overlapping (measured: episode 11.6 -> 15.3 s).  High-priority streams draw from
    // their own pool.  AZ_STREAM_PRIORITY=0 keeps the default priority.
    for (int i = 0; i < K && hr == hipSuccess; i++) {
        e->lanes[i].index = i;
        hr = g_streams.acquire(cfg->device, true, &e->lanes[i].stream);
    }
    if (hr != hipSuccess) {
        int rc = fail(nullptr, AZ_ERR_HIP, "device init failed: %s", hipGetErrorString(hr));
        az_destroy(e);
        return rc;
    }
    e->stream = e->lanes[0].stream;
    {
        const char *sm = getenv("AZ_SPLIT_MAX");      // 0 disables the split trunk, a large value forces it
        if (sm) e->split_max = atoi(sm);
    }
    int rc = AZ_OK;
    for (int i = 0; i < K && !rc; i++) {
        Lane &L = e->lanes[i];
        const size_t B = (size_t)std::min(per, cfg->slots - i * per);
#define ALLOC(buf, bytes) if (!rc) rc = dev_alloc(e, L.buf, (bytes))
        ALLOC(board, B * 8 * sizeof(u64));
        ALLOC(s_game, B * 4); ALLOC(s_ply, B * 4); ALLOC(s_player, B * 4); ALLOC(s_last, B * 4);
        ALLOC(s_status, B * 4); ALLOC(s_net, B * 4);
        ALLOC(edges, B * (size_t)e->R * e->RW * sizeof(Edge));
        ALLOC(rows_used, B * 4);
        ALLOC(cnt, B * CNT_STRIDE * sizeof(unsigned long long)); ALLOC(active_dev, 16);
        ALLOC(carried, B * 4);
#undef ALLOC
        DevState &d = L.d;
        d.B = (int)B; d.R = e->R; d.S = cfg->num_simulations; d.k = cfg->win_length;
        d.c_puct = cfg->c_puct; d.w_noise = cfg->dirichlet_weight;
        d.one_minus_w = (float)(1.0 - cfg->dirichlet_weight);   // Python float (1 - w) as a weak scalar -> float32 (Q8)
        d.board = (u64 *)L.board.p;
        d.s_game = (int *)L.s_game.p; d.s_ply = (int *)L.s_ply.p; d.s_player = (int *)L.s_player.p;
        d.s_last = (int *)L.s_last.p; d.s_status = (int *)L.s_status.p; d.s_net = (int *)L.s_net.p;
        d.edges = (Edge *)L.edges.p; d.rows_used = (int *)L.rows_used.p;
        d.cnt = (unsigned long long *)L.cnt.p; d.active = (int *)L.active_dev.p;
        d.carried = (int *)L.carried.p; d.reuse = 0;
        d.v2w[0] = d.v2w[1] = d.v2b[0] = d.v2b[1] = nullptr;
        d.cache = nullptr; d.cache_mask = 0; d.cache_gen = e->cache_gen; d.ext_eval = 0; d.leaf_sym = nullptr; d.game_key0 = 0;
        if (!rc) rc = alloc_items(e, L, 1);
    }
    if (!rc) rc = dev_alloc(e, e->next_game, 16);
    if (!rc) rc = dev_alloc(e, e->T_table, (size_t)(e->nn + 4) * sizeof(double));
    if (!rc) rc = setup_tables(e);
    if (rc) {
        g_create_error = e->err;
        az_destroy(e);
        return rc;
    }
    each_state(e, [&](DevState &d) {
        d.T_table = (const double *)e->T_table.p; d.log_table = (const float *)e->log_table.p;
        d.sqrt_table = (const double *)e->sqrt_table.p; d.noise_off = (const int *)e->noise_off.p;
        d.next_game = (int *)e->next_game.p;
    });
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) e->num_cus = prop.multiProcessorCount;
    }
    const char *pe = getenv("AZ_PROFILE_EVENTS");
    e->profile = pe && pe[0] == '1';
    const char *pz = getenv("AZ_PERSIST");
    e->persist_allowed = !(pz && pz[0] == '0');
    const char *cz = getenv("AZ_COMPACT");
    e->compact = !(cz && cz[0] == '0');
    const char *ge = getenv("AZ_GRAPH");
    e->use_graph = !(ge && ge[0] == '0');
    const char *ts = getenv("AZ_TAPE_STREAM");
    e->stream_tapes = !(ts && ts[0] == '0');
    {
        // Tape producer threads.  What this process may use is its affinity mask and cgroup CPU quota, not the machine's
        // hardware threads (a GPU box grants a job a share of its cores); the ranks of a node split that between them
        // unless the launcher already pinned each rank to its own cores (then the mask is per rank and the split would
        // count twice -- AZ_HOST_CPUS_PER_RANK=1 says so).  K lane threads drive the play streams; the rest, up to 32,
        // produce tapes.
        e->host_cpus = usable_cpus();
        const char *lw = getenv("LOCAL_WORLD_SIZE");
        const char *pr = getenv("AZ_HOST_CPUS_PER_RANK");
        const int ranks = (pr && pr[0] == '1') ? 1 : (lw && atoi(lw) > 0 ? atoi(lw) : 1);
        int t = e->host_cpus / ranks - K;
        const char *tt = getenv("AZ_TAPE_THREADS");
        if (tt) t = atoi(tt);
        e->tape_threads = t < 1 ? 1 : (t > 32 ? 32 : t);
    }
    if (hipStreamSynchronize(e->stream) != hipSuccess) {
        g_create_error = "stream sync failed in az_create";
        az_destroy(e);
        return AZ_ERR_HIP;
    }
    *out = e;
    return AZ_OK;
}

static void dist_destroy(az_engine *e);

extern "C" void az_destroy(az_engine *e)
{
    if (!e) return;
    DeviceGuard guard(e->cfg.device);
    stop_tapes(e);
    for (Lane &L : e->lanes)
        if (L.stream) (void)hipStreamSynchronize(L.stream);
    dist_destroy(e);
    for (Lane &L : e->lanes) {
        DevBuf *all[] = {&L.board, &L.s_game, &L.s_ply, &L.s_player, &L.s_last, &L.s_status, &L.s_net, &L.edges, &L.rows_used,
                         &L.path, &L.depth, &L.leaf_kind, &L.leaf, &L.leaf_last, &L.logits, &L.vhid, &L.pol_feat, &L.cnt,
                         &L.active_dev, &L.carried, &L.dbg, &L.scratch, &L.it_status, &L.it_net, &L.leaf_sym};
        for (DevBuf *b : all) dev_free(*b);
        for (hipEvent_t ev : L.ev) (void)hipEventDestroy(ev);
        for (int i = 0; i < 4; i++)
            if (L.graph[i >> 1][i & 1].exec) (void)hipGraphExecDestroy(L.graph[i >> 1][i & 1].exec);
    }
    DevBuf *shared[] = {&e->T_table, &e->log_table, &e->sqrt_table, &e->noise_off, &e->next_game, &e->noise, &e->u,
                        &e->rec_planes, &e->rec_last, &e->rec_action, &e->rec_mover, &e->rec_pi, &e->rec_visits, &e->g_nply,
                        &e->g_result, &e->src_index, &e->cache};
    for (DevBuf *b : shared) dev_free(*b);
    for (int s = 0; s < 2; s++) {
        PackedNet &p = e->net[s];
        DevBuf *nb[] = {&p.c1, &p.c2, &p.c3, &p.hd, &p.pf, &p.vf, &p.c1b, &p.c2b, &p.c3b, &p.hdb, &p.pfb, &p.vfb, &p.v2w, &p.v2b, &p.c1x[0], &p.c1x[1], &p.c2x[0], &p.c2x[1], &p.c3x[0], &p.c3x[1], &p.hdx[0], &p.hdx[1]};
        for (DevBuf *b : nb) dev_free(*b);
        for (int i = 0; i < 6; i++) { dev_free(p.rblk[i]); dev_free(p.rblkb[i]); dev_free(p.rblkx[0][i]); dev_free(p.rblkx[1][i]); }
    }
    for (Lane &L : e->lanes)
        if (L.stream) g_streams.release(e->cfg.device, true, L.stream);
    delete e;
}

// every weight of the layers the emulated trunks run is inside float16's range (the only part of the split that is checked
// eagerly at az_load_weights*; NaN counts as outside)
static bool in_f16_range(const PackedNet &p)
{
    auto ok = [](const std::vector<float> &v) {
        for (float x : v)
            if (!(std::fabs(x) < 65504.0f)) return false;
        return true;
    };
    bool r = ok(p.hw_first) && ok(p.hw_pol) && ok(p.hw_val);
    for (int i = 0; i < 6; i++) r = r && ok(p.hw_conv[i]);
    return r;
}

// packs and uploads the split fragments of weight slot `slot` for `scheme` unless that has been done since the last load
static int ensure_emul(az_engine *e, int slot, int scheme)
{
    PackedNet &p = e->net[slot];
    const int si = scheme - 1;
    if (!p.loaded || scheme == AZ_TRUNK_F32 || p.emul_ready[si]) return AZ_OK;
    bool ok = true;
    int rc = AZ_OK;
    auto upx = [&](DevBuf &b, const std::vector<uint16_t> &x) { if (!rc) rc = upload(e, b, x.data(), x.size() * 2); };
    if (e->cfg.model == AZ_MODEL_PLAIN) {
        upx(p.c1x[si], pack_first_emul(scheme, p.hw_first.data(), 32, &ok));
        upx(p.c2x[si], pack_conv_emul(scheme, p.hw_conv[0].data(), 64, 32, &ok));
        upx(p.c3x[si], pack_conv_emul(scheme, p.hw_conv[1].data(), 128, 64, &ok));
        upx(p.hdx[si], pack_heads_emul(scheme, p.hw_pol.data(), 4, p.hw_val.data(), 2, 128, &ok));
        if (rc) return rc;
        p.w.c1x[si] = p.c1x[si].p; p.w.c2x[si] = p.c2x[si].p; p.w.c3x[si] = p.c3x[si].p; p.w.hdx[si] = p.hdx[si].p;
    } else {
        upx(p.c1x[si], pack_first_emul(scheme, p.hw_first.data(), 64, &ok));
        for (int i = 0; i < 6; i++) upx(p.rblkx[si][i], pack_conv_emul(scheme, p.hw_conv[i].data(), 64, 64, &ok));
        upx(p.hdx[si], pack_heads_emul(scheme, p.hw_pol.data(), 2, p.hw_val.data(), 1, 64, &ok));
        if (rc) return rc;
        for (int i = 0; i < 6; i++) p.rw.blkx[si][i] = p.rblkx[si][i].p;
        p.rw.hdx[si] = p.hdx[si].p;
        p.rw.stemx[si] = p.c1x[si].p;
    }
    p.emul_ready[si] = true;
    return AZ_OK;
}

extern "C" int az_load_weights(az_engine *e, int slot, const float *const *t)
{
    if (!e || !t || slot < 0 || slot > 1) return fail(e, AZ_ERR_INVALID, "az_load_weights: bad argument");
    if (e->cfg.model != AZ_MODEL_PLAIN) return fail(e, AZ_ERR_INVALID, "az_load_weights: engine was created for the ResidualBlock model");
    for (int i = 0; i < 16; i++)
        if (!t[i]) return fail(e, AZ_ERR_INVALID, "az_load_weights: tensor %d is null", i);
    DEVICE_GUARD(e);
    const int nn = e->nn;
    PackedNet &p = e->net[slot];
    int rc = AZ_OK;
    auto up = [&](DevBuf &b, const std::vector<float> &v) { if (!rc) rc = upload(e, b, v.data(), v.size() * sizeof(float)); };
    auto upraw = [&](DevBuf &b, const float *v, size_t cnt) { if (!rc) rc = upload(e, b, v, cnt * sizeof(float)); };
    up(p.c1, pack_conv(t[0], 32, 4));   upraw(p.c1b, t[1], 32);
    up(p.c2, pack_conv(t[2], 64, 32));  upraw(p.c2b, t[3], 64);
    up(p.c3, pack_conv(t[4], 128, 64)); upraw(p.c3b, t[5], 128);
    p.hw_first.assign(t[0], t[0] + 32 * 4 * 9);
    p.hw_conv[0].assign(t[2], t[2] + 64 * 32 * 9);
    p.hw_conv[1].assign(t[4], t[4] + 128 * 64 * 9);
    p.hw_pol.assign(t[6], t[6] + 4 * 128);
    p.hw_val.assign(t[10], t[10] + 2 * 128);
    p.emul_ready[0] = p.emul_ready[1] = false;
    p.f16_ok = in_f16_range(p);
    up(p.hd, pack_heads(t[6], 4, t[10], 2, 128));
    float hb[6] = {t[7][0], t[7][1], t[7][2], t[7][3], t[11][0], t[11][1]};
    upraw(p.hdb, hb, 6);
    up(p.pf, pack_fc(t[8], nn, 4 * nn));  upraw(p.pfb, t[9], nn);
    up(p.vf, pack_fc(t[12], 64, 2 * nn)); upraw(p.vfb, t[13], 64);
    upraw(p.v2w, t[14], 64); upraw(p.v2b, t[15], 1);
    if (rc) return rc;
    p.w.c1 = (const float *)p.c1.p; p.w.c2 = (const float *)p.c2.p; p.w.c3 = (const float *)p.c3.p;
    p.w.hd = (const float *)p.hd.p; p.w.pf = (const float *)p.pf.p; p.w.vf = (const float *)p.vf.p;
    p.w.c1b = (const float *)p.c1b.p; p.w.c2b = (const float *)p.c2b.p; p.w.c3b = (const float *)p.c3b.p;
    p.w.hdb = (const float *)p.hdb.p; p.w.pfb = (const float *)p.pfb.p; p.w.vfb = (const float *)p.vfb.p;
    e->cache_gen++;           // evaluations cached under the previous weights never match again
    each_state(e, [&](DevState &d) { d.v2w[slot] = (const float *)p.v2w.p; d.v2b[slot] = (const float *)p.v2b.p; d.cache_gen = e->cache_gen; });
    p.loaded = true;
    if (e->trunk_mode == AZ_TRUNK_F16X2 && !p.f16_ok) {
        e->trunk_mode = AZ_TRUNK_F32;       // never run a net outside float16's range in the float16 scheme
        return fail(e, AZ_ERR_INVALID, "weights outside float16's range (|w| >= 65504): AZ_TRUNK_F16X2 switched off, the engine is back on AZ_TRUNK_F32");
    }
    return e->trunk_mode != AZ_TRUNK_F32 ? ensure_emul(e, slot, e->trunk_mode) : AZ_OK;
}

extern "C" int az_load_weights_resnet(az_engine *e, int slot, const float *const *t)
{
    if (!e || !t || slot < 0 || slot > 1) return fail(e, AZ_ERR_INVALID, "az_load_weights_resnet: bad argument");
    if (e->cfg.model != AZ_MODEL_RESNET) return fail(e, AZ_ERR_INVALID, "az_load_weights_resnet: engine was created for the plain model");
    for (int i = 0; i < 24; i++)
        if (!t[i]) return fail(e, AZ_ERR_INVALID, "az_load_weights_resnet: tensor %d is null", i);
    DEVICE_GUARD(e);
    const int nn = e->nn;
    PackedNet &p = e->net[slot];
    int rc = AZ_OK;
    auto up = [&](DevBuf &b, const std::vector<float> &v) { if (!rc) rc = upload(e, b, v.data(), v.size() * sizeof(float)); };
    auto upraw = [&](DevBuf &b, const float *v, size_t cnt) { if (!rc) rc = upload(e, b, v, cnt * sizeof(float)); };
    up(p.c1, pack_conv(t[0], 64, 4)); upraw(p.c1b, t[1], 64);
    for (int i = 0; i < 6; i++) { up(p.rblk[i], pack_conv(t[2 + 2 * i], 64, 64)); upraw(p.rblkb[i], t[3 + 2 * i], 64); }
    p.hw_first.assign(t[0], t[0] + 64 * 4 * 9);
    for (int i = 0; i < 6; i++) p.hw_conv[i].assign(t[2 + 2 * i], t[2 + 2 * i] + 64 * 64 * 9);
    p.hw_pol.assign(t[14], t[14] + 2 * 64);
    p.hw_val.assign(t[16], t[16] + 64);
    p.emul_ready[0] = p.emul_ready[1] = false;
    p.f16_ok = in_f16_range(p);
    up(p.hd, pack_heads(t[14], 2, t[16], 1, 64));
    float hb[3] = {t[15][0], t[15][1], t[17][0]};
    upraw(p.hdb, hb, 3);
    up(p.pf, pack_fc(t[18], nn, 2 * nn)); upraw(p.pfb, t[19], nn);
    up(p.vf, pack_fc(t[20], 64, nn));     upraw(p.vfb, t[21], 64);
    upraw(p.v2w, t[22], 64); upraw(p.v2b, t[23], 1);
    if (rc) return rc;
    p.rw.stem = (const float *)p.c1.p; p.rw.stemb = (const float *)p.c1b.p;
    for (int i = 0; i < 6; i++) { p.rw.blk[i] = (const float *)p.rblk[i].p; p.rw.blkb[i] = (const float *)p.rblkb[i].p; }
    p.rw.hd = (const float *)p.hd.p; p.rw.hdb = (const float *)p.hdb.p;
    p.w = NetWeights{};
    p.w.pf = (const float *)p.pf.p; p.w.vf = (const float *)p.vf.p;
    p.w.pfb = (const float *)p.pfb.p; p.w.vfb = (const float *)p.vfb.p;
    e->cache_gen++;           // evaluations cached under the previous weights never match again
    each_state(e, [&](DevState &d) { d.v2w[slot] = (const float *)p.v2w.p; d.v2b[slot] = (const float *)p.v2b.p; d.cache_gen = e->cache_gen; });
    p.loaded = true;
    if (e->trunk_mode == AZ_TRUNK_F16X2 && !p.f16_ok) {
        e->trunk_mode = AZ_TRUNK_F32;       // never run a net outside float16's range in the float16 scheme
        return fail(e, AZ_ERR_INVALID, "weights outside float16's range (|w| >= 65504): AZ_TRUNK_F16X2 switched off, the engine is back on AZ_TRUNK_F32");
    }
    return e->trunk_mode != AZ_TRUNK_F32 ? ensure_emul(e, slot, e->trunk_mode) : AZ_OK;
}

// ------------------------------------------------------------------------------------------------
// episode machinery
// ------------------------------------------------------------------------------------------------
static int ensure_episode_buffers(az_engine *e, int games, bool need_noise)
{
    const size_t nn = e->nn, G = (size_t)games;
    int rc = AZ_OK;
    stop_tapes(e);            // a producer of an abandoned episode still writes into the buffers below
#define ALLOC(buf, bytes, zero) if (!rc) rc = dev_alloc(e, e->buf, (bytes), zero)
    ALLOC(rec_planes, G * nn * 8 * sizeof(u64), false);
    ALLOC(rec_last, G * nn * 2, false); ALLOC(rec_action, G * nn * 2, false); ALLOC(rec_mover, G * nn, false);
    ALLOC(rec_pi, G * nn * nn * 4, false); ALLOC(rec_visits, G * nn * nn * 2, false);
    ALLOC(g_nply, G * 4, true); ALLOC(g_result, G * 4, true);
    ALLOC(u, G * nn * sizeof(double), false);
    if (need_noise) ALLOC(noise, G * (size_t)e->tape_len * sizeof(double), false);
#undef ALLOC
    if (rc) return rc;
    each_state(e, [&](DevState &d) {
        d.rec_planes = (u64 *)e->rec_planes.p; d.rec_last = (short *)e->rec_last.p; d.rec_action = (short *)e->rec_action.p;
        d.rec_mover = (unsigned char *)e->rec_mover.p; d.rec_pi = (float *)e->rec_pi.p;
        d.rec_visits = (unsigned short *)e->rec_visits.p; d.g_nply = (int *)e->g_nply.p; d.g_result = (int *)e->g_result.p;
        d.u = (const double *)e->u.p; d.noise = (const double *)e->noise.p; d.noise_stride = e->tape_len;
    });
    return AZ_OK;
}

struct EpisodeSpec {
    int num_games = 0, max_plies = 0;
    bool add_noise = true, arena = false;
    bool preset = false;      // slot 0 of lane 0 already holds a position (az_search); skip the initial refill
    bool profile = true;      // this episode may be timed with HIP events (only when az_set_profiling is on)
    unsigned game_key0 = 0;   // low 32 bits of seed0: the leaf-symmetry hash names game g by its seed, seed0 + g
};

static int host_threads()
{
    int t = usable_cpus();
    const char *ev = getenv("AZ_HOST_THREADS");
    if (ev) t = atoi(ev);
    return t < 1 ? 1 : (t > 64 ? 64 : t);
}

// errors inside a lane's host thread are kept on the lane and reported by the joining caller
static int lane_fail(Lane &L, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    L.err = buf;
    L.rc = code;
    return code;
}
#define HIPCHECK_L(L, call)                                                                        \
    do {                                                                                           \
        hipError_t _r = (call);                                                                    \
        if (_r != hipSuccess)                                                                      \
            return lane_fail(L, AZ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

static int episode_begin(az_engine *e, const EpisodeSpec &sp)
{
    const bool net = e->cfg.eval_kind == AZ_EVAL_NET;
    if (net && !e->net[0].loaded) return fail(e, AZ_ERR_NO_WEIGHTS, "weights slot 0 not loaded");
    if (net && sp.arena && !e->net[1].loaded) return fail(e, AZ_ERR_NO_WEIGHTS, "weights slot 1 (baseline) not loaded");
    each_state(e, [&](DevState &d) {
        d.max_plies = sp.max_plies; d.add_noise = sp.add_noise ? 1 : 0; d.arena = sp.arena ? 1 : 0;
        d.total_games = sp.num_games; d.reuse = e->reuse; d.game_key0 = sp.game_key0;
        d.cache = (float *)e->cache.p; d.cache_mask = e->cache_mask; d.cache_gen = e->cache_gen;
        d.ext_eval = 0;
    });
    // persistent search kernel: plain net or synthetic evaluator, the reference's sequential search, trees that fit into LDS
    e->persist_gp = 0;
    if (e->persist_allowed && !e->vl_kernel && e->trunk_mode == AZ_TRUNK_F32) {
        const int synth = e->cfg.eval_kind == AZ_EVAL_SYNTHETIC ? 1 : 0;
        const int S = e->cfg.num_simulations;
        if (!sp.arena && !sp.preset && e->ops->search_prepare(S, 2, synth, e->cfg.model)) e->persist_gp = 2;
        else if (e->ops->search_prepare(S, 1, synth, e->cfg.model)) e->persist_gp = 1;     // arena: a workgroup's games must share one net
    }
    az_engine::Run &r = e->run;
    r = az_engine::Run();
    r.num_games = sp.num_games; r.max_plies = sp.max_plies; r.add_noise = sp.add_noise; r.arena = sp.arena;
    r.preset = sp.preset; r.profile = sp.profile;
    HIPCHECK(e, hipMemsetAsync(e->next_game.p, 0, 16, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    // the first refill hands every lane an equal share of the games (as many as its slots hold); afterwards a freed
    // slot of ANY lane takes the next id of the shared queue, so no lane idles while another still has games waiting
    const int K = (int)e->lanes.size();
    const int share = (sp.num_games + K - 1) / K;
    for (Lane &L : e->lanes) {
        L.active = 0; L.cur = L.d.B; L.plies_played = 0; L.steps = L.trunk_launches = L.plies = 0;
        L.trunk_ms = L.nn_ms = L.step_ms = 0.0; L.tape_wait_s = 0.0; L.rc = AZ_OK; L.err.clear();
        HIPCHECK(e, hipMemsetAsync(L.cnt.p, 0, L.cnt.bytes, L.stream));
        HIPCHECK(e, hipMemsetAsync(L.carried.p, 0xFF, L.carried.bytes, L.stream));      // -1: every slot starts from a fresh root
        if (!sp.preset) {
            HIPCHECK(e, hipMemsetAsync(L.s_status.p, 0, L.s_status.bytes, L.stream));
            hipLaunchKernelGGL(k_refill, dim3(1), dim3(1024), 0, L.stream, L.d, share, e->compact ? 1 : 0);
            HIPCHECK(e, hipMemcpyAsync(&L.active, L.active_dev.p, 4, hipMemcpyDeviceToHost, L.stream));
        } else if (L.index != 0) {
            HIPCHECK(e, hipMemsetAsync(L.s_status.p, 0, L.s_status.bytes, L.stream));
        }
        HIPCHECK(e, hipStreamSynchronize(L.stream));
        if (!sp.preset && e->compact && !e->reuse) L.cur = L.active;       // the active slots are the first L.active ones
    }
    if (sp.preset) e->lanes[0].active = 1;
    r.open = true;
    e->have_episode = false;
    return AZ_OK;
}

// Evaluation batches of one ply: the root's evaluation, then the simulations -- one per batch in the reference's
// sequential loop (mcts.py:123-141), `vl` per batch with virtual-loss batching.
static int ply_batches(const az_engine *e) { return 1 + (e->vl_kernel ? (e->cfg.num_simulations + e->vl - 1) / e->vl : e->cfg.num_simulations); }

// the tree kernel that consumes evaluation batch `idx` (0 = the root evaluation) and selects the leaves of batch idx + 1
static void launch_step(const az_engine *e, const LaunchCtx &lc, int idx)
{
    const int S = lc.d.S;
    if (e->vl_kernel) {
        const int done = std::min(idx * e->vl, S);             // simulations finished once batch idx is consumed
        e->ops->step_vl(lc, done, std::min(S - done, e->vl));
    } else {
        e->ops->step(lc, idx, idx < S ? 1 : 0);                // root N = idx for the next selection
    }
}

// the kernel sequence of one ply
static void launch_ply(az_engine *e, const LaunchCtx &lc, bool use_split, int nnets, bool net)
{
    const DevState &d = lc.d;
    hipLaunchKernelGGL(k_begin, dim3((d.B + 255) / 256), dim3(256), 0, lc.stream, d);
    if (e->persist_gp) {               // small boards: the whole search of the ply in one persistent kernel (az_search.h)
        e->ops->search(lc, e->persist_gp);
        e->ops->move(lc);
        return;
    }
    if (net && d.cache) e->ops->root_cache(lc);
    const int nb = ply_batches(e);
    for (int idx = 0; idx < nb; idx++) {
        if (net) {
            for (int id = 0; id < nnets; id++) { if (use_split) e->ops->trunk_split(lc, id); else e->ops->trunk(lc, id); }
            for (int id = 0; id < nnets; id++) e->ops->fc(lc, id);
        }
        launch_step(e, lc, idx);
    }
    e->ops->move(lc);
}

static std::mutex g_capture_mutex;     // lanes of one process capture one at a time (instantiate is not cheap, and rare)

// returns the instantiated graph of one ply for these kernel arguments, capturing it on first use
static int ply_graph(az_engine *e, Lane &L, const LaunchCtx &lc, bool use_split, int nnets, bool net, hipGraphExec_t *out)
{
    Lane::PlyGraph &g = L.graph[use_split ? 1 : 0][nnets - 1];
    if (g.exec && memcmp(&g.key, &lc, sizeof(LaunchCtx)) == 0) { *out = g.exec; return AZ_OK; }
    std::lock_guard<std::mutex> lock(g_capture_mutex);
    if (g.exec) { HIPCHECK_L(L, hipGraphExecDestroy(g.exec)); g.exec = nullptr; }
    hipGraph_t graph = nullptr;
    HIPCHECK_L(L, hipStreamBeginCapture(lc.stream, hipStreamCaptureModeThreadLocal));
    launch_ply(e, lc, use_split, nnets, net);
    HIPCHECK_L(L, hipStreamEndCapture(lc.stream, &graph));
    hipError_t rc = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (rc != hipSuccess) { g.exec = nullptr; return lane_fail(L, AZ_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(rc)); }
    memcpy(&g.key, &lc, sizeof(LaunchCtx));
    *out = g.exec;
    return AZ_OK;
}

// one lane plays up to max_steps plies of its active slots in lock step (one "step" = MCTS.run for each active game);
// runs on the lane's own host thread
static int lane_plies(az_engine *e, Lane &L, int max_steps)
{
    const az_engine::Run &r = e->run;
    const int S = L.d.S;
    const bool net = e->cfg.eval_kind == AZ_EVAL_NET;
    const bool prof = net && r.profile && e->profile;
    const int nbat = ply_batches(e);
    const size_t need_ev = prof ? (size_t)4 * nbat : 0;
    while (L.ev.size() < need_ev) {
        hipEvent_t ev;
        HIPCHECK_L(L, hipEventCreate(&ev));
        L.ev.push_back(ev);
    }
    const int nnets = r.arena ? 2 : 1;
    const LaunchCtx lc_full = ctx_of_impl(e, L);
    // timing-only diagnostics (results are wrong): AZ_DIAG_SKIP=fc | step | fcstep
    const char *skip = getenv("AZ_DIAG_SKIP");
    const bool skip_fc = skip && strstr(skip, "fc"), skip_step = skip && strstr(skip, "step");
    for (int step = 0; step < max_steps && L.active > 0; step++) {
        // this ply runs over the first L.cur slots: all of them, or -- after a compacting refill -- exactly the active ones
        LaunchCtx lc = lc_full;
        lc.d.B = L.cur;
        lc.dv.B = L.cur * L.d.L;
        const DevState &d = lc.d;
        const bool full = L.cur == L.d.B;          // the captured graph holds the full grid; a shrunken ply is launched kernel by kernel
        // few pending boards: latency path (the emulated trunk has one kernel, so that its results never depend on the slot count)
        const bool use_split = e->split_max > 0 && L.scratch.p && L.active * e->vl <= e->split_max && e->trunk_mode == AZ_TRUNK_F32;
        if (e->tapes) {       // the tapes of this ply must be on the device (streamed a wave ahead of the games)
            hipEvent_t ev = nullptr;
            const auto tw0 = std::chrono::steady_clock::now();
            hipError_t trc = e->tapes->need(L.plies_played, &ev);
            if (trc != hipSuccess) return lane_fail(L, AZ_ERR_HIP, "tape producer: %s", hipGetErrorString(trc));
            // a wave whose upload is still in flight would stall the play stream by the same amount: wait for it here so
            // that the stall is counted (az_counters.tape_wait_seconds); normally the wave landed a ply or two ago
            if (ev && hipEventQuery(ev) != hipSuccess) HIPCHECK_L(L, hipEventSynchronize(ev));
            L.tape_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
            if (ev) HIPCHECK_L(L, hipStreamWaitEvent(L.stream, ev, 0));
        }
        if (e->use_graph && !prof && !skip && full) {
            hipGraphExec_t exec = nullptr;
            int rcg = ply_graph(e, L, lc, use_split, nnets, net, &exec);
            if (rcg) return rcg;
            HIPCHECK_L(L, hipGraphLaunch(exec, L.stream));
            if (net) L.trunk_launches += e->persist_gp ? 1 : (int64_t)nnets * nbat;
            L.steps += nbat;
        } else {
            hipLaunchKernelGGL(k_begin, dim3((d.B + 255) / 256), dim3(256), 0, L.stream, d);
            if (net && d.cache && !e->persist_gp) e->ops->root_cache(lc);
            if (e->persist_gp) {
                // one launch does the whole search; the events time it as a whole (it contains the conv trunk, the FC
                // layers and the tree steps of S + 1 evaluation batches)
                if (prof) HIPCHECK_L(L, hipEventRecord(L.ev[0], L.stream));
                e->ops->search(lc, e->persist_gp);
                if (prof) {
                    HIPCHECK_L(L, hipEventRecord(L.ev[1], L.stream));
                    HIPCHECK_L(L, hipEventRecord(L.ev[2], L.stream));
                    HIPCHECK_L(L, hipEventRecord(L.ev[3], L.stream));
                }
                if (net) L.trunk_launches += 1;
                L.steps += nbat;
            }
            for (int idx = 0; idx < nbat && !e->persist_gp; idx++) {
                if (net) {
                    const int ei = 4 * idx;
                    if (prof) HIPCHECK_L(L, hipEventRecord(L.ev[ei], L.stream));
                    for (int id = 0; id < nnets; id++) { if (use_split) e->ops->trunk_split(lc, id); else e->ops->trunk(lc, id); }
                    if (prof) HIPCHECK_L(L, hipEventRecord(L.ev[ei + 1], L.stream));
                    if (!skip_fc) for (int id = 0; id < nnets; id++) e->ops->fc(lc, id);
                    if (prof) HIPCHECK_L(L, hipEventRecord(L.ev[ei + 2], L.stream));
                    L.trunk_launches += nnets;
                }
                if (!skip_step || idx == 0) launch_step(e, lc, idx);
                if (prof) HIPCHECK_L(L, hipEventRecord(L.ev[4 * idx + 3], L.stream));
                L.steps++;
            }
            e->ops->move(lc);
        }
        L.plies += L.active;
        if (r.preset) {
            L.active = 0;
        } else {
            hipLaunchKernelGGL(k_refill, dim3(1), dim3(1024), 0, L.stream, d, 1 << 30, e->compact ? 1 : 0);
            HIPCHECK_L(L, hipMemcpyAsync(&L.active, L.active_dev.p, 4, hipMemcpyDeviceToHost, L.stream));
        }
        if (e->tapes)         // finished games need no further tape (g_nply only ever goes from 0 to the game's length, so
                              // a snapshot that lands late can only cost the producer some extra work)
            HIPCHECK_L(L, hipMemcpyAsync(e->tapes->h_done, e->g_nply.p, (size_t)r.num_games * 4, hipMemcpyDeviceToHost, L.stream));
        HIPCHECK_L(L, hipStreamSynchronize(L.stream));
        HIPCHECK_L(L, hipGetLastError());
        if (!r.preset && e->compact && !e->reuse && L.active > 0) L.cur = L.active;
        L.plies_played++;
        if (prof) {
            for (int i = 0; i < (e->persist_gp ? 1 : nbat); i++) {
                float a = 0.f, b = 0.f, c = 0.f;
                HIPCHECK_L(L, hipEventElapsedTime(&a, L.ev[4 * i], L.ev[4 * i + 1]));
                HIPCHECK_L(L, hipEventElapsedTime(&b, L.ev[4 * i + 1], L.ev[4 * i + 2]));
                HIPCHECK_L(L, hipEventElapsedTime(&c, L.ev[4 * i + 2], L.ev[4 * i + 3]));
                L.trunk_ms += a;
                L.nn_ms += a + b;
                L.step_ms += c;
            }
        }
    }
    return AZ_OK;
}

// every lane plays up to max_steps plies, each on its own host thread (the caller's thread drives lane 0).  With
// profiling on the lanes play one after another instead, so that the HIP events time each kernel alone on the GPU.
static int episode_plies(az_engine *e, int max_steps)
{
    az_engine::Run &r = e->run;
    if (!r.open) return fail(e, AZ_ERR_STATE, "no episode is open");
    const int K = (int)e->lanes.size();
    const bool sequential = K == 1 || (e->profile && r.profile);
    auto t0 = std::chrono::steady_clock::now();
    if (max_steps == 0) {
        // nothing to play (callers read the counters this way): no threads
    } else if (sequential) {
        for (Lane &L : e->lanes)
            if (lane_plies(e, L, max_steps)) break;
    } else {
        // nothing may unwind through the C ABI: a thread that cannot be created (std::system_error at a thread limit,
        // bad_alloc) leaves its lane to the calling thread, which then plays it after its own
        std::vector<std::thread> th;
        std::vector<int> inline_lanes;
        const int dev = e->cfg.device;
        for (int i = 1; i < K; i++) {
            try {
                th.emplace_back([e, i, max_steps, dev]() {
                    Lane &L = e->lanes[i];
                    if (hipSetDevice(dev) != hipSuccess) { lane_fail(L, AZ_ERR_HIP, "hipSetDevice(%d) failed on a lane thread", dev); return; }
                    lane_plies(e, L, max_steps);
                });
            } catch (...) {
                inline_lanes.push_back(i);
            }
        }
        lane_plies(e, e->lanes[0], max_steps);
        for (int i : inline_lanes) lane_plies(e, e->lanes[i], max_steps);
        for (auto &t : th) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    r.c.seconds += std::chrono::duration<double>(t1 - t0).count();
    for (Lane &L : e->lanes)
        if (L.rc) return fail(e, L.rc, "lane %d: %s", L.index, L.err.c_str());
    return AZ_OK;
}

static int active_slots(const az_engine *e)
{
    int a = 0;
    for (const Lane &L : e->lanes) a += L.active;
    return a;
}

static int read_counters(az_engine *e, az_counters &c)
{
    c.expansions = c.simulations = c.terminal_hits = c.depth_sum = 0;
    c.steps = c.trunk_launches = c.plies = 0;
    c.duplicate_leaves = c.cache_lookups = c.cache_hits = 0;
    int64_t reused = 0;
    double trunk_ms = 0.0, nn_ms = 0.0, step_ms = 0.0, tape_wait = 0.0;
    for (Lane &L : e->lanes) {
        std::vector<unsigned long long> hc((size_t)L.d.B * CNT_STRIDE);
        HIPCHECK(e, az_memcpy(L.stream, hc.data(), L.cnt.p, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int b = 0; b < L.d.B; b++) {
            c.expansions += (int64_t)hc[(size_t)b * CNT_STRIDE + 0];
            c.simulations += (int64_t)hc[(size_t)b * CNT_STRIDE + 1];
            c.terminal_hits += (int64_t)hc[(size_t)b * CNT_STRIDE + 2];
            c.depth_sum += (int64_t)hc[(size_t)b * CNT_STRIDE + 3];
            reused += (int64_t)hc[(size_t)b * CNT_STRIDE + 4];
            c.duplicate_leaves += (int64_t)hc[(size_t)b * CNT_STRIDE + 5];
            c.cache_lookups += (int64_t)hc[(size_t)b * CNT_STRIDE + 6];
            c.cache_hits += (int64_t)hc[(size_t)b * CNT_STRIDE + 7];
        }
        c.steps += L.steps; c.trunk_launches += L.trunk_launches; c.plies += L.plies;
        trunk_ms += L.trunk_ms; nn_ms += L.nn_ms; step_ms += L.step_ms;
        if (L.tape_wait_s > tape_wait) tape_wait = L.tape_wait_s;
    }
    c.tape_wait_seconds = tape_wait;
    c.tape_threads = e->tapes || e->stream_tapes ? e->tape_threads : 0;
    c.host_cpus = e->host_cpus;
    e->run.reused_roots = reused;
    c.trunk_seconds = trunk_ms * 1e-3;
    c.nn_seconds = nn_ms * 1e-3;
    c.step_seconds = step_ms * 1e-3;
    c.root_evals = c.plies - reused;      // a retained root (subtree reuse) is not evaluated again
    c.records = c.plies;
    c.trunk_boards = c.expansions + c.root_evals - c.cache_hits;     // evaluations the net kernels actually ran
    c.games = e->run.num_games;
    return AZ_OK;
}

static int episode_end(az_engine *e, az_counters *out)
{
    az_engine::Run &r = e->run;
    if (!r.open) return fail(e, AZ_ERR_STATE, "no episode is open");
    int rc = read_counters(e, r.c);
    if (rc) return rc;
    e->h_nply.assign(r.num_games, 0);
    e->h_result.assign(r.num_games, 0);
    HIPCHECK(e, az_memcpy(e->stream, e->h_nply.data(), e->g_nply.p, (size_t)r.num_games * 4, hipMemcpyDeviceToHost));
    HIPCHECK(e, az_memcpy(e->stream, e->h_result.data(), e->g_result.p, (size_t)r.num_games * 4, hipMemcpyDeviceToHost));
    if (r.preset) {
        e->h_nply[0] = 1;   // az_search plays exactly one ply; the game itself is not finished by it
    } else {
        // games still in flight when the caller stops early: report the plies played so far
        for (Lane &L : e->lanes) {
            const int B = L.d.B;
            std::vector<int> sg(B), sp(B), ss(B);
            HIPCHECK(e, az_memcpy(L.stream, sg.data(), L.s_game.p, sg.size() * 4, hipMemcpyDeviceToHost));
            HIPCHECK(e, az_memcpy(L.stream, sp.data(), L.s_ply.p, sp.size() * 4, hipMemcpyDeviceToHost));
            HIPCHECK(e, az_memcpy(L.stream, ss.data(), L.s_status.p, ss.size() * 4, hipMemcpyDeviceToHost));
            for (int b = 0; b < B; b++)
                if (ss[b] == SLOT_ACTIVE && sg[b] >= 0 && sg[b] < r.num_games) e->h_nply[sg[b]] = sp[b];
        }
    }
    int64_t plies = 0;
    for (int g = 0; g < r.num_games; g++) plies += e->h_nply[g];
    r.c.plies = r.c.records = plies;
    r.c.root_evals = plies - r.reused_roots;
    r.c.trunk_boards = r.c.expansions + r.c.root_evals - r.c.cache_hits;
    e->last = r.c;
    e->episode_games = r.num_games;
    e->have_episode = true;
    r.open = false;
    stop_tapes(e);
    if (out) *out = r.c;
    return AZ_OK;
}

static int run_episode(az_engine *e, const EpisodeSpec &sp, az_counters *out)
{
    int rc = episode_begin(e, sp);
    if (!rc) rc = episode_plies(e, 1 << 30);
    if (!rc) rc = episode_end(e, out);
    if (rc) { e->run.open = false; stop_tapes(e); }
    return rc;
}

static int upload_T(az_engine *e, const double *table, bool arena)
{
    const int nn = e->nn;
    std::vector<double> T(nn + 4);
    for (int m = 0; m < nn + 4; m++) {
        if (table && m <= nn) T[m] = table[m];
        else T[m] = arena ? 0.3 * std::exp(-(double)m / 4.0) : (std::exp(-(double)m / 100.0) + 0.01) / 1.01;
    }
    HIPCHECK(e, hipMemcpyAsync(e->T_table.p, T.data(), T.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    return AZ_OK;
}

extern "C" int az_selfplay_begin(az_engine *e, const az_selfplay_args *a)
{
    if (!e || !a || a->num_games < 1) return fail(e, AZ_ERR_INVALID, "az_selfplay: bad argument");
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_selfplay_begin: an episode is already open (az_selfplay_end first)");
    DEVICE_GUARD(e);
    const int nn = e->nn, G = a->num_games;
    int rc = ensure_episode_buffers(e, G, true);
    if (rc) return rc;
    if ((rc = upload_T(e, a->temperature_table, false))) return rc;
    // tapes: explicit, or numpy-compatible RandomState(seed0 + g) streams generated on the host cores
    const int plies = a->max_plies > 0 && a->max_plies < nn ? a->max_plies : nn;
    if (a->noise_tape && a->u_tape) {
        // the explicit tape must cover every ply the games can reach: off(plies) = sum_{m < plies} (nn - m) doubles per game
        int64_t need = 0;
        for (int m = 0; m < plies; m++) need += nn - m;
        if (a->tape_stride < need) return fail(e, AZ_ERR_INVALID, "tape_stride %lld is shorter than the %lld doubles %d plies need", (long long)a->tape_stride, (long long)need, plies);
        HIPCHECK(e, hipMemcpy2DAsync(e->noise.p, (size_t)e->tape_len * 8, a->noise_tape, (size_t)a->tape_stride * 8,
                                     (size_t)std::min<int64_t>(a->tape_stride, e->tape_len) * 8, G, hipMemcpyHostToDevice, e->stream));
        HIPCHECK(e, hipMemcpyAsync(e->u.p, a->u_tape, (size_t)G * nn * 8, hipMemcpyHostToDevice, e->stream));
        HIPCHECK(e, hipStreamSynchronize(e->stream));
    } else if (e->stream_tapes) {
        e->tapes = new TapeProducer();
        hipError_t trc = e->tapes->start(e->cfg.device, a->seed0, G, nn, plies, e->cfg.dirichlet_alpha, e->tape_len,
                                         (double *)e->noise.p, (double *)e->u.p, e->tape_threads);
        if (trc != hipSuccess) { stop_tapes(e); return fail(e, AZ_ERR_HIP, "tape producer: %s", hipGetErrorString(trc)); }
    } else {
        const int chunk = 256;
        std::vector<double> hn((size_t)chunk * e->tape_len), hu((size_t)chunk * nn);
        for (int g0 = 0; g0 < G; g0 += chunk) {
            int cnt = std::min(chunk, G - g0);
            azrng::selfplay_tapes_parallel(a->seed0, g0, cnt, nn, e->cfg.dirichlet_alpha, plies, hn.data(), e->tape_len,
                                           hu.data(), host_threads());
            HIPCHECK(e, az_memcpy(e->stream, (double *)e->noise.p + (size_t)g0 * e->tape_len, hn.data(), (size_t)cnt * e->tape_len * 8,
                                  hipMemcpyHostToDevice));
            HIPCHECK(e, az_memcpy(e->stream, (double *)e->u.p + (size_t)g0 * nn, hu.data(), (size_t)cnt * nn * 8, hipMemcpyHostToDevice));
        }
    }
    EpisodeSpec sp;
    sp.num_games = G; sp.max_plies = a->max_plies; sp.add_noise = true; sp.arena = false;
    sp.game_key0 = (unsigned)a->seed0;
    return episode_begin(e, sp);
}

extern "C" int az_selfplay_step(az_engine *e, int max_steps, int32_t *active_out, az_counters *progress)
{
    if (!e || max_steps < 0) return fail(e, AZ_ERR_INVALID, "az_selfplay_step: bad argument");
    DEVICE_GUARD(e);
    int rc = episode_plies(e, max_steps);
    if (rc) return rc;
    if (active_out) *active_out = active_slots(e);
    if (progress) {
        az_counters c = e->run.c;
        if ((rc = read_counters(e, c))) return rc;
        *progress = c;
    }
    return AZ_OK;
}

extern "C" int az_selfplay_end(az_engine *e, az_counters *out)
{
    if (!e) return AZ_ERR_INVALID;
    DEVICE_GUARD(e);
    const int rc = episode_end(e, out);
    if (rc) { e->run.open = false; stop_tapes(e); }      // a failed episode is closed too: the engine stays usable
    return rc;
}

extern "C" int az_selfplay(az_engine *e, const az_selfplay_args *a, az_counters *out)
{
    if (!e) return AZ_ERR_INVALID;
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_selfplay: an episode is already open (az_selfplay_end first)");
    DEVICE_GUARD(e);
    int rc = az_selfplay_begin(e, a);
    if (!rc) rc = episode_plies(e, 1 << 30);
    if (!rc) rc = episode_end(e, out);
    if (rc) { e->run.open = false; stop_tapes(e); }
    return rc;
}

extern "C" int az_selfplay_games(az_engine *e, int32_t *nply, int32_t *result)
{
    if (!e || !e->have_episode) return fail(e, AZ_ERR_STATE, "no episode has been run");
    for (int g = 0; g < e->episode_games; g++) {
        if (nply) nply[g] = e->h_nply[g];
        if (result) result[g] = e->h_result[g];
    }
    return AZ_OK;
}

extern "C" int az_selfplay_records(az_engine *e, uint8_t *boards, uint8_t *movers, int16_t *lasts, int16_t *actions,
                                   float *pis, int32_t *visits, int8_t *z)
{
    if (!e || !e->have_episode) return fail(e, AZ_ERR_STATE, "no episode has been run");
    DEVICE_GUARD(e);
    const int nn = e->nn, G = e->episode_games;
    const size_t tot = (size_t)G * nn;
    std::vector<u64> pl(tot * 8);
    std::vector<short> la(tot), ac(tot);
    std::vector<unsigned char> mv(tot);
    std::vector<float> pi(tot * nn);
    std::vector<unsigned short> vis(tot * nn);
    HIPCHECK(e, az_memcpy(e->stream, pl.data(), e->rec_planes.p, pl.size() * 8, hipMemcpyDeviceToHost));
    HIPCHECK(e, az_memcpy(e->stream, la.data(), e->rec_last.p, la.size() * 2, hipMemcpyDeviceToHost));
    HIPCHECK(e, az_memcpy(e->stream, ac.data(), e->rec_action.p, ac.size() * 2, hipMemcpyDeviceToHost));
    HIPCHECK(e, az_memcpy(e->stream, mv.data(), e->rec_mover.p, mv.size(), hipMemcpyDeviceToHost));
    HIPCHECK(e, az_memcpy(e->stream, pi.data(), e->rec_pi.p, pi.size() * 4, hipMemcpyDeviceToHost));
    HIPCHECK(e, az_memcpy(e->stream, vis.data(), e->rec_visits.p, vis.size() * 2, hipMemcpyDeviceToHost));
    size_t r = 0;
    for (int g = 0; g < G; g++) {
        for (int m = 0; m < e->h_nply[g]; m++, r++) {
            size_t si = (size_t)g * nn + m;
            int mover = mv[si];
            if (boards)
                for (int j = 0; j < nn; j++) {
                    bool me = (pl[si * 8 + (j >> 6)] >> (j & 63)) & 1ull, op = (pl[si * 8 + 4 + (j >> 6)] >> (j & 63)) & 1ull;
                    boards[r * nn + j] = (uint8_t)(me ? mover : (op ? 3 - mover : 0));
                }
            if (movers) movers[r] = (uint8_t)mover;
            if (lasts) lasts[r] = la[si];
            if (actions) actions[r] = ac[si];
            if (pis) memcpy(pis + r * nn, pi.data() + si * nn, (size_t)nn * 4);
            if (visits) for (int j = 0; j < nn; j++) visits[r * nn + j] = vis[si * nn + j];
            int res = e->h_result[g];
            if (z) z[r] = (int8_t)(res == 0 ? 99 : (res == 3 ? 0 : (mover == res ? 1 : -1)));
        }
    }
    return AZ_OK;
}

extern "C" int64_t az_record_bytes(const az_engine *e) { return e ? record_bytes(e->nn) : 0; }

extern "C" int az_selfplay_pack(az_engine *e, void *packed_dev)
{
    if (!e || !e->have_episode || !packed_dev) return fail(e, AZ_ERR_STATE, "az_selfplay_pack: no episode / null buffer");
    DEVICE_GUARD(e);
    const int nn = e->nn;
    std::vector<int> src;
    for (int g = 0; g < e->episode_games; g++)
        for (int m = 0; m < e->h_nply[g]; m++) src.push_back(g * nn + m);
    if (src.empty()) return AZ_OK;
    int rc = upload(e, e->src_index, src.data(), src.size() * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)src.size()), dim3(64), 0, e->stream, e->lanes[0].d, (const int *)e->src_index.p,
                       (int64_t)src.size(), nn, record_bytes(nn), (unsigned char *)packed_dev);
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    HIPCHECK(e, hipGetLastError());
    return AZ_OK;
}

extern "C" int az_examples_from_packed(az_engine *e, const void *packed_dev, int64_t records, int aug, float *states_dev,
                                       float *pis_dev, float *z_dev)
{
    if (!e || !packed_dev || !states_dev || !pis_dev || !z_dev || records < 0) return fail(e, AZ_ERR_INVALID, "az_examples_from_packed: bad argument");
    if (aug != AZ_AUG_NONE && aug != AZ_AUG_REFERENCE4 && aug != AZ_AUG_DIHEDRAL8) return fail(e, AZ_ERR_INVALID, "aug must be 1, 4 or 8");
    if (records == 0) return AZ_OK;
    DEVICE_GUARD(e);
    hipLaunchKernelGGL(k_examples, dim3((unsigned)records), dim3(256), 0, e->stream, (const unsigned char *)packed_dev,
                       records, e->n, record_bytes(e->nn), aug, states_dev, pis_dev, z_dev);
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    HIPCHECK(e, hipGetLastError());
    return AZ_OK;
}

// ------------------------------------------------------------------------------------------------
// single-position entry points
// ------------------------------------------------------------------------------------------------
static void planes_from_cells(const uint8_t *cells, int nn, u64 *x, u64 *o)
{
    for (int q = 0; q < 4; q++) x[q] = o[q] = 0;
    for (int j = 0; j < nn; j++) {
        if (cells[j] == 1) x[j >> 6] |= 1ull << (j & 63);
        else if (cells[j] == 2) o[j >> 6] |= 1ull << (j & 63);
    }
}

extern "C" int az_net_eval(az_engine *e, int slot, int count, const uint8_t *boards, const uint8_t *players,
                           const int16_t *lasts, float *logits, float *policy, float *value)
{
    if (!e || slot < 0 || slot > 1 || count < 0 || !boards || !players || !lasts) return fail(e, AZ_ERR_INVALID, "az_net_eval: bad argument");
    if (!e->net[slot].loaded) return fail(e, AZ_ERR_NO_WEIGHTS, "weights slot %d not loaded", slot);
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_net_eval: a self-play episode is open on this engine");
    DEVICE_GUARD(e);
    Lane &L = e->lanes[0];            // single-position and batch evaluation calls use lane 0
    const int nn = e->nn, B = L.dv.B;              // boards per pass = evaluation items of the lane
    DevBuf dpol, dval;
    int rc = dev_alloc(e, dpol, (size_t)B * nn * 4);
    if (!rc) rc = dev_alloc(e, dval, (size_t)B * 4);
    for (int c0 = 0; !rc && c0 < count; c0 += B) {
        const int cnt = std::min(B, count - c0);
        std::vector<u64> lf((size_t)B * 8, 0);
        std::vector<int> kind(B, LEAF_NONE), st(B, SLOT_IDLE), nets(B, slot), ll(B, -1);
        for (int i = 0; i < cnt; i++) {
            u64 x[4], o[4];
            planes_from_cells(boards + (size_t)(c0 + i) * nn, nn, x, o);
            const bool xm = players[c0 + i] == 1;
            for (int q = 0; q < 4; q++) { lf[(size_t)i * 8 + q] = xm ? x[q] : o[q]; lf[(size_t)i * 8 + 4 + q] = xm ? o[q] : x[q]; }
            kind[i] = LEAF_ROOT; st[i] = SLOT_ACTIVE; ll[i] = lasts[c0 + i];
        }
        hipError_t hr = az_memcpy(e->stream, L.leaf.p, lf.data(), lf.size() * 8, hipMemcpyHostToDevice);
        if (hr == hipSuccess) hr = az_memcpy(e->stream, L.leaf_kind.p, kind.data(), (size_t)B * 4, hipMemcpyHostToDevice);
        if (hr == hipSuccess) hr = az_memcpy(e->stream, L.dv.s_status, st.data(), (size_t)B * 4, hipMemcpyHostToDevice);
        if (hr == hipSuccess) hr = az_memcpy(e->stream, L.dv.s_net, nets.data(), (size_t)B * 4, hipMemcpyHostToDevice);
        if (hr == hipSuccess) hr = az_memcpy(e->stream, L.leaf_last.p, ll.data(), (size_t)B * 4, hipMemcpyHostToDevice);
        if (hr != hipSuccess) { rc = fail(e, AZ_ERR_HIP, "az_net_eval upload: %s", hipGetErrorString(hr)); break; }
        {
            LaunchCtx lc = ctx_of_impl(e, L);
            lc.d.leaf_sym = lc.dv.leaf_sym = nullptr;       // controller.py:39-53 evaluates the position as it is
            if (e->split_max > 0 && L.scratch.p && cnt <= e->split_max && e->trunk_mode == AZ_TRUNK_F32) e->ops->trunk_split(lc, slot); else e->ops->trunk(lc, slot);
            e->ops->fc(lc, slot);
            e->ops->eval_tail(lc, cnt, (float *)dpol.p, (float *)dval.p);
        }
        hr = hipStreamSynchronize(e->stream);
        if (hr == hipSuccess) hr = hipGetLastError();
        if (hr == hipSuccess && logits)
            hr = az_memcpy2d(e->stream, logits + (size_t)c0 * nn, (size_t)nn * 4, L.logits.p, (size_t)e->RW * 4, (size_t)nn * 4, cnt, hipMemcpyDeviceToHost);
        if (hr == hipSuccess && policy) hr = az_memcpy(e->stream, policy + (size_t)c0 * nn, dpol.p, (size_t)cnt * nn * 4, hipMemcpyDeviceToHost);
        if (hr == hipSuccess && value) hr = az_memcpy(e->stream, value + c0, dval.p, (size_t)cnt * 4, hipMemcpyDeviceToHost);
        if (hr != hipSuccess) { rc = fail(e, AZ_ERR_HIP, "az_net_eval: %s", hipGetErrorString(hr)); break; }
    }
    // leave the slots idle
    (void)hipMemsetAsync(L.dv.s_status, 0, (size_t)B * 4, e->stream);
    (void)hipMemsetAsync(L.leaf_kind.p, 0, L.leaf_kind.bytes, e->stream);
    (void)hipStreamSynchronize(e->stream);
    dev_free(dpol);
    dev_free(dval);
    return rc;
}

extern "C" int az_search(az_engine *e, int slot, const uint8_t *board, int player, int last, double temperature,
                         const double *noise, double u, float *pi, int32_t *action, int32_t *visits, double *W,
                         float *prior)
{
    if (!e || !board || (player != 1 && player != 2) || slot < 0 || slot > 1) return fail(e, AZ_ERR_INVALID, "az_search: bad argument");
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_search: a self-play episode is open on this engine");
    DEVICE_GUARD(e);
    const int nn = e->nn;
    int stones = 0;
    for (int j = 0; j < nn; j++) {
        if (board[j] > 2) return fail(e, AZ_ERR_INVALID, "az_search: cell value %d", board[j]);
        stones += board[j] != 0;
    }
    if (stones >= nn) return fail(e, AZ_ERR_INVALID, "az_search: no legal action");
    if (last >= nn || (last >= 0 && board[last] == 0)) return fail(e, AZ_ERR_INVALID, "az_search: bad last action");
    int rc = ensure_episode_buffers(e, 1, true);
    if (rc) return rc;
    std::vector<double> T(nn + 1, temperature);
    if ((rc = upload_T(e, T.data(), false))) return rc;
    // tape: the single ply `stones`
    std::vector<double> hn((size_t)e->tape_len, 0.0), hu(nn, u);
    int off = 0;
    for (int m = 0; m < stones; m++) off += nn - m;
    if (noise) memcpy(hn.data() + off, noise, (size_t)(nn - stones) * 8);
    HIPCHECK(e, az_memcpy(e->stream, e->noise.p, hn.data(), hn.size() * 8, hipMemcpyHostToDevice));
    HIPCHECK(e, az_memcpy(e->stream, e->u.p, hu.data(), hu.size() * 8, hipMemcpyHostToDevice));
    u64 bd[8];
    planes_from_cells(board, nn, bd, bd + 4);
    Lane &L0 = e->lanes[0];
    HIPCHECK(e, hipMemsetAsync(L0.s_status.p, 0, L0.s_status.bytes, e->stream));
    HIPCHECK(e, hipMemcpyAsync(L0.board.p, bd, sizeof bd, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_set_position, dim3(1), dim3(64), 0, e->stream, L0.d, 0, 0, player, last, stones);
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    EpisodeSpec sp;
    sp.num_games = 1; sp.max_plies = 0; sp.add_noise = noise != nullptr; sp.arena = false; sp.preset = true; sp.profile = false;
    // the arena flag only selects the net through s_player; a search with the baseline net uses slot 1 weights as slot 0
    PackedNet saved0 = e->net[0];
    const float *sv2w = L0.d.v2w[0], *sv2b = L0.d.v2b[0];
    // (the evaluation cache keys its entries by net id: the swapped search runs under its own cache generation)
    if (slot == 1) { e->net[0] = e->net[1]; L0.d.v2w[0] = L0.d.v2w[1]; L0.d.v2b[0] = L0.d.v2b[1]; e->cache_gen ^= 0x80000000u; }
    rc = run_episode(e, sp, nullptr);
    if (slot == 1) { e->net[0] = saved0; L0.d.v2w[0] = sv2w; L0.d.v2b[0] = sv2b; e->cache_gen ^= 0x80000000u; }
    if (rc) return rc;
    // outputs: record 0*nn + stones
    const size_t ri = (size_t)stones;
    if (pi) HIPCHECK(e, az_memcpy(e->stream, pi, (float *)e->rec_pi.p + ri * nn, (size_t)nn * 4, hipMemcpyDeviceToHost));
    if (action) {
        short a = -1;
        HIPCHECK(e, az_memcpy(e->stream, &a, (short *)e->rec_action.p + ri, 2, hipMemcpyDeviceToHost));
        *action = a;
    }
    if (visits || W || prior) {
        std::vector<Edge> row(e->RW);
        HIPCHECK(e, az_memcpy(e->stream, row.data(), L0.edges.p, row.size() * sizeof(Edge), hipMemcpyDeviceToHost));
        for (int j = 0; j < nn; j++) {
            const bool legal = board[j] == 0;
            if (visits) visits[j] = legal ? row[j].N : 0;
            if (W) W[j] = legal ? row[j].W : 0.0;
            if (prior) prior[j] = legal ? row[j].P : 0.0f;
        }
    }
    e->have_episode = false;
    return AZ_OK;
}

extern "C" int az_arena(az_engine *e, const az_arena_args *a, az_arena_result *out, int32_t *results, int16_t *actions,
                        int32_t *nply)
{
    if (!e || !a || a->num_games < 1) return fail(e, AZ_ERR_INVALID, "az_arena: bad argument");
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_arena: a self-play episode is open on this engine");
    DEVICE_GUARD(e);
    const int nn = e->nn, G = a->num_games;
    int rc = ensure_episode_buffers(e, G, false);
    if (rc) return rc;
    if ((rc = upload_T(e, a->temperature_table, true))) return rc;
    std::vector<double> hu((size_t)G * nn);
    if (a->u_tape) memcpy(hu.data(), a->u_tape, hu.size() * 8);
    else for (int g = 0; g < G; g++) azrng::uniforms(a->seed0 + (uint64_t)g, nn, hu.data() + (size_t)g * nn);
    HIPCHECK(e, az_memcpy(e->stream, e->u.p, hu.data(), hu.size() * 8, hipMemcpyHostToDevice));
    EpisodeSpec sp;
    sp.num_games = G; sp.max_plies = 0; sp.add_noise = false; sp.arena = true;
    sp.game_key0 = (unsigned)a->seed0;
    az_counters c;
    if ((rc = run_episode(e, sp, &c))) return rc;
    int w = 0, l = 0, dr = 0;
    for (int g = 0; g < G; g++) {
        int r = e->h_result[g];
        w += r == AZ_RES_X; l += r == AZ_RES_O; dr += r == AZ_RES_DRAW;   // candidate = X, baseline = O (evaluator.py:64)
        if (results) results[g] = r;
        if (nply) nply[g] = e->h_nply[g];
    }
    if (actions) {
        std::vector<short> ac((size_t)G * nn);
        HIPCHECK(e, az_memcpy(e->stream, ac.data(), e->rec_action.p, ac.size() * 2, hipMemcpyDeviceToHost));
        for (int g = 0; g < G; g++)
            for (int m = 0; m < nn; m++) actions[(size_t)g * nn + m] = m < e->h_nply[g] ? ac[(size_t)g * nn + m] : (int16_t)-1;
    }
    if (out) {
        out->wins = w; out->losses = l; out->draws = dr; out->total = w + l + dr;
        out->win_rate = out->total ? (w + 0.5 * dr) / out->total : 0.0;     // evaluator.py:106-109
    }
    return AZ_OK;
}

extern "C" int az_rules_replay(az_engine *e, int games, int max_len, const int16_t *actions, uint8_t *term_before,
                               uint8_t *boards, int32_t *players, int32_t *results, int32_t *first_illegal)
{
    if (!e || games < 0 || max_len < 0 || !actions || !boards || !players || !results || !first_illegal)
        return fail(e, AZ_ERR_INVALID, "az_rules_replay: bad argument");
    if (games == 0) return AZ_OK;
    DEVICE_GUARD(e);
    const size_t G = (size_t)games, L = (size_t)(max_len > 0 ? max_len : 1), nn = (size_t)e->nn;
    DevBuf da, dt, db, dp, dr, ds;
    int rc = dev_alloc(e, da, G * L * 2, true);
    if (!rc) rc = dev_alloc(e, dt, G * L, true);
    if (!rc) rc = dev_alloc(e, db, G * nn, false);
    if (!rc) rc = dev_alloc(e, dp, G * 4, false);
    if (!rc) rc = dev_alloc(e, dr, G * 4, false);
    if (!rc) rc = dev_alloc(e, ds, G * 4, false);
    hipError_t hr = hipSuccess;
    if (!rc) {
        if (max_len > 0) hr = az_memcpy(e->stream, da.p, actions, G * L * 2, hipMemcpyHostToDevice);
        else hr = hipMemsetAsync(da.p, 0xFF, G * L * 2, e->stream);
        if (hr == hipSuccess) {
            hipLaunchKernelGGL(k_rules_replay, dim3((games + 63) / 64), dim3(64), 0, e->stream, e->n, e->cfg.win_length, games,
                               (int)L, (const short *)da.p, term_before ? (unsigned char *)dt.p : nullptr, (unsigned char *)db.p,
                               (int *)dp.p, (int *)dr.p, (int *)ds.p);
            hr = hipStreamSynchronize(e->stream);
        }
        if (hr == hipSuccess) hr = hipGetLastError();
        if (hr == hipSuccess && term_before && max_len > 0) hr = az_memcpy(e->stream, term_before, dt.p, G * L, hipMemcpyDeviceToHost);
        if (hr == hipSuccess) hr = az_memcpy(e->stream, boards, db.p, G * nn, hipMemcpyDeviceToHost);
        if (hr == hipSuccess) hr = az_memcpy(e->stream, players, dp.p, G * 4, hipMemcpyDeviceToHost);
        if (hr == hipSuccess) hr = az_memcpy(e->stream, results, dr.p, G * 4, hipMemcpyDeviceToHost);
        if (hr == hipSuccess) hr = az_memcpy(e->stream, first_illegal, ds.p, G * 4, hipMemcpyDeviceToHost);
        if (hr != hipSuccess) rc = fail(e, AZ_ERR_HIP, "az_rules_replay: %s", hipGetErrorString(hr));
    }
    dev_free(da); dev_free(dt); dev_free(db); dev_free(dp); dev_free(dr); dev_free(ds);
    return rc;
}

// diagnostic builds (-DAZ_STAMPS): per-workgroup phase stamps of the last k_trunk launch, 16 u64 per workgroup
extern "C" int az_debug_stamps(az_engine *e, unsigned long long *out, int max_groups)
{
    if (!e || !out || !e->lanes[0].dbg.p) return AZ_ERR_STATE;
    DevBuf &dbg = e->lanes[0].dbg;
    size_t n = std::min<size_t>((size_t)max_groups * 16, dbg.bytes / 8);
    if (max_groups < 0) n = dbg.bytes / 8;   // everything (trunk stamps, then k_fc stamps at offset B*16)
    HIPCHECK(e, az_memcpy(e->stream, out, dbg.p, n * 8, hipMemcpyDeviceToHost));
    return AZ_OK;
}

extern "C" int az_examples_gather(az_engine *e, const void *packed_dev, const int64_t *idx_dev, const int32_t *sym_dev,
                                  int count, int reference_pi, float *states_dev, float *pis_dev, float *z_dev)
{
    if (!e || !packed_dev || !idx_dev || !sym_dev || !states_dev || !pis_dev || !z_dev || count < 0)
        return fail(e, AZ_ERR_INVALID, "az_examples_gather: bad argument");
    if (count == 0) return AZ_OK;
    DEVICE_GUARD(e);
    hipLaunchKernelGGL(k_examples_gather, dim3((unsigned)count), dim3(256), 0, e->stream, (const unsigned char *)packed_dev,
                       (const long long *)idx_dev, (const int *)sym_dev, count, e->n, record_bytes(e->nn), reference_pi,
                       states_dev, pis_dev, z_dev);
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    HIPCHECK(e, hipGetLastError());
    return AZ_OK;
}

extern "C" int az_set_subtree_reuse(az_engine *e, int on)
{
    if (!e) return AZ_ERR_INVALID;
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_set_subtree_reuse: an episode is open");
    if (on && e->vl > 1) return fail(e, AZ_ERR_INVALID, "subtree reuse and virtual-loss batching cannot be combined");
    if (on && e->R > REUSE_MAX_ROWS)
        return fail(e, AZ_ERR_INVALID, "subtree reuse supports at most %d simulations per move", REUSE_MAX_ROWS - 1);
    e->reuse = on ? 1 : 0;
    return AZ_OK;
}

extern "C" int az_set_profiling(az_engine *e, int on)
{
    if (!e) return AZ_ERR_INVALID;
    e->profile = on != 0;
    return AZ_OK;
}

extern "C" int az_get_counters(const az_engine *e, az_counters *out)
{
    if (!e || !out) return AZ_ERR_INVALID;
    *out = e->last;
    return AZ_OK;
}

extern "C" int az_get_lanes(const az_engine *e) { return e ? (int)e->lanes.size() : AZ_ERR_INVALID; }

extern "C" int az_set_virtual_loss(az_engine *e, int leaves)
{
    if (!e) return AZ_ERR_INVALID;
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_set_virtual_loss: an episode is open");
    if (leaves < 1 || leaves > VL_MAX) return fail(e, AZ_ERR_INVALID, "virtual-loss batching supports 1..%d leaves per batch", VL_MAX);
    if (leaves > 1 && e->reuse) return fail(e, AZ_ERR_INVALID, "subtree reuse and virtual-loss batching cannot be combined");
    DEVICE_GUARD(e);
    if (leaves != e->vl) {
        for (Lane &L : e->lanes) {
            int rc = alloc_items(e, L, leaves);
            if (rc) return rc;
            for (int i = 0; i < 4; i++)       // the captured plies bake the old buffers in
                if (L.graph[i >> 1][i & 1].exec) { (void)hipGraphExecDestroy(L.graph[i >> 1][i & 1].exec); L.graph[i >> 1][i & 1].exec = nullptr; }
        }
        HIPCHECK(e, hipStreamSynchronize(e->stream));
    }
    e->vl = leaves;
    const char *f = getenv("AZ_VL_FORCE");       // AZ_VL_FORCE=1: the batched kernel also for batches of one (must equal k_step)
    e->vl_kernel = leaves > 1 || (f && f[0] == '1');
    each_state(e, [&](DevState &d) { (void)d; });        // refresh the item views
    return AZ_OK;
}

extern "C" int az_set_eval_cache(az_engine *e, int64_t entries)
{
    if (!e) return AZ_ERR_INVALID;
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_set_eval_cache: an episode is open");
    if (entries < 0 || entries > ((int64_t)1 << 26)) return fail(e, AZ_ERR_INVALID, "cache entries must be 0 (off) .. 2^26");
    DEVICE_GUARD(e);
    if (entries == 0) {
        dev_free(e->cache);
        e->cache_mask = 0;
    } else {
        int64_t n = 1024;
        while (n < entries) n <<= 1;
        const size_t ces = (size_t)(32 + e->RW + 64) * sizeof(float);       // CacheGeo<n>::CES floats per entry
        dev_free(e->cache);
        int rc = dev_alloc(e, e->cache, (size_t)n * ces, true);
        if (rc) return rc;
        HIPCHECK(e, hipStreamSynchronize(e->stream));
        e->cache_mask = (unsigned)(n - 1);
    }
    e->cache_gen++;
    each_state(e, [&](DevState &d) { d.cache = (float *)e->cache.p; d.cache_mask = e->cache_mask; d.cache_gen = e->cache_gen; });
    return AZ_OK;
}

extern "C" int az_set_leaf_symmetry(az_engine *e, int on)
{
    if (!e) return AZ_ERR_INVALID;
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_set_leaf_symmetry: an episode is open");
    if (on && e->cfg.eval_kind != AZ_EVAL_NET) return fail(e, AZ_ERR_INVALID, "random-symmetry leaf evaluation needs the net evaluator");
    e->leaf_symmetry = on ? 1 : 0;
    for (Lane &L : e->lanes) L.d.leaf_sym = on ? (int *)L.leaf_sym.p : nullptr;
    each_state(e, [&](DevState &d) { (void)d; });        // refresh the item views
    return AZ_OK;
}

extern "C" int az_set_trunk_mode(az_engine *e, int mode)
{
    if (!e) return AZ_ERR_INVALID;
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_set_trunk_mode: an episode is open");
    if (mode != AZ_TRUNK_F32 && mode != AZ_TRUNK_BF16X3 && mode != AZ_TRUNK_F16X2) return fail(e, AZ_ERR_INVALID, "az_set_trunk_mode: unknown mode %d", mode);
    if (mode == AZ_TRUNK_F16X2)
        for (int i = 0; i < 2; i++)
            if (e->net[i].loaded && !e->net[i].f16_ok)
                return fail(e, AZ_ERR_INVALID, "az_set_trunk_mode: weight slot %d holds a value outside float16's range (|w| >= 65504): use AZ_TRUNK_BF16X3", i);
    if (mode != AZ_TRUNK_F32) {
        DEVICE_GUARD(e);
        for (int i = 0; i < 2; i++) {
            const int rc = ensure_emul(e, i, mode);
            if (rc) return rc;
        }
    }
    if (mode != e->trunk_mode) {
        e->trunk_mode = mode;
        e->cache_gen++;           // cached evaluations of the other arithmetic never match again
        each_state(e, [&](DevState &d) { d.cache_gen = e->cache_gen; });
    }
    return AZ_OK;
}

extern "C" int az_get_trunk_mode(const az_engine *e) { return e ? e->trunk_mode : AZ_ERR_INVALID; }

extern "C" int az_emul_split(int mode, float x, uint16_t *parts)
{
    if (!parts || (mode != AZ_TRUNK_BF16X3 && mode != AZ_TRUNK_F16X2)) return AZ_ERR_INVALID;
    return emul_split(mode, x, parts) ? emul_parts(mode) : AZ_ERR_INVALID;
}

extern "C" int az_get_persistent(const az_engine *e) { return e ? e->persist_gp : AZ_ERR_INVALID; }

// MCTS.run with the evaluator outside the engine: the policy_value_fn seam of mcts.py:87-93 for evaluators that are not
// this engine's net (any callable in the reference).  The tree kernels run on the GPU as always; each of the
// num_simulations + 1 evaluations crosses to the host: the pending leaf is read back, the callback fills priors and
// value, they are uploaded, and the tree step consumes them as they are (no softmax, no value head).
extern "C" int az_search_callback(az_engine *e, const uint8_t *board, int player, int last, double temperature,
                                  const double *noise, double u, az_eval_callback fn, void *user, float *pi, int32_t *action,
                                  int32_t *visits, double *W, float *prior)
{
    if (!e || !board || !fn || (player != 1 && player != 2)) return fail(e, AZ_ERR_INVALID, "az_search_callback: bad argument");
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_search_callback: a self-play episode is open on this engine");
    if (e->vl > 1 || e->reuse || e->leaf_symmetry)
        return fail(e, AZ_ERR_INVALID, "az_search_callback: not combinable with virtual-loss batching, subtree reuse or random-symmetry leaf evaluation");
    DEVICE_GUARD(e);
    const int nn = e->nn, S = e->cfg.num_simulations;
    int stones = 0;
    for (int j = 0; j < nn; j++) {
        if (board[j] > 2) return fail(e, AZ_ERR_INVALID, "az_search_callback: cell value %d", board[j]);
        stones += board[j] != 0;
    }
    if (stones >= nn) return fail(e, AZ_ERR_INVALID, "az_search_callback: no legal action");
    if (last >= nn || (last >= 0 && board[last] == 0)) return fail(e, AZ_ERR_INVALID, "az_search_callback: bad last action");
    int rc = ensure_episode_buffers(e, 1, true);
    if (rc) return rc;
    std::vector<double> T(nn + 1, temperature);
    if ((rc = upload_T(e, T.data(), false))) return rc;
    std::vector<double> hn((size_t)e->tape_len, 0.0), hu(nn, u);
    int off = 0;
    for (int m = 0; m < stones; m++) off += nn - m;
    if (noise) memcpy(hn.data() + off, noise, (size_t)(nn - stones) * 8);
    HIPCHECK(e, az_memcpy(e->stream, e->noise.p, hn.data(), hn.size() * 8, hipMemcpyHostToDevice));
    HIPCHECK(e, az_memcpy(e->stream, e->u.p, hu.data(), hu.size() * 8, hipMemcpyHostToDevice));
    u64 bd[8];
    planes_from_cells(board, nn, bd, bd + 4);
    Lane &L = e->lanes[0];
    HIPCHECK(e, hipMemsetAsync(L.s_status.p, 0, L.s_status.bytes, e->stream));
    HIPCHECK(e, hipMemcpyAsync(L.board.p, bd, sizeof bd, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_set_position, dim3(1), dim3(64), 0, e->stream, L.d, 0, 0, player, last, stones);
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    // the episode machinery is bypassed: this loop is the episode (one game, one ply), driven step by step from the host
    const bool had_cache = e->cache.p != nullptr;
    HIPCHECK(e, hipMemsetAsync(L.cnt.p, 0, L.cnt.bytes, e->stream));
    HIPCHECK(e, hipMemsetAsync(L.carried.p, 0xFF, L.carried.bytes, e->stream));
    // from here on the lane states say "priors come from outside": put them back on EVERY way out (the guard below),
    // or a later az_selfplay / az_arena would feed raw logits to the tree step as priors
    each_state(e, [&](DevState &d) {
        d.max_plies = 0; d.add_noise = noise ? 1 : 0; d.arena = 0; d.total_games = 1; d.reuse = 0; d.game_key0 = 0;
        d.cache = nullptr; d.ext_eval = 1;
    });
    struct Restore {
        az_engine *e; bool had_cache;
        ~Restore() { each_state(e, [&](DevState &d) { d.ext_eval = 0; d.cache = had_cache ? (float *)e->cache.p : nullptr; }); }
    } restore{e, had_cache};
    e->persist_gp = 0;
    LaunchCtx lc = ctx_of_impl(e, L);
    lc.synthetic = 0;
    hipLaunchKernelGGL(k_begin, dim3((L.d.B + 255) / 256), dim3(256), 0, e->stream, L.d);
    std::vector<float> pol(e->RW, 0.0f), hid(64, 0.0f);
    std::vector<uint8_t> cells(nn);
    int cbrc = 0;
    for (int idx = 0; idx <= S && !rc; idx++) {
        int kind = LEAF_NONE, depth = 0, llast = -1;
        u64 lf[8];
        hipError_t hr = az_memcpy(e->stream, &kind, L.leaf_kind.p, 4, hipMemcpyDeviceToHost);
        if (hr == hipSuccess && leaf_needs_net(kind)) {
            hr = az_memcpy(e->stream, lf, L.leaf.p, sizeof lf, hipMemcpyDeviceToHost);
            if (hr == hipSuccess) hr = az_memcpy(e->stream, &depth, L.depth.p, 4, hipMemcpyDeviceToHost);
            if (hr == hipSuccess) hr = az_memcpy(e->stream, &llast, L.leaf_last.p, 4, hipMemcpyDeviceToHost);
            if (hr == hipSuccess) {
                const int mover = (depth & 1) ? 3 - player : player;     // side to move at the leaf
                for (int j = 0; j < nn; j++) {
                    const bool me = (lf[j >> 6] >> (j & 63)) & 1ull, op = (lf[4 + (j >> 6)] >> (j & 63)) & 1ull;
                    cells[j] = (uint8_t)(me ? mover : (op ? 3 - mover : 0));
                }
                float value = 0.0f;
                std::fill(pol.begin(), pol.end(), 0.0f);
                cbrc = fn(user, cells.data(), mover, llast, pol.data(), &value);
                if (cbrc) { rc = fail(e, AZ_ERR_INVALID, "az_search_callback: the evaluator returned %d", cbrc); break; }
                hid[0] = value;
                hr = az_memcpy(e->stream, L.logits.p, pol.data(), (size_t)e->RW * 4, hipMemcpyHostToDevice);
                if (hr == hipSuccess) hr = az_memcpy(e->stream, L.vhid.p, hid.data(), 64 * 4, hipMemcpyHostToDevice);
            }
        }
        if (hr != hipSuccess) { rc = fail(e, AZ_ERR_HIP, "az_search_callback: %s", hipGetErrorString(hr)); break; }
        e->ops->step(lc, idx, idx < S ? 1 : 0);
    }
    if (!rc) {
        e->ops->move(lc);
        hipError_t hr = hipStreamSynchronize(e->stream);
        if (hr == hipSuccess) hr = hipGetLastError();
        if (hr != hipSuccess) rc = fail(e, AZ_ERR_HIP, "az_search_callback: %s", hipGetErrorString(hr));
    } else {
        (void)hipStreamSynchronize(e->stream);
    }
    (void)hipMemsetAsync(L.s_status.p, 0, L.s_status.bytes, e->stream);
    (void)hipMemsetAsync(L.leaf_kind.p, 0, L.leaf_kind.bytes, e->stream);
    (void)hipStreamSynchronize(e->stream);
    e->have_episode = false;
    if (rc) return rc;
    const size_t ri = (size_t)stones;
    if (pi) HIPCHECK(e, az_memcpy(e->stream, pi, (float *)e->rec_pi.p + ri * nn, (size_t)nn * 4, hipMemcpyDeviceToHost));
    if (action) {
        short a = -1;
        HIPCHECK(e, az_memcpy(e->stream, &a, (short *)e->rec_action.p + ri, 2, hipMemcpyDeviceToHost));
        *action = a;
    }
    if (visits || W || prior) {
        std::vector<Edge> row(e->RW);
        HIPCHECK(e, az_memcpy(e->stream, row.data(), L.edges.p, row.size() * sizeof(Edge), hipMemcpyDeviceToHost));
        for (int j = 0; j < nn; j++) {
            const bool legal = board[j] == 0;
            if (visits) visits[j] = legal ? row[j].N : 0;
            if (W) W[j] = legal ? row[j].W : 0.0;
            if (prior) prior[j] = legal ? row[j].P : 0.0f;
        }
    }
    return AZ_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU: the episode-end exchange on RCCL, inside the library (include/az_engine.h az_dist_*).  librccl.so.1 is bound
// at run time -- the copy a host framework (PyTorch) has already mapped if there is one, so that the process holds ONE
// RCCL, else the system's -- and only the handful of entry points used here are resolved.
// ------------------------------------------------------------------------------------------------
namespace {
typedef int nccl_result;                     // ncclResult_t; ncclSuccess = 0
enum { NCCL_INT8 = 0, NCCL_UINT8 = 1, NCCL_INT64 = 4 };      // ncclDataType_t values used (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt64 4)
enum { NCCL_SUM = 0 };
struct nccl_id { char internal[AZ_DIST_ID_BYTES]; };
struct Rccl {
    void *h = nullptr;
    nccl_result (*GetUniqueId)(nccl_id *) = nullptr;
    nccl_result (*CommInitRank)(void **, int, nccl_id, int) = nullptr;
    nccl_result (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(nccl_result) = nullptr;
    nccl_result (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    nccl_result (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    nccl_result (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    nccl_result (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    nccl_result (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    nccl_result (*GroupStart)() = nullptr;
    nccl_result (*GroupEnd)() = nullptr;
    std::string err;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

const Rccl *rccl_load(std::string *err)
{
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    Rccl &r = g_rccl;
    if (r.h) return &r;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy already in the process (PyTorch's), if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { *err = std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "?"); return nullptr; }
    bool ok = true;
    auto sym = [&](const char *name) { void *p = dlsym(h, name); if (!p) { ok = false; *err = std::string("librccl.so.1 lacks ") + name; } return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    if (!ok) return nullptr;
    r.h = h;
    return &r;
}
}  // namespace

#define NCCLCHECK(e, R, call)                                                                      \
    do {                                                                                           \
        nccl_result _r = (call);                                                                   \
        if (_r != 0) return fail(e, AZ_ERR_HIP, "%s failed: %s (%s:%d)", #call, (R)->GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

static void dist_destroy(az_engine *e)
{
    if (e->comm && g_rccl.h) (void)g_rccl.CommDestroy(e->comm);
    e->comm = nullptr;
    e->dist_rank = 0;
    e->dist_world = 1;
    dev_free(e->dist_buf);
    dev_free(e->dist_pack);
}

extern "C" int az_dist_unique_id(void *id)
{
    if (!id) return fail(nullptr, AZ_ERR_INVALID, "az_dist_unique_id: null argument");
    std::string err;
    const Rccl *R = rccl_load(&err);
    if (!R) return fail(nullptr, AZ_ERR_HIP, "%s", err.c_str());
    nccl_id nid;
    nccl_result rc = R->GetUniqueId(&nid);
    if (rc != 0) return fail(nullptr, AZ_ERR_HIP, "ncclGetUniqueId failed: %s", R->GetErrorString(rc));
    memcpy(id, nid.internal, AZ_DIST_ID_BYTES);
    return AZ_OK;
}

extern "C" int az_dist_init(az_engine *e, const void *id, int rank, int world)
{
    if (!e || !id || world < 1 || rank < 0 || rank >= world) return fail(e, AZ_ERR_INVALID, "az_dist_init: bad argument");
    if (e->run.open) return fail(e, AZ_ERR_STATE, "az_dist_init: an episode is open");
    std::string err;
    const Rccl *R = rccl_load(&err);
    if (!R) return fail(e, AZ_ERR_HIP, "%s", err.c_str());
    DEVICE_GUARD(e);
    dist_destroy(e);
    nccl_id nid;
    memcpy(nid.internal, id, AZ_DIST_ID_BYTES);
    void *comm = nullptr;
    NCCLCHECK(e, R, R->CommInitRank(&comm, world, nid, rank));
    e->comm = comm;
    e->dist_rank = rank;
    e->dist_world = world;
    return dev_alloc(e, e->dist_buf, (size_t)(world + 1) * 64 * sizeof(int64_t));
}

extern "C" int az_dist_rank(const az_engine *e) { return e ? e->dist_rank : AZ_ERR_INVALID; }
extern "C" int az_dist_world(const az_engine *e) { return e ? e->dist_world : AZ_ERR_INVALID; }

// records this rank contributes: the last episode's, 0 without one
static int64_t dist_local_records(const az_engine *e)
{
    if (!e->have_episode) return 0;
    int64_t n = 0;
    for (int g = 0; g < e->episode_games; g++) n += e->h_nply[g];
    return n;
}

static int dist_counts(az_engine *e, const Rccl *R, std::vector<int64_t> &counts)
{
    const int W = e->dist_world;
    int64_t *dv = (int64_t *)e->dist_buf.p;          // [0] mine, [64 ..] everybody's
    const int64_t mine = dist_local_records(e);
    HIPCHECK(e, hipMemcpyAsync(dv, &mine, 8, hipMemcpyHostToDevice, e->stream));
    NCCLCHECK(e, R, R->AllGather(dv, dv + 64, 1, NCCL_INT64, e->comm, e->stream));
    counts.assign(W, 0);
    HIPCHECK(e, az_memcpy(e->stream, counts.data(), dv + 64, (size_t)W * 8, hipMemcpyDeviceToHost));
    return AZ_OK;
}

extern "C" int az_dist_counts(az_engine *e, int64_t *counts)
{
    if (!e || !counts) return fail(e, AZ_ERR_INVALID, "az_dist_counts: bad argument");
    if (!e->comm) return fail(e, AZ_ERR_STATE, "az_dist_counts: az_dist_init first");
    DEVICE_GUARD(e);
    std::vector<int64_t> c;
    const int rc = dist_counts(e, &g_rccl, c);
    if (rc) return rc;
    memcpy(counts, c.data(), c.size() * 8);
    return AZ_OK;
}

extern "C" int az_dist_gather_records(az_engine *e, int dst, void *packed_dev)
{
    if (!e || dst < -1) return fail(e, AZ_ERR_INVALID, "az_dist_gather_records: bad argument");
    if (!e->comm) return fail(e, AZ_ERR_STATE, "az_dist_gather_records: az_dist_init first");
    if (dst >= e->dist_world) return fail(e, AZ_ERR_INVALID, "az_dist_gather_records: rank %d of %d", dst, e->dist_world);
    const Rccl *R = &g_rccl;
    DEVICE_GUARD(e);
    const int W = e->dist_world, me = e->dist_rank;
    const bool receive = dst < 0 || dst == me;
    if (receive && !packed_dev) return fail(e, AZ_ERR_INVALID, "az_dist_gather_records: a receiving rank needs a destination buffer");
    std::vector<int64_t> counts;
    int rc = dist_counts(e, R, counts);
    if (rc) return rc;
    const int64_t rb = record_bytes(e->nn), mine = counts[me];
    std::vector<int64_t> off(W + 1, 0);
    for (int r = 0; r < W; r++) off[r + 1] = off[r] + counts[r] * rb;
    // this rank's records: straight into their place of the destination where this rank receives, else into a send buffer
    unsigned char *out = (unsigned char *)packed_dev;
    void *my_dev = nullptr;
    if (mine > 0) {
        if (receive) my_dev = out + off[me];
        else {
            if ((rc = dev_alloc(e, e->dist_pack, (size_t)(mine * rb), false))) return rc;
            my_dev = e->dist_pack.p;
        }
        if ((rc = az_selfplay_pack(e, my_dev))) return rc;
    }
    if (W == 1) return AZ_OK;
    NCCLCHECK(e, R, R->GroupStart());
    nccl_result gr = 0;
    if (mine > 0)
        for (int r = 0; r < W && gr == 0; r++)
            if (r != me && (dst < 0 || dst == r)) gr = R->Send(my_dev, (size_t)(mine * rb), NCCL_UINT8, r, e->comm, e->stream);
    if (receive)
        for (int r = 0; r < W && gr == 0; r++)
            if (r != me && counts[r] > 0) gr = R->Recv(out + off[r], (size_t)(counts[r] * rb), NCCL_UINT8, r, e->comm, e->stream);
    const nccl_result ge = R->GroupEnd();
    if (gr != 0 || ge != 0) return fail(e, AZ_ERR_HIP, "RCCL send/recv of the records failed: %s", R->GetErrorString(gr != 0 ? gr : ge));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    return AZ_OK;
}

extern "C" int az_dist_allreduce_sum(az_engine *e, int64_t *values, int n)
{
    if (!e || !values || n < 1 || n > 64) return fail(e, AZ_ERR_INVALID, "az_dist_allreduce_sum: 1..64 values");
    if (!e->comm) return fail(e, AZ_ERR_STATE, "az_dist_allreduce_sum: az_dist_init first");
    const Rccl *R = &g_rccl;
    DEVICE_GUARD(e);
    int64_t *dv = (int64_t *)e->dist_buf.p;
    HIPCHECK(e, hipMemcpyAsync(dv, values, (size_t)n * 8, hipMemcpyHostToDevice, e->stream));
    NCCLCHECK(e, R, R->AllReduce(dv, dv, (size_t)n, NCCL_INT64, NCCL_SUM, e->comm, e->stream));
    HIPCHECK(e, az_memcpy(e->stream, values, dv, (size_t)n * 8, hipMemcpyDeviceToHost));
    return AZ_OK;
}

extern "C" int az_dist_broadcast(az_engine *e, void *buf_dev, int64_t bytes, int root)
{
    if (!e || !buf_dev || bytes < 0) return fail(e, AZ_ERR_INVALID, "az_dist_broadcast: bad argument");
    if (!e->comm) return fail(e, AZ_ERR_STATE, "az_dist_broadcast: az_dist_init first");
    if (root < 0 || root >= e->dist_world) return fail(e, AZ_ERR_INVALID, "az_dist_broadcast: root %d of %d", root, e->dist_world);
    if (bytes == 0) return AZ_OK;
    const Rccl *R = &g_rccl;
    DEVICE_GUARD(e);
    NCCLCHECK(e, R, R->Broadcast(buf_dev, buf_dev, (size_t)bytes, NCCL_UINT8, root, e->comm, e->stream));
    HIPCHECK(e, hipStreamSynchronize(e->stream));
    return AZ_OK;
}
