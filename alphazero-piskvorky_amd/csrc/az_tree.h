// az_tree.h -- per-wavefront MCTS kernels (select / expand / backup / move) over an array tree in HBM.
//
// One 64-lane wavefront owns one game slot.  A node that has been expanded owns one ROW of
// RW = roundup(n*n, 64) edge records (16 B each: W f64, P f32, N u16, child-row u16), indexed by
// board cell, so a level of the descent is CPL = RW/64 coalesced 16-byte loads per lane and the
// reference's "first maximal child in row-major order" (mcts.py:71) is a (max score, min cell)
// wave reduction.  Boards are two bit-planes of 4 x u64.  Reference semantics: mcts.py:25-183,
// games.py:35-166, self_play.py:48-73, evaluator.py:64-90 (cited inline).
#pragma once
#include "az_device.h"

enum { LEAF_NONE = 0, LEAF_EXPAND = 1, LEAF_TERM_LOSS = 2, LEAF_TERM_DRAW = 3, LEAF_ROOT = 4,
       LEAF_REUSE = 5,        // root retained from the previous ply's search (subtree reuse): no evaluation, noise mix only
       LEAF_EXPAND_HIT = 6, LEAF_ROOT_HIT = 7,   // evaluation cache hit: logits / hidden row already in place, the net kernels skip the slot
       LEAF_DUP_BASE = 16 };  // virtual-loss batching: LEAF_DUP_BASE + i = same leaf as item i of this batch (no second evaluation)
__host__ __device__ constexpr bool leaf_needs_net(int kind) { return kind == LEAF_ROOT || kind == LEAF_EXPAND; }
constexpr int VL_MAX = 32;             // leaves per batch in virtual-loss mode (5 spare bits of the edge's visit field)
constexpr int EDGE_N_BITS = 11, EDGE_N_MASK = (1 << EDGE_N_BITS) - 1;   // visit count 0..1024 in the low bits, in-flight count above
constexpr int REUSE_MAX_ROWS = 1024;   // rows per slot (S + 2) the in-place compaction of k_move can renumber
constexpr int CNT_STRIDE = 8;  // per-slot counters: expansions, simulations, terminal hits, depth sum, reused roots, duplicate leaves,
                               // cache lookups, cache hits
enum { SLOT_IDLE = 0, SLOT_ACTIVE = 1, SLOT_FINISHED = 2 };

struct __attribute__((aligned(16))) Edge {
    double W;              // mcts.py:38 total value (float64 like the reference)
    float P;               // mcts.py:63 prior (float32 value widened on use)
    unsigned short N;      // mcts.py:37 visit count
    unsigned short child;  // row of the expanded child node, 0 = not expanded (row 0 is the root)
};

struct DevState {
    int B, R, S, k, max_plies, add_noise, arena, total_games;
    double c_puct, w_noise;
    float one_minus_w;
    // --- slots (current real game per slot) ---
    u64 *board;            // [B][8] absolute bit-planes: words 0-3 X, 4-7 O
    int *s_game, *s_ply, *s_player, *s_last, *s_status, *s_net;
    // --- search state per slot ---
    Edge *edges;           // [B][R][RW]
    int *rows_used;        // [B]
    unsigned *path;        // [B][PATH] (row << 16 | cell)
    int *depth;            // [B]
    int *leaf_kind;        // [B]
    u64 *leaf;             // [B][8] mover-relative planes of the pending leaf: words 0-3 side to move, 4-7 opponent
    int *leaf_last;        // [B]
    // --- evaluator outputs ---
    float *logits;         // [B][RW]
    float *vhid;           // [B][64]
    const float *v2w[2];   // value_fc2.weight per net slot
    const float *v2b[2];
    // --- RNG tapes per game (numpy legacy stream, see az_rng.cpp) ---
    const double *noise;   // [G][noise_stride]
    long long noise_stride;
    const double *u;       // [G][nn]
    const int *noise_off;  // [nn+1] offset of ply m inside a game's noise tape
    // --- tables ---
    const double *T_table; // temperature by ply (self-play) or step (arena)
    const float *log_table;   // float32 log(N + 1e-8), N = 0..S
    const double *sqrt_table; // sqrt(N + 1e-8), N = 0..S+1
    // --- records per game: index game*nn + ply ---
    u64 *rec_planes;       // [G*nn][8] mover-relative planes before the move
    short *rec_last, *rec_action;
    unsigned char *rec_mover;
    float *rec_pi;         // [G*nn][nn]
    unsigned short *rec_visits; // [G*nn][nn]
    int *g_nply, *g_result;
    // --- bookkeeping ---
    unsigned long long *cnt;   // [B][CNT_STRIDE] expansions, simulations, terminal hits, depth sum, reused roots
    // --- opt-in subtree reuse between plies (the reference's TODO, mcts.py:17-22,106) ---
    int reuse;             // 0: every ply searches from a fresh root like the reference
    int *carried;          // [B] visits the retained root already holds (sum of its children's N), -1 = fresh root
    int *next_game;        // device counter
    int *active;           // number of active slots after refill
    // --- opt-in evaluation cache (the reference's TODO, mcts.py:17 DEFAULT_CACHE_SIZE, mcts.py:22 "add caching") ---
    float *cache;          // [cache_mask + 1][CACHE_HDR + RW + 64] floats, nullptr = off; shared by the lanes of an engine
    unsigned cache_mask;   // entries - 1 (a power of two)
    unsigned cache_gen;    // bumped by every weight load: entries of older weights never match
    // --- opt-in virtual-loss batching (the reference's TODO, mcts.py:17-22): L leaves per game and evaluation batch ---
    int L;                 // 1 = the reference's sequential simulation loop (mcts.py:123-141)
    int *it_status, *it_net;   // [B*L] per evaluation item, what the net kernels read as s_status / s_net (alias them when L == 1)
    int *leaf_sym;         // [B] opt-in random-symmetry leaf evaluation (az_set_leaf_symmetry): symmetry 0..7 the net sees the pending leaf in; nullptr = off
    int ext_eval;          // the pending rows were filled by an evaluator outside the engine (az_search_callback): priors and value as given
    unsigned game_key0;    // leaf-symmetry hash key of game id 0 = low 32 bits of the episode's seed0: key(g) = game_key0 + g is the game's
                           // seed, a GLOBAL name of the game that does not depend on which rank / slot / lane plays it
};

template <int N>
struct TreeGeo {
    static constexpr int n = N, nn = N * N, RW = ((nn + 63) / 64) * 64, CPL = RW / 64, PATH = nn + 1;
};

__device__ __forceinline__ void wave_mem_sync()
{
    // orders this wave's global/LDS writes before its later reads (cross-lane hand-off inside one wave)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}


// ------------------------------------------------------------------------------------------------
// Evaluation cache (opt-in; the reference only names it: mcts.py:17 DEFAULT_CACHE_SIZE = 500_000, mcts.py:22 "TODO: add
// caching").  Key = what the net sees: mover planes, opponent planes, last move (games.py:86-129 encode) + which net +
// the weights generation.  Value = the net's raw outputs for that input, policy logits and value-head hidden row, i.e.
// exactly the floats the trunk + FC kernels would produce again, so a hit changes nothing downstream (visit counts are
// bit-identical with the cache on or off).  Direct-mapped table in HBM shared by every lane of the engine; entries are
// written and read by whole wavefronts with plain loads/stores and carry a checksum bound to the key: an entry torn by
// a concurrent writer (another lane's kernel) fails the check and counts as a miss.
//   entry = [9 x u64 key][u64 checksum][pad to 128 B][RW logits][64 hidden]
// ------------------------------------------------------------------------------------------------
constexpr int CACHE_HDR = 32;      // floats
template <int N>
struct CacheGeo {
    static constexpr int CES = CACHE_HDR + TreeGeo<N>::RW + 64;     // floats per entry
};
__device__ __forceinline__ u64 cache_mix(u64 h, u64 v)
{
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 33;
    return h;
}
__device__ __forceinline__ u64 cache_key8(int last, int net, unsigned gen)
{
    return (u64)(unsigned)(last + 1) | ((u64)(unsigned)net << 16) | ((u64)gen << 32);
}
__device__ __forceinline__ u64 cache_hash(const Plane &me, const Plane &opp, u64 k8)
{
    u64 h = 0x243F6A8885A308D3ull;
#pragma unroll
    for (int q = 0; q < 4; q++) h = cache_mix(h, me.w[q]);
#pragma unroll
    for (int q = 0; q < 4; q++) h = cache_mix(h, opp.w[q]);
    return cache_mix(h, k8);
}
// the key word lane `lane` (0..8) is responsible for
__device__ __forceinline__ u64 cache_keyword(const Plane &me, const Plane &opp, u64 k8, int lane)
{
    u64 w = k8;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        w = lane == q ? me.w[q] : w;
        w = lane == 4 + q ? opp.w[q] : w;
    }
    return w;
}
template <int CPL>
__device__ __forceinline__ u64 cache_checksum(const float (&x)[CPL], float h, int lane, u64 hsh)
{
    unsigned a = 0u, b = 0u;
#pragma unroll
    for (int i = 0; i <= CPL; i++) {
        const unsigned w = __float_as_uint(i < CPL ? x[i < CPL ? i : 0] : h);
        const unsigned pos = (unsigned)(lane + 64 * i);
        a ^= az_fmix32(w + pos * 0x9E3779B1u);
        b ^= az_fmix32((w ^ 0xA5A5A5A5u) + pos * 0x85EBCA77u + 0x5BD1E995u);
    }
    a = wave_xor_u(a);
    b = wave_xor_u(b);
    return ((((u64)a) << 32) | (u64)b) ^ hsh;
}
// wave-uniform call: true = the entry of (me, opp, last, net) was found intact; its rows are then in x / h
template <int N>
__device__ __forceinline__ bool cache_lookup(const DevState &d, const Plane &me, const Plane &opp, int last, int net, int lane,
                                             float (&x)[TreeGeo<N>::CPL], float &h)
{
    typedef TreeGeo<N> G;
    const u64 k8 = cache_key8(last, net, d.cache_gen);
    const u64 hsh = cache_hash(me, opp, k8);
    const float *ent = d.cache + (size_t)((unsigned)hsh & d.cache_mask) * CacheGeo<N>::CES;
    const u64 *kp = reinterpret_cast<const u64 *>(ent);
    const u64 mine = cache_keyword(me, opp, k8, lane);
    const u64 got = lane < 10 ? kp[lane] : 0ull;
#pragma unroll
    for (int i = 0; i < G::CPL; i++) x[i] = ent[CACHE_HDR + lane + 64 * i];
    h = ent[CACHE_HDR + G::RW + lane];
    const bool key_ok = __ballot(lane < 9 && got != mine) == 0ull;
    const u64 cs = cache_checksum<G::CPL>(x, h, lane, hsh);
    const u64 stored = (u64)__shfl((long long)got, 9, 64);
    return key_ok && cs == stored;
}
template <int N>
__device__ __forceinline__ void cache_insert(const DevState &d, const Plane &me, const Plane &opp, int last, int net, int lane,
                                             const float (&x)[TreeGeo<N>::CPL], float h)
{
    typedef TreeGeo<N> G;
    const u64 k8 = cache_key8(last, net, d.cache_gen);
    const u64 hsh = cache_hash(me, opp, k8);
    float *ent = d.cache + (size_t)((unsigned)hsh & d.cache_mask) * CacheGeo<N>::CES;
    u64 *kp = reinterpret_cast<u64 *>(ent);
    const u64 cs = cache_checksum<G::CPL>(x, h, lane, hsh);
    const u64 mine = lane == 9 ? cs : cache_keyword(me, opp, k8, lane);
    if (lane < 10) kp[lane] = mine;
#pragma unroll
    for (int i = 0; i < G::CPL; i++) ent[CACHE_HDR + lane + 64 * i] = x[i];
    ent[CACHE_HDR + G::RW + lane] = h;
}

// With random-symmetry leaf evaluation (az_set_leaf_symmetry) the net's outputs depend on the symmetry it was shown, so the
// symmetry is part of the key (folded into the net field), and a hit lands in the logits row the way the net would have written
// it -- in image coordinates, board cell j at image cell sym_src(t^-1, j) -- because the tree step maps every row back.
__device__ __forceinline__ int cache_net_key(int net, int sym) { return net | (sym << 1); }
template <int N>
__device__ __forceinline__ void cache_hit_store(float *lg, const float (&cx)[TreeGeo<N>::CPL], int lane, bool sym_on, int sym)
{
    typedef TreeGeo<N> G;
    if (sym_on) {
        const int ti = sym_inverse(sym);
#pragma unroll
        for (int i = 0; i < G::CPL; i++) {
            const int j = lane + 64 * i, r = j / N;
            if (j < G::nn) lg[sym_src(ti, r, j - r * N, N)] = cx[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < G::CPL; i++) lg[lane + 64 * i] = cx[i];
    }
}

// controller.py:49 softmax over all n^2 logits (no legality mask) in the canonical wave order, and the
// value tail value_fc2 + tanh (net.py:70) as one k-ordered fma chain.  P[i] is the prior of cell lane+64*i.
template <int N>
__device__ __forceinline__ void eval_tail(const DevState &d, int b, int lane, float (&P)[TreeGeo<N>::CPL], float &v)
{
    typedef TreeGeo<N> G;
    const float *lg = d.logits + (size_t)b * G::RW;
    float x[G::CPL];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < G::CPL; i++) {
        int j = lane + 64 * i;
        x[i] = j < G::nn ? lg[j] : -INFINITY;
        mx = fmaxf(mx, x[i]);
    }
    mx = wave_max_f(mx);
    float part = 0.0f;
#pragma unroll
    for (int i = 0; i < G::CPL; i++) {
        int j = lane + 64 * i;
        P[i] = 0.0f;
        if (j < G::nn) {
            P[i] = az_expf(x[i] - mx);
            part = part + P[i];
        }
    }
    float s = wave_sum_butterfly(part);
#pragma unroll
    for (int i = 0; i < G::CPL; i++) P[i] = P[i] / s;
    const float *h = d.vhid + (size_t)b * 64;
    const float *w2 = d.v2w[d.s_net[b]];
    float acc = 0.0f;
    for (int i = 0; i < 64; i++) acc = __builtin_fmaf(h[i], w2[i], acc);
    v = az_tanhf(acc + d.v2b[d.s_net[b]][0]);
}

// az_net_eval: softmax policy + value for boards staged as pending leaves (controller.py:39-53)
template <int N>
__global__ __launch_bounds__(256) void k_eval_tail(DevState d, int count, float *policy, float *value)
{
    typedef TreeGeo<N> G;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= count) return;
    float P[G::CPL], v;
    eval_tail<N>(d, b, lane, P, v);
#pragma unroll
    for (int i = 0; i < G::CPL; i++) {
        int j = lane + 64 * i;
        if (j < G::nn) policy[(size_t)b * G::nn + j] = P[i];
    }
    if (lane == 0) value[b] = v;
}

#ifdef AZ_ENGINE_TU   // size-independent kernels live in the engine translation unit only
// ------------------------------------------------------------------------------------------------
// k_begin: the root of every active slot becomes the pending evaluation (mcts.py:106-109)
// ------------------------------------------------------------------------------------------------
__global__ void k_begin(DevState d)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= d.B) return;
    const int L = d.L;
    const size_t it = (size_t)b * L;                     // evaluation items of this game: the root is item 0
    if (d.s_status[b] != SLOT_ACTIVE) {
        for (int j = 0; j < L; j++) {
            d.leaf_kind[it + j] = LEAF_NONE;
            if (L > 1) d.it_status[it + j] = SLOT_IDLE;
        }
        return;
    }
    int pl = d.s_player[b];
    const u64 *bd = d.board + (size_t)b * 8;
    u64 *lf = d.leaf + it * 8;
    for (int i = 0; i < 4; i++) {
        lf[i] = pl == 1 ? bd[i] : bd[4 + i];
        lf[4 + i] = pl == 1 ? bd[4 + i] : bd[i];
    }
    d.leaf_last[it] = d.s_last[b];
    if (d.leaf_sym) d.leaf_sym[it] = leaf_sym_of((int)(d.game_key0 + (unsigned)d.s_game[b]), d.s_ply[b], 0);
    d.leaf_kind[it] = (d.reuse && d.carried[b] >= 0) ? LEAF_REUSE : LEAF_ROOT;
    d.depth[it] = 0;
    const int netid = d.arena ? (pl == 1 ? 0 : 1) : 0;   // evaluator.py:73-79: each side searches with its own net
    d.s_net[b] = netid;
    for (int j = 1; j < L; j++) d.leaf_kind[it + j] = LEAF_NONE;
    if (L > 1)
        for (int j = 0; j < L; j++) { d.it_status[it + j] = SLOT_ACTIVE; d.it_net[it + j] = netid; }
}

#endif  // AZ_ENGINE_TU

// ------------------------------------------------------------------------------------------------
#ifndef AZ_STEP_WAVES
#define AZ_STEP_WAVES 4      // games (wavefronts) per k_step workgroup
#endif
// k_step: consume the evaluation of the pending leaf (expand + backup), then select the next leaf.
// Latency-bound (one wave per game, ~4 waves per CU): every load that does not depend on another load is
// issued up front in one round trip; the sqrt table lives in LDS; the chosen edge is broadcast by shuffle.
// Dynamic LDS: (S + 2) doubles.
// ------------------------------------------------------------------------------------------------
template <int N, bool SYNTH>
__global__ __launch_bounds__(AZ_STEP_WAVES * 64) void k_step(DevState d, int rootN, int do_select)
{
    typedef TreeGeo<N> G;
    extern __shared__ double sq_lds[];                 // np.sqrt(N + 1e-8), N = 0..S+1 (mcts.py:73)
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * AZ_STEP_WAVES + (threadIdx.x >> 6);
    const bool inb = b < d.B;
    const int bb = inb ? b : 0;
    for (int i = threadIdx.x; i < d.S + 2; i += AZ_STEP_WAVES * 64) sq_lds[i] = d.sqrt_table[i];

    // ---- independent loads, all in flight together ----
    const int status = d.s_status[bb];
    const int kind_raw = d.leaf_kind[bb];
    // a cache hit (LEAF_*_HIT) is consumed exactly like the evaluation it stands for; it is just not inserted again
    const int kind = kind_raw == LEAF_EXPAND_HIT ? LEAF_EXPAND : (kind_raw == LEAF_ROOT_HIT ? LEAF_ROOT : kind_raw);
    const int depth0 = d.depth[bb];
    const int rows0 = d.rows_used[bb];
    const int pl = d.s_player[bb];
    const int slast = d.s_last[bb];
    const int netid = d.s_net[bb];
    const int leaf_last = d.leaf_last[bb];
    const int game = d.s_game[bb];
    const int ply = d.s_ply[bb];
    const int carried = d.reuse ? d.carried[bb] : -1;
    const Plane lme = pl_load(d.leaf + (size_t)bb * 8), lopp = pl_load(d.leaf + (size_t)bb * 8 + 4);
    const Plane bX = pl_load(d.board + (size_t)bb * 8), bO = pl_load(d.board + (size_t)bb * 8 + 4);
    unsigned *path = d.path + (size_t)bb * G::PATH;
    const unsigned path_l = path[lane];                // PATH = n*n+1 >= 26 entries; lanes beyond read a neighbour's slot (unused)
    float x[G::CPL];
    float h_l = 0.0f, w2a = 0.0f, w2b = 0.0f, b2a = 0.0f, b2b = 0.0f;
    if (!SYNTH) {
        const float *lg = d.logits + (size_t)bb * G::RW;
        if (d.leaf_sym) {
            // the net saw the leaf under symmetry t: board cell j sits at image cell sym_src(t^-1, j)
            const int ti = sym_inverse(d.leaf_sym[bb]);
#pragma unroll
            for (int i = 0; i < G::CPL; i++) {
                const int j = lane + 64 * i, r = j / N;
                x[i] = j < G::nn ? lg[sym_src(ti, r, j - r * N, N)] : 0.0f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < G::CPL; i++) x[i] = lg[lane + 64 * i];
        }
        h_l = d.vhid[(size_t)bb * 64 + lane];
        if (d.v2w[0]) { w2a = d.v2w[0][lane]; b2a = d.v2b[0][0]; }
        if (d.v2w[1]) { w2b = d.v2w[1][lane]; b2b = d.v2b[1][0]; }
    }
    __syncthreads();
    if (!inb || status != SLOT_ACTIVE) return;
    Edge *rows = d.edges + (size_t)b * d.R * G::RW;

    // ---------------- stage 1: evaluation -> expand -> backup ----------------
    if (kind == LEAF_REUSE) {
        // retained root: its row (compacted to row 0 by k_move) already holds the un-noised priors and the visit
        // statistics of the previous search; only the fresh Dirichlet sample is mixed in (same arithmetic as a new root)
        if (d.add_noise) {
            Plane occ;
#pragma unroll
            for (int q = 0; q < 4; q++) occ.w[q] = bX.w[q] | bO.w[q];
            const double *nz = d.noise + (size_t)game * d.noise_stride + d.noise_off[ply];
#pragma unroll
            for (int i = 0; i < G::CPL; i++) {
                int j = lane + 64 * i;
                if (j < G::nn && !pl_get(occ, j)) {
                    int rank = j - pl_rank(occ, j);
                    Edge *e = rows + j;
                    float scaled = d.one_minus_w * e->P;
                    e->P = (float)((double)scaled + d.w_noise * nz[rank]);
                }
            }
        }
        if (lane == 0) d.cnt[(size_t)b * CNT_STRIDE + 4] += 1ull;
        wave_mem_sync();
    } else if (kind != LEAF_NONE) {
        float v = 0.0f;
        if (kind == LEAF_ROOT || kind == LEAF_EXPAND) {
            float P[G::CPL];
            if (SYNTH) {
                // build-owned deterministic evaluator (test hook behind the policy_value_fn seam, mcts.py:87-93)
                unsigned hx = 0;
#pragma unroll
                for (int i = 0; i < G::CPL; i++) {
                    int j = lane + 64 * i;
                    if (j < G::nn) {
                        unsigned code = pl_get(lme, j) ? 1u : (pl_get(lopp, j) ? 2u : 0u);
                        hx ^= az_fmix32((unsigned)j * 3u + code + 0x9E3779B9u);
                    }
                }
                unsigned hs = wave_xor_u(hx);
                hs ^= az_fmix32(0x51ED270Bu + (unsigned)(leaf_last + 1));
#pragma unroll
                for (int i = 0; i < G::CPL; i++) {
                    int j = lane + 64 * i;
                    unsigned r = az_fmix32(hs + (unsigned)(j + 1) * 0x9E3779B1u);
                    P[i] = (float)(((r >> 8) & 0xFFFFu) + 1u) * 0x1p-23f;
                }
                int vv = (int)(az_fmix32(hs ^ 0x7F4A7C15u) & 0x1FFu);
                v = (float)(vv - 256) / 256.0f;
            } else {
                if (d.cache && kind_raw == kind) {            // a fresh evaluation: remember it
                    float xc[G::CPL];
#pragma unroll
                    for (int i = 0; i < G::CPL; i++) xc[i] = lane + 64 * i < G::nn ? x[i] : 0.0f;
                    cache_insert<N>(d, lme, lopp, leaf_last, cache_net_key(netid, d.leaf_sym ? d.leaf_sym[bb] : 0), lane, xc, h_l);
                }
                if (d.ext_eval) {
                    // the evaluator lives outside the engine (policy_value_fn seam, mcts.py:87-93,109,137): the row holds
                    // the priors it returned, as they are, and the first hidden entry its value
#pragma unroll
                    for (int i = 0; i < G::CPL; i++) P[i] = lane + 64 * i < G::nn ? x[i] : 0.0f;
                    v = __shfl(h_l, 0, 64);
                } else {
                // controller.py:49 softmax over all n^2 logits (no legality mask), canonical wave order
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < G::CPL; i++) {
                    if (lane + 64 * i >= G::nn) x[i] = -INFINITY;
                    mx = fmaxf(mx, x[i]);
                }
                mx = wave_max_f(mx);
                float part = 0.0f;
#pragma unroll
                for (int i = 0; i < G::CPL; i++) {
                    P[i] = 0.0f;
                    if (lane + 64 * i < G::nn) {
                        P[i] = az_expf(x[i] - mx);
                        part = part + P[i];
                    }
                }
                const float ssum = wave_sum_butterfly(part);
#pragma unroll
                for (int i = 0; i < G::CPL; i++) P[i] = P[i] / ssum;
                // value tail: value_fc2 + tanh (net.py:70), one k-ordered fma chain over the lane-held operands
                const float w2_l = netid ? w2b : w2a;
                const float acc = value_fc2_chain(h_l, w2_l);
                v = az_tanhf(acc + (netid ? b2b : b2a));
                }
            }
            if (kind == LEAF_ROOT && d.add_noise) {
                // mcts.py:113-116; float32 multiply, float64 add, float32 store
                Plane occ;
#pragma unroll
                for (int q = 0; q < 4; q++) occ.w[q] = lme.w[q] | lopp.w[q];
                const double *nz = d.noise + (size_t)game * d.noise_stride + d.noise_off[ply];
#pragma unroll
                for (int i = 0; i < G::CPL; i++) {
                    int j = lane + 64 * i;
                    if (j < G::nn && !pl_get(occ, j)) {
                        int rank = j - pl_rank(occ, j);
                        float scaled = d.one_minus_w * P[i];
                        P[i] = (float)((double)scaled + d.w_noise * nz[rank]);
                    }
                }
            }
            // mcts.py:50-64 expand: one edge per cell (occupied cells are never selected)
            const int row = kind == LEAF_ROOT ? 0 : rows0;
#pragma unroll
            for (int i = 0; i < G::CPL; i++) {
                int j = lane + 64 * i;
                Edge e;
                e.W = 0.0; e.P = P[i]; e.N = 0; e.child = 0;
                rows[(size_t)row * G::RW + j] = e;
            }
            if (lane == 0) d.rows_used[b] = row + 1;
            if (kind == LEAF_EXPAND) {
                const unsigned pe = depth0 - 1 < 64 ? (unsigned)__shfl((int)path_l, depth0 - 1, 64) : path[depth0 - 1];
                if (lane == 0) rows[(size_t)(pe >> 16) * G::RW + (pe & 0xFFFFu)].child = (unsigned short)row;
            }
        }
        if (kind != LEAF_ROOT) {
            // mcts.py:132-134,141,76-82: value w.r.t. the side to move at the leaf, backed up with alternating sign
            const double value = kind == LEAF_EXPAND ? (double)v : (kind == LEAF_TERM_LOSS ? -1.0 : 0.0);
            for (int dd = lane; dd < depth0; dd += 64) {
                const unsigned pe = dd < 64 ? path_l : path[dd];
                Edge *e = rows + (size_t)(pe >> 16) * G::RW + (pe & 0xFFFFu);
                const double val = ((depth0 - 1 - dd) & 1) ? value : -value;
                e->N = (unsigned short)(e->N + 1);
                e->W = e->W + val;
            }
            if (lane == 0) {
                unsigned long long *c = d.cnt + (size_t)b * CNT_STRIDE;
                c[0] += kind == LEAF_EXPAND ? 1ull : 0ull;
                c[1] += 1ull;
                c[2] += kind == LEAF_EXPAND ? 0ull : 1ull;
                c[3] += (unsigned long long)depth0;
            }
        }
        wave_mem_sync();
    }

    // ---------------- stage 2: selection (mcts.py:124-129) ----------------
    if (!do_select || rootN < carried) {          // a retained root tops its visits up to S: idle until simulation `carried`
        if (lane == 0) d.leaf_kind[b] = LEAF_NONE;
        return;
    }
    Plane me = pl == 1 ? bX : bO;
    Plane opp = pl == 1 ? bO : bX;
    int row = 0, npar = rootN, depth = 0, last = slast, out_kind = LEAF_NONE;
    for (;;) {
        Plane occ;
#pragma unroll
        for (int q = 0; q < 4; q++) occ.w[q] = me.w[q] | opp.w[q];
        const double sq = sq_lds[npar];                       // np.sqrt(self.N + 1e-8), mcts.py:73
        double best = 0.0;
        int bi = -1, bN = 0, bC = 0;
#pragma unroll
        for (int i = 0; i < G::CPL; i++) {
            int j = lane + 64 * i;
            if (j < G::nn && !pl_get(occ, j)) {
                Edge e = rows[(size_t)row * G::RW + j];
                double Q = e.N ? e.W / (double)e.N : 0.0;     // mcts.py:80 Q = W/N (0.0 while unvisited)
                double sc = Q + ((d.c_puct * (double)e.P) * sq) / (double)(1 + (int)e.N);
                if (bi < 0 || sc > best) { best = sc; bi = j; bN = e.N; bC = e.child; }
            }
        }
        wave_argmax(best, bi);
        const int a = __builtin_amdgcn_readfirstlane(bi);
        if (a < 0) { out_kind = LEAF_NONE; break; }            // unreachable for a non-terminal root; never index with -1
        // the lane that owns cell a (a & 63) holds its edge: the global best is also that lane's best
        const int child = __builtin_amdgcn_readlane(bC, a & 63);
        const int an = __builtin_amdgcn_readlane(bN, a & 63);
        if (lane == 0) path[depth] = ((unsigned)row << 16) | (unsigned)a;
        depth++;
        pl_set(me, a);                                        // games.py:79-81 place, flip player, remember action
        Plane t = me; me = opp; opp = t;
        last = a;
        if (wins_through_wave(opp, a, N, d.k, lane)) { out_kind = LEAF_TERM_LOSS; break; }  // the side to move has lost
        if (pl_count(me) + pl_count(opp) == G::nn) { out_kind = LEAF_TERM_DRAW; break; }
        if (child == 0) { out_kind = LEAF_EXPAND; break; }    // mcts.py:127 node.is_leaf()
        npar = an;
        row = child;
    }
    if (!SYNTH && d.cache && out_kind == LEAF_EXPAND) {
        // the leaf's evaluation may be known already (an earlier ply's search, another game, a transposition)
        float cx[G::CPL], ch;
        const int sym = d.leaf_sym ? leaf_sym_of((int)(d.game_key0 + (unsigned)game), ply, rootN + 1) : 0;     // as stored below
        const bool hit = cache_lookup<N>(d, me, opp, last, cache_net_key(netid, sym), lane, cx, ch);
        if (hit) {
            cache_hit_store<N>(d.logits + (size_t)b * G::RW, cx, lane, d.leaf_sym != nullptr, sym);
            d.vhid[(size_t)b * 64 + lane] = ch;
            out_kind = LEAF_EXPAND_HIT;
        }
        if (lane == 0) {
            d.cnt[(size_t)b * CNT_STRIDE + 6] += 1ull;
            d.cnt[(size_t)b * CNT_STRIDE + 7] += hit ? 1ull : 0ull;
        }
    }
    if (lane == 0) {
        pl_store(d.leaf + (size_t)b * 8, me);
        pl_store(d.leaf + (size_t)b * 8 + 4, opp);
        d.leaf_last[b] = last;
        d.leaf_kind[b] = out_kind;
        d.depth[b] = depth;
        if (d.leaf_sym) d.leaf_sym[b] = leaf_sym_of((int)(d.game_key0 + (unsigned)game), ply, rootN + 1);     // this leaf is evaluation rootN + 1 of the search
    }
}

// Evaluation cache for ROOT evaluations: runs after k_begin, one wave per slot.
template <int N>
__global__ __launch_bounds__(256) void k_root_cache(DevState d)
{
    typedef TreeGeo<N> G;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= d.B || d.s_status[b] != SLOT_ACTIVE) return;
    const size_t it = (size_t)b * d.L;                 // the root is item 0 of its game
    if (d.leaf_kind[it] != LEAF_ROOT) return;
    const Plane me = pl_load(d.leaf + it * 8), opp = pl_load(d.leaf + it * 8 + 4);
    float cx[G::CPL], ch;
    const int sym = d.leaf_sym ? d.leaf_sym[it] : 0;
    const bool hit = cache_lookup<N>(d, me, opp, d.leaf_last[it], cache_net_key(d.s_net[b], sym), lane, cx, ch);
    if (hit) {
        cache_hit_store<N>(d.logits + it * G::RW, cx, lane, d.leaf_sym != nullptr, sym);
        d.vhid[it * 64 + lane] = ch;
    }
    if (lane == 0) {
        if (hit) d.leaf_kind[it] = LEAF_ROOT_HIT;
        d.cnt[(size_t)b * CNT_STRIDE + 6] += 1ull;
        d.cnt[(size_t)b * CNT_STRIDE + 7] += hit ? 1ull : 0ull;
    }
}

// ------------------------------------------------------------------------------------------------
// k_step_vl: the tree step with virtual-loss batching (opt-in; the reference's loop mcts.py:123-141 is strictly
// sequential and only lists "virtual loss" as a TODO, mcts.py:17-22).  A game owns L evaluation items; every launch
//   stage 1: finishes the simulations of the pending batch in selection order -- takes the in-flight visit back from the
//            path, expands the leaf with its evaluation and backs the value up exactly like mcts.py:136-141; a
//            "duplicate" (its leaf was already pending for an earlier item of the batch) backs up that item's value;
//   stage 2: selects the next batch: item j descends with every edge on the paths of items 0..j-1 counting one extra
//            visit that lost (N + 1, W - 1), and leaves its own in-flight mark on the edges it takes.
// The in-flight count lives in the 5 spare bits of the edge's 16-bit visit field; W is never touched by it, so taking
// a virtual visit back is exact.  Same arithmetic and operator order as the oracle's restatement (orc_cfg.vl); with
// L = 1 it reproduces k_step.  Dynamic LDS: (S + 2) doubles.
// ------------------------------------------------------------------------------------------------
template <int N, bool SYNTH>
__device__ __forceinline__ void vl_leaf_eval(const DevState &d, size_t it, const Plane &lme, const Plane &lopp, int leaf_last,
                                             int netid, bool fresh, int lane, float (&P)[TreeGeo<N>::CPL], float &v)
{
    typedef TreeGeo<N> G;
    if (SYNTH) {
        unsigned hx = 0;
#pragma unroll
        for (int i = 0; i < G::CPL; i++) {
            int j = lane + 64 * i;
            if (j < G::nn) {
                unsigned code = pl_get(lme, j) ? 1u : (pl_get(lopp, j) ? 2u : 0u);
                hx ^= az_fmix32((unsigned)j * 3u + code + 0x9E3779B9u);
            }
        }
        unsigned hs = wave_xor_u(hx);
        hs ^= az_fmix32(0x51ED270Bu + (unsigned)(leaf_last + 1));
#pragma unroll
        for (int i = 0; i < G::CPL; i++) {
            int j = lane + 64 * i;
            unsigned r = az_fmix32(hs + (unsigned)(j + 1) * 0x9E3779B1u);
            P[i] = (float)(((r >> 8) & 0xFFFFu) + 1u) * 0x1p-23f;
        }
        int vv = (int)(az_fmix32(hs ^ 0x7F4A7C15u) & 0x1FFu);
        v = (float)(vv - 256) / 256.0f;
        return;
    }
    float x[G::CPL];
    const float *lg = d.logits + it * G::RW;
    if (d.leaf_sym) {
        // the net saw this leaf under symmetry t: board cell j sits at image cell sym_src(t^-1, j)
        const int ti = sym_inverse(d.leaf_sym[it]);
#pragma unroll
        for (int i = 0; i < G::CPL; i++) {
            const int j = lane + 64 * i, r = j / N;
            x[i] = j < G::nn ? lg[sym_src(ti, r, j - r * N, N)] : 0.0f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < G::CPL; i++) x[i] = lane + 64 * i < G::nn ? lg[lane + 64 * i] : 0.0f;
    }
    const float h_l = d.vhid[it * 64 + lane];
    const float w2_l = d.v2w[netid][lane];
    const float b2 = d.v2b[netid][0];
    if (d.cache && fresh) cache_insert<N>(d, lme, lopp, leaf_last, cache_net_key(netid, d.leaf_sym ? d.leaf_sym[it] : 0), lane, x, h_l);
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < G::CPL; i++) {
        if (lane + 64 * i >= G::nn) x[i] = -INFINITY;
        mx = fmaxf(mx, x[i]);
    }
    mx = wave_max_f(mx);
    float part = 0.0f;
#pragma unroll
    for (int i = 0; i < G::CPL; i++) {
        P[i] = 0.0f;
        if (lane + 64 * i < G::nn) {
            P[i] = az_expf(x[i] - mx);
            part = part + P[i];
        }
    }
    const float ssum = wave_sum_butterfly(part);
#pragma unroll
    for (int i = 0; i < G::CPL; i++) P[i] = P[i] / ssum;
    const float acc = value_fc2_chain(h_l, w2_l);
    v = az_tanhf(acc + b2);
}

template <int N, bool SYNTH>
__global__ __launch_bounds__(256) void k_step_vl(DevState d, int sims_done, int nb_next)
{
    typedef TreeGeo<N> G;
    extern __shared__ double sq_lds[];                 // np.sqrt(N + 1e-8), N = 0..S+1 (mcts.py:73)
    __shared__ float vals_lds[4][VL_MAX];              // leaf values of the batch being finished (duplicates read them)
    __shared__ unsigned pend_lds[4][VL_MAX];           // leaf edge (row << 16 | cell) of every item waiting for an evaluation
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    for (int i = threadIdx.x; i < d.S + 2; i += 256) sq_lds[i] = d.sqrt_table[i];
    __syncthreads();
    if (b >= d.B || d.s_status[b] != SLOT_ACTIVE) return;
    const int L = d.L;
    const size_t it0 = (size_t)b * L;
    Edge *rows = d.edges + (size_t)b * d.R * G::RW;
    const int pl = d.s_player[b], slast = d.s_last[b], netid = d.s_net[b], game = d.s_game[b], ply = d.s_ply[b];
    const Plane bX = pl_load(d.board + (size_t)b * 8), bO = pl_load(d.board + (size_t)b * 8 + 4);
    float *vals = vals_lds[wv];
    unsigned *pend = pend_lds[wv];
    int rows_used = d.rows_used[b];
    unsigned long long c_exp = 0, c_sim = 0, c_term = 0, c_depth = 0, c_dup = 0, c_look = 0, c_hit = 0;

    // ---------------- stage 1: finish the pending batch in selection order ----------------
    for (int j = 0; j < L; j++) {
        const size_t it = it0 + j;
        const int kind_raw = d.leaf_kind[it];
        if (kind_raw == LEAF_NONE) continue;
        const int kind = kind_raw == LEAF_EXPAND_HIT ? LEAF_EXPAND : (kind_raw == LEAF_ROOT_HIT ? LEAF_ROOT : kind_raw);
        const int depth0 = d.depth[it];
        unsigned *path = d.path + it * G::PATH;
        float v = 0.0f;
        if (kind == LEAF_ROOT || kind == LEAF_EXPAND) {
            const Plane lme = pl_load(d.leaf + it * 8), lopp = pl_load(d.leaf + it * 8 + 4);
            const int leaf_last = d.leaf_last[it];
            float P[G::CPL];
            vl_leaf_eval<N, SYNTH>(d, it, lme, lopp, leaf_last, netid, kind_raw == kind, lane, P, v);
            if (kind == LEAF_ROOT && d.add_noise) {
                // mcts.py:113-116; float32 multiply, float64 add, float32 store
                Plane occ;
#pragma unroll
                for (int q = 0; q < 4; q++) occ.w[q] = lme.w[q] | lopp.w[q];
                const double *nz = d.noise + (size_t)game * d.noise_stride + d.noise_off[ply];
#pragma unroll
                for (int i = 0; i < G::CPL; i++) {
                    int c = lane + 64 * i;
                    if (c < G::nn && !pl_get(occ, c)) {
                        int rank = c - pl_rank(occ, c);
                        float scaled = d.one_minus_w * P[i];
                        P[i] = (float)((double)scaled + d.w_noise * nz[rank]);
                    }
                }
            }
            // mcts.py:50-64 expand: one edge per cell (occupied cells are never selected)
            const int row = kind == LEAF_ROOT ? 0 : rows_used;
#pragma unroll
            for (int i = 0; i < G::CPL; i++) {
                Edge e;
                e.W = 0.0; e.P = P[i]; e.N = 0; e.child = 0;
                rows[(size_t)row * G::RW + lane + 64 * i] = e;
            }
            rows_used = row + 1;
            if (kind == LEAF_EXPAND) {
                const unsigned pe = path[depth0 - 1];
                if (lane == 0) rows[(size_t)(pe >> 16) * G::RW + (pe & 0xFFFFu)].child = (unsigned short)row;
            }
        }
        vals[j] = v;                                    // every lane holds the same v
        if (kind != LEAF_ROOT) {
            // mcts.py:132-134,141,76-82: value w.r.t. the side to move at the leaf, backed up with alternating sign
            double value;
            if (kind == LEAF_EXPAND) value = (double)v;
            else if (kind >= LEAF_DUP_BASE) value = (double)vals[kind - LEAF_DUP_BASE];
            else value = kind == LEAF_TERM_LOSS ? -1.0 : 0.0;
            for (int dd = lane; dd < depth0; dd += 64) {
                const unsigned pe = path[dd];
                Edge *e = rows + (size_t)(pe >> 16) * G::RW + (pe & 0xFFFFu);
                const double val = ((depth0 - 1 - dd) & 1) ? value : -value;
                e->N = (unsigned short)(e->N - (1 << EDGE_N_BITS) + 1);      // the in-flight visit becomes a real one
                e->W = e->W + val;
            }
            c_exp += kind == LEAF_EXPAND ? 1ull : 0ull;
            c_dup += kind >= LEAF_DUP_BASE ? 1ull : 0ull;
            c_term += (kind == LEAF_TERM_LOSS || kind == LEAF_TERM_DRAW) ? 1ull : 0ull;
            c_sim += 1ull;
            c_depth += (unsigned long long)depth0;
        }
        wave_mem_sync();
    }

    // ---------------- stage 2: select the next batch (mcts.py:124-129 with in-flight visits counted as losses) ----------------
    const Plane rme = pl == 1 ? bX : bO, ropp = pl == 1 ? bO : bX;
    for (int j = 0; j < L; j++) {
        const size_t it = it0 + j;
        if (j >= nb_next) {
            if (lane == 0) d.leaf_kind[it] = LEAF_NONE;
            continue;
        }
        Plane me = rme, opp = ropp;
        int row = 0, npar = sims_done + j, depth = 0, last = slast, out_kind = LEAF_NONE;
        unsigned *path = d.path + it * G::PATH;
        unsigned leaf_edge = 0xFFFFFFFFu;
        for (;;) {
            Plane occ;
#pragma unroll
            for (int q = 0; q < 4; q++) occ.w[q] = me.w[q] | opp.w[q];
            const double sq = sq_lds[npar];                       // np.sqrt(self.N + 1e-8), mcts.py:73
            double best = 0.0;
            int bi = -1, bN = 0, bC = 0;
#pragma unroll
            for (int i = 0; i < G::CPL; i++) {
                int c = lane + 64 * i;
                if (c < G::nn && !pl_get(occ, c)) {
                    Edge e = rows[(size_t)row * G::RW + c];
                    const int fl = (int)e.N >> EDGE_N_BITS;               // simulations of this batch in flight through the edge
                    const int nv = ((int)e.N & EDGE_N_MASK) + fl;
                    const double wvv = e.W - (double)fl;
                    double Q = nv ? wvv / (double)nv : 0.0;
                    double sc = Q + ((d.c_puct * (double)e.P) * sq) / (double)(1 + nv);
                    if (bi < 0 || sc > best) { best = sc; bi = c; bN = nv; bC = e.child; }
                }
            }
            wave_argmax(best, bi);
            const int a = __builtin_amdgcn_readfirstlane(bi);
            if (a < 0) { out_kind = LEAF_NONE; break; }
            const int child = __builtin_amdgcn_readlane(bC, a & 63);
            const int an = __builtin_amdgcn_readlane(bN, a & 63);
            if (lane == (a & 63)) {
                Edge *e = rows + (size_t)row * G::RW + a;
                e->N = (unsigned short)(e->N + (1 << EDGE_N_BITS));
            }
            if (lane == 0) path[depth] = ((unsigned)row << 16) | (unsigned)a;
            leaf_edge = ((unsigned)row << 16) | (unsigned)a;
            depth++;
            pl_set(me, a);                                        // games.py:79-81 place, flip player, remember action
            Plane t = me; me = opp; opp = t;
            last = a;
            if (wins_through_wave(opp, a, N, d.k, lane)) { out_kind = LEAF_TERM_LOSS; break; }
            if (pl_count(me) + pl_count(opp) == G::nn) { out_kind = LEAF_TERM_DRAW; break; }
            if (child == 0) { out_kind = LEAF_EXPAND; break; }    // mcts.py:127 node.is_leaf()
            npar = an;
            row = child;
        }
        unsigned mark = 0xFFFFFFFFu;
        if (out_kind == LEAF_EXPAND) {
            int dup = -1;
            for (int i = 0; i < j; i++)
                if (dup < 0 && pend[i] == leaf_edge) dup = i;
            if (dup >= 0) {
                out_kind = LEAF_DUP_BASE + dup;
            } else {
                mark = leaf_edge;
                if (!SYNTH && d.cache) {
                    float cx[G::CPL], ch;
                    const int sym = d.leaf_sym ? leaf_sym_of((int)(d.game_key0 + (unsigned)game), ply, sims_done + j + 1) : 0;
                    const bool hit = cache_lookup<N>(d, me, opp, last, cache_net_key(netid, sym), lane, cx, ch);
                    if (hit) {
                        cache_hit_store<N>(d.logits + it * G::RW, cx, lane, d.leaf_sym != nullptr, sym);
                        d.vhid[it * 64 + lane] = ch;
                        out_kind = LEAF_EXPAND_HIT;
                    }
                    c_look += 1ull;
                    c_hit += hit ? 1ull : 0ull;
                }
            }
        }
        pend[j] = mark;
        if (lane == 0) {
            pl_store(d.leaf + it * 8, me);
            pl_store(d.leaf + it * 8 + 4, opp);
            d.leaf_last[it] = last;
            d.leaf_kind[it] = out_kind;
            d.depth[it] = depth;
            // simulation sims_done + j of the search: its leaf is evaluation sims_done + j + 1 (0 = the root), as in k_step
            if (d.leaf_sym) d.leaf_sym[it] = leaf_sym_of((int)(d.game_key0 + (unsigned)game), ply, sims_done + j + 1);
        }
        wave_mem_sync();
    }
    if (lane == 0) {
        d.rows_used[b] = rows_used;
        unsigned long long *c = d.cnt + (size_t)b * CNT_STRIDE;
        c[0] += c_exp; c[1] += c_sim; c[2] += c_term; c[3] += c_depth; c[5] += c_dup; c[6] += c_look; c[7] += c_hit;
    }
}

// numpy pairwise sum of a contiguous float64 block (n <= 128): np.add.reduce on a contiguous array accumulates in 8 lanes
// over blocks of <= 128 and combines them pairwise; probs.sum() at mcts.py:163 goes through it
__device__ inline double pw_block(const double *a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}

// ------------------------------------------------------------------------------------------------
// k_move: visit counts -> pi -> sampled action (mcts.py:144-177), record (self_play.py:63),
//         apply (games.py:64-82), terminal check (games.py:133-166).
// ------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void k_move(DevState d)
{
    typedef TreeGeo<N> G;
    __shared__ double ebuf_all[4][G::RW];
    __shared__ unsigned short idx_all[4][REUSE_MAX_ROWS];   // subtree reuse: new row index of every kept row
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    if (b >= d.B) return;
    if (d.s_status[b] != SLOT_ACTIVE) return;
    double *ebuf = ebuf_all[wv];
    const int g = d.s_game[b], ply = d.s_ply[b], pl = d.s_player[b];
    Plane X = pl_load(d.board + (size_t)b * 8), O = pl_load(d.board + (size_t)b * 8 + 4);
    Plane occ;
#pragma unroll
    for (int q = 0; q < 4; q++) occ.w[q] = X.w[q] | O.w[q];
    const int A = G::nn - pl_count(occ);
    const Edge *root = d.edges + (size_t)b * d.R * G::RW;
    const double T = d.T_table[d.arena ? ((ply + 1) >> 1) : ply];   // evaluator.py:71-90 step quirk
    const double u = d.u[(size_t)g * G::nn + ply];

    int Nj[G::CPL], rank[G::CPL];
    bool legal[G::CPL];
    double e[G::CPL];
#pragma unroll
    for (int i = 0; i < G::CPL; i++) {
        int j = lane + 64 * i;
        legal[i] = j < G::nn && !pl_get(occ, j);
        Nj[i] = legal[i] ? (int)root[j].N & EDGE_N_MASK : 0;
        rank[i] = legal[i] ? j - pl_rank(occ, j) : 0;
    }
    bool uniform = false;
    double s = 1.0;
    if (T <= 1e-7) {
        // temperature clipped to the Python float 1e-7 -> float32 arithmetic, exp() is exactly 0 or 1
        float y[G::CPL], m = -INFINITY;
#pragma unroll
        for (int i = 0; i < G::CPL; i++) {
            y[i] = legal[i] ? d.log_table[Nj[i]] / 1e-7f : -INFINITY;
            m = fmaxf(m, y[i]);
        }
        m = wave_max_f(m);
        int ties = 0;
#pragma unroll
        for (int i = 0; i < G::CPL; i++) {
            y[i] = legal[i] ? ((y[i] - m) == 0.0f ? 1.0f : az_expf(y[i] - m)) : 0.0f;
            ties += y[i] == 1.0f ? 1 : 0;
        }
        ties = wave_sum_i(ties);
#pragma unroll
        for (int i = 0; i < G::CPL; i++) e[i] = (double)(y[i] / (float)ties);
    } else {
        double m = -INFINITY;
#pragma unroll
        for (int i = 0; i < G::CPL; i++) {
            e[i] = legal[i] ? (double)d.log_table[Nj[i]] / T : -INFINITY;
            m = e[i] > m ? e[i] : m;
        }
        m = wave_max_d(m);
#pragma unroll
        for (int i = 0; i < G::CPL; i++) {
            e[i] = legal[i] ? az_exp(e[i] - m) : 0.0;
            if (legal[i]) ebuf[rank[i]] = e[i];
        }
        wave_mem_sync();
        if (lane == 0) {
            int n2 = A / 2;
            n2 -= n2 % 8;
            s = A <= 128 ? 0.0 + pw_block(ebuf, A) : 0.0 + (pw_block(ebuf, n2) + pw_block(ebuf + n2, A - n2));
        }
        s = __shfl(s, 0, 64);
        uniform = (s < 1e-8) || (s != s);
#pragma unroll
        for (int i = 0; i < G::CPL; i++) e[i] = uniform ? 1.0 / (double)A : e[i] / s;
        wave_mem_sync();
    }
    // RandomState.choice: cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(u, side='right')
#pragma unroll
    for (int i = 0; i < G::CPL; i++)
        if (legal[i]) ebuf[rank[i]] = e[i];
    wave_mem_sync();
    double lastc = 0.0;
    if (lane == 0) {
        double run = ebuf[0];
        for (int i = 1; i < A; i++) { run = run + ebuf[i]; ebuf[i] = run; }
        lastc = run;
    }
    lastc = __shfl(lastc, 0, 64);
    wave_mem_sync();
    int cntle = 0;
    for (int i = lane; i < A; i += 64) cntle += (ebuf[i] / lastc <= u) ? 1 : 0;
    int idx = wave_sum_i(cntle);
    if (idx >= A) idx = A - 1;
    int mycell = -1;
#pragma unroll
    for (int i = 0; i < G::CPL; i++)
        if (legal[i] && rank[i] == idx) mycell = lane + 64 * i;
    u64 bal = __ballot(mycell >= 0);
    int src = __ffsll((long long)bal) - 1;
    const int a = __shfl(mycell, src, 64);

    // ---- record (self_play.py:63): state before the move, pi, mover ----
    const size_t ri = (size_t)g * G::nn + ply;
#pragma unroll
    for (int i = 0; i < G::CPL; i++) {
        int j = lane + 64 * i;
        if (j < G::nn) {
            d.rec_pi[ri * G::nn + j] = legal[i] ? (float)e[i] : 0.0f;
            d.rec_visits[ri * G::nn + j] = (unsigned short)Nj[i];
        }
    }
    // ---- apply + terminal ----
    Plane mine = pl == 1 ? X : O;
    pl_set(mine, a);
    Plane occ2 = occ;
    pl_set(occ2, a);
    bool win = wins_through(mine, a, N, d.k);
    bool full = pl_count(occ2) == G::nn;
    if (lane == 0) {
        u64 *rp = d.rec_planes + ri * 8;
        for (int q = 0; q < 4; q++) {
            rp[q] = pl == 1 ? X.w[q] : O.w[q];
            rp[4 + q] = pl == 1 ? O.w[q] : X.w[q];
        }
        d.rec_last[ri] = (short)d.s_last[b];
        d.rec_mover[ri] = (unsigned char)pl;
        d.rec_action[ri] = (short)a;
        pl_store(d.board + (size_t)b * 8 + (pl == 1 ? 0 : 4), mine);
        d.s_player[b] = 3 - pl;
        d.s_last[b] = a;
        d.s_ply[b] = ply + 1;
        int res = win ? pl : (full ? 3 : 0);
        bool cut = d.max_plies > 0 && ply + 1 >= d.max_plies;
        if (res != 0 || cut) {
            d.g_result[g] = res;
            d.g_nply[g] = ply + 1;
            d.s_status[b] = SLOT_FINISHED;
        }
        d.leaf_kind[(size_t)b * d.L] = LEAF_NONE;
    }
    if (!d.reuse) return;
    // ---- subtree reuse: the chosen child's subtree becomes the next ply's tree ----
    // Rows are created in visiting order, so a child row always has a larger index than its parent: one ascending
    // pass marks the rows reachable from the child, a second one moves them down in place (destination <= source)
    // and renumbers the child links (R <= REUSE_MAX_ROWS, checked by az_set_subtree_reuse).
    {
        const bool cont = !(win || full) && !(d.max_plies > 0 && ply + 1 >= d.max_plies) && !d.arena;
        Edge *rows = d.edges + (size_t)b * d.R * G::RW;
        const int c = cont ? (int)rows[a].child : 0;
        if (c == 0) {
            if (lane == 0) d.carried[b] = -1;
            return;
        }
        unsigned short *idx = idx_all[wv];
        const unsigned short NOT = 0xFFFFu, KEEP = 0xFFFEu;
        const int used = d.rows_used[b];
        wave_mem_sync();
        for (int r = lane; r < used; r += 64) idx[r] = r == c ? KEEP : NOT;
        wave_mem_sync();
        for (int r = c; r < used; r++) {
            if (idx[r] != KEEP) continue;                     // uniform: every lane reads the same LDS word
#pragma unroll
            for (int i = 0; i < G::CPL; i++) {
                const unsigned short ch = rows[(size_t)r * G::RW + lane + 64 * i].child;
                if (ch) idx[ch] = KEEP;
            }
            wave_mem_sync();
        }
        int kept = 0;
        if (lane == 0) {
            for (int r = c; r < used; r++)
                if (idx[r] == KEEP) idx[r] = (unsigned short)kept++;
        }
        wave_mem_sync();
        kept = __shfl(kept, 0, 64);
        int csum = 0;
        for (int r = c; r < used; r++) {
            const unsigned short dst = idx[r];
            if (dst == NOT) continue;
            Edge e[G::CPL];
#pragma unroll
            for (int i = 0; i < G::CPL; i++) {
                e[i] = rows[(size_t)r * G::RW + lane + 64 * i];
                if (e[i].child) e[i].child = idx[e[i].child];
                if (dst == 0) csum += (int)e[i].N;
            }
#pragma unroll
            for (int i = 0; i < G::CPL; i++) rows[(size_t)dst * G::RW + lane + 64 * i] = e[i];
        }
        csum = wave_sum_i(csum);
        if (lane == 0) {
            d.rows_used[b] = kept;
            d.carried[b] = csum;
        }
    }
}

#ifdef AZ_ENGINE_TU
// ------------------------------------------------------------------------------------------------
// k_refill: finished/idle slots receive the next game ids in slot order (ballot + prefix sum), replacing the task
//           queue of self_play.py:114-118,41-45.  The queue is ONE device counter shared by the lanes (streams) of an
//           engine: a chunk of slots claims its ids with a single atomicAdd, so a slot freed in any lane takes the next
//           waiting game.  claim_cap bounds the ids this launch may take (the first refill of an episode spreads the
//           games evenly over the lanes).  One block of 1024 threads.
//           compact != 0: afterwards the active slots are moved to the front of the lane, in order (a slot between plies is
//           its board, game id, ply, side to move and last move; the tree is rebuilt every ply), so that the next ply's
//           kernels can be launched over the active slots only.  A workgroup of a slot without a game would exit at once, but
//           it still needs its LDS allocation to be dispatched and so queues behind the other lanes' working workgroups: in
//           the tail of an episode that head-of-line blocking cost 30 % of a ply.  Invisible in the results (records go by
//           game id).  Not with subtree reuse (the tree outlives the ply).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_refill(DevState d, int claim_cap, int compact)
{
    __shared__ int wsum[16];
    __shared__ int base_s, take_s, cap_s, active_s, cbase_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { cap_s = claim_cap; active_s = 0; }
    __syncthreads();
    for (int c0 = 0; c0 < d.B; c0 += 1024) {
        int b = c0 + tid;
        bool need = b < d.B && d.s_status[b] != SLOT_ACTIVE;
        bool act = b < d.B && d.s_status[b] == SLOT_ACTIVE;
        u64 bal = __ballot(need);
        int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(bal);
        u64 bact = __ballot(act);
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < 16; w++) { off += w < wv ? wsum[w] : 0; tot += wsum[w]; }
        if (tid == 0) {
            int want = min(tot, cap_s), base = 0, got = 0;
            if (want > 0 && *(volatile int *)d.next_game < d.total_games) {     // an exhausted queue is left alone (no overflow)
                base = atomicAdd(d.next_game, want);
                got = max(0, min(want, d.total_games - base));
            }
            base_s = base; take_s = got; cap_s -= want;
        }
        __syncthreads();
        int newact = 0;
        if (need) {
            const int idx = off + pre;
            if (idx < take_s) {
                u64 *bd = d.board + (size_t)b * 8;
                for (int q = 0; q < 8; q++) bd[q] = 0ull;
                const int gid = base_s + idx;
                d.s_game[b] = gid;
                d.s_ply[b] = 0;
                d.s_player[b] = (d.arena && (gid & 1)) ? 2 : 1;   // evaluator.py:64-69
                d.s_last[b] = -1;
                d.s_status[b] = SLOT_ACTIVE;
                d.leaf_kind[(size_t)b * d.L] = LEAF_NONE;
                d.carried[b] = -1;
                newact = 1;
            } else {
                d.s_status[b] = SLOT_IDLE;
                d.s_game[b] = -1;
            }
        }
        u64 bnew = __ballot(newact != 0);
        if (lane == 0) atomicAdd(&active_s, __popcll(bnew) + __popcll(bact));
        __syncthreads();
    }
    if (tid == 0) *d.active = active_s;
    if (!compact || d.reuse) return;
    if (tid == 0) cbase_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < d.B; c0 += 1024) {
        const int b = c0 + tid;
        const bool act = b < d.B && d.s_status[b] == SLOT_ACTIVE;
        u64 bd[8];
        int g = 0, p = 0, pl = 0, la = 0;
        if (act) {
            for (int q = 0; q < 8; q++) bd[q] = d.board[(size_t)b * 8 + q];
            g = d.s_game[b]; p = d.s_ply[b]; pl = d.s_player[b]; la = d.s_last[b];
        }
        const u64 bal = __ballot(act);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();                 // every thread holds its slot in registers; the chunk's counts are known
        int off = 0, tot = 0;
        for (int w = 0; w < 16; w++) { off += w < wv ? wsum[w] : 0; tot += wsum[w]; }
        const int base = cbase_s;
        __syncthreads();
        if (act) {
            const int nb = base + off + pre;             // <= b: only slots already read are overwritten
            if (nb != b) {
                for (int q = 0; q < 8; q++) d.board[(size_t)nb * 8 + q] = bd[q];
                d.s_game[nb] = g; d.s_ply[nb] = p; d.s_player[nb] = pl; d.s_last[nb] = la;
                d.s_status[nb] = SLOT_ACTIVE;
                d.carried[nb] = -1;
            }
        }
        if (tid == 0) cbase_s = base + tot;
        __syncthreads();
    }
    for (int b = cbase_s + tid; b < d.B; b += 1024) { d.s_status[b] = SLOT_IDLE; d.s_game[b] = -1; }
}

// az_rules_replay: Gomoku.apply_action / is_terminal / get_game_result (games.py:64-82,133-179) for whole action lists,
// one thread per game, with the bit-plane helpers the search kernels use (pl_set, wins_through, pl_count).  The winner is
// sticky like the reference's cached `winner` that clone() copies (games.py:140-141,206).
__global__ void k_rules_replay(int n, int k, int games, int max_len, const short *actions, unsigned char *term_before,
                               unsigned char *boards, int *players, int *results, int *status)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= games) return;
    const int nn = n * n;
    Plane X{}, O{};
    int pl = 1, res = 0, bad = -1;
    for (int i = 0; i < max_len; i++) {
        const int a = actions[(size_t)g * max_len + i];
        if (a < 0) break;
        if (term_before) term_before[(size_t)g * max_len + i] = res != 0;
        if (a >= nn || pl_get(X, a) || pl_get(O, a)) { bad = i; break; }       // games.py:76-77 ValueError("Invalid move")
        Plane &mine = pl == 1 ? X : O;
        pl_set(mine, a);
        if (res == 0) {
            if (wins_through(mine, a, n, k)) res = pl;
            else if (pl_count(X) + pl_count(O) == nn) res = 3;
        }
        pl = 3 - pl;
    }
    for (int j = 0; j < nn; j++) boards[(size_t)g * nn + j] = pl_get(X, j) ? 1 : (pl_get(O, j) ? 2 : 0);
    players[g] = pl;
    results[g] = res;
    status[g] = bad;
}

// custom start position for az_search: slot 0 gets the given state
__global__ void k_set_position(DevState d, int slot, int game, int player, int last, int ply)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    d.s_game[slot] = game;
    d.s_ply[slot] = ply;
    d.s_player[slot] = player;
    d.s_last[slot] = last;
    d.s_status[slot] = SLOT_ACTIVE;
    d.leaf_kind[(size_t)slot * d.L] = LEAF_NONE;
    d.carried[slot] = -1;
}
#endif  // AZ_ENGINE_TU
