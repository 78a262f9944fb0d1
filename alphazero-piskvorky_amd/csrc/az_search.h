// az_search.h -- persistent per-workgroup search kernel for small boards: the whole MCTS.run of a ply (mcts.py:101-141)
// in ONE launch, with the tree of every game resident in LDS.
//
// The lock-step pipeline (az_engine.hip launch_ply) spends three dependent dispatches per simulation -- k_trunk, k_fc,
// k_step -- and on small boards the ~5 us between dependent dispatches is a third of a simulation's time.  Here a
// workgroup of 8 waves owns GP games for a whole ply and loops S + 1 times over
//     encode -> conv1 -> conv2 -> conv3 -> 1x1 heads        (all 8 waves, MFMA, activations in LDS like k_trunk)
//     policy_fc / value_fc1                                  (one 16-output tile per wave on the MFMA, the canonical four k-chains)
//     softmax + value tail, expand, backup, PUCT select      (one wavefront per game, like k_step)
// What a phase needs from L2 is asked for before the phase in front of it (the first weight fragments of conv2 / conv3, the FC
// weights) or lives in LDS for the whole ply (the head convs' weights, value_fc2); the images' padding ring is cleared while the
// FC phase waits for its MFMA chains.  tools/stamps_search.py shows the cycles of every phase.
// without leaving the CU: the node array -- one row of n*n 16-byte edges per expanded node, (S + 1) rows per game --
// lives in LDS next to the activations, so a level of the descent is one ds_read_b128 per lane instead of an L2 round
// trip, and nothing but the weights (L2 hits), the root's Dirichlet noise and the final root row crosses the CU boundary.
// Same fma chains and the same operator order as the three kernels it replaces: results are bit-identical (every
// parity test runs on both paths).  Used when the rows fit: GP * (S + 1) * n*n * 16 B beside the activation image,
// i.e. boards up to 7x7 at the reference's simulation counts (5x5: two games per workgroup at S = 100).
#pragma once
#include "az_net.h"

#ifdef AZ_STAMPS
#define KS_PARAM , unsigned long long *st
#define KS_ARG , ks_ph
#else
#define KS_PARAM
#define KS_ARG
#endif
#ifndef AZ_SEARCH_SKIP
#define AZ_SEARCH_SKIP 0      // timing-only experiment builds (results are wrong): 1 = no FC, 2 = no tree step
#endif

template <int N, int GP, bool RES = false>
struct PersistGeo {
    static constexpr int n = N, nn = N * N, PW = N + 2, PP = PW * PW;
    static constexpr int G = GP;                                   // games (boards) per workgroup
    static constexpr int M = G * nn;
    static constexpr bool ROWT = false;
    static constexpr int MT = (M + 15) / 16, MR = MT * 16;
    static constexpr int CS = up16(G * PP);
    // row stride of the conv3 image [co][cell]: 16 mod 32, so that the four k-rows the head convs read with one instruction
    // (channels q, q + 1, q + 2, q + 3 of a group: q x CS3 floats apart) fall into different LDS banks
    static constexpr int CS3 = up16(MR) % 32 == 0 ? up16(MR) + 16 : up16(MR);
    // plain net: conv1 / conv2 outputs (32 + 64 channels) with conv3's [co][cell] image on top; ResidualBlock net: two 64-channel images
    static constexpr int LDSF = RES ? 128 * CS : ((96 * CS > 128 * CS3) ? 96 * CS : 128 * CS3);
    static constexpr int RW = ((nn + 63) / 64) * 64;
    static constexpr int NW = AZ_NW;
    static constexpr int PC = RES ? 2 : 4, VC = RES ? 1 : 2;
    static constexpr int FROW = (((PC + VC) * nn + 3) / 4) * 4;
    static constexpr int PATH = nn + 1;
    static constexpr int ROWE = nn;                                // edges per tree row in LDS (no padding to 64)
};

// LDS-resident search state of one game
struct GameLds {
    u64 leaf[8];           // mover-relative planes of the pending leaf
    int leaf_last, leaf_kind, depth, rows_used;
    int sym;               // symmetry the net sees the pending leaf under (az_set_leaf_symmetry; 0 = as it is)
};

// the tree step of one game on LDS rows: k_step's two stages (az_tree.h) with the node array, the path and the evaluator
// outputs in LDS.  One wavefront.  Subtree reuse: k_search loads the retained rows and writes the whole tree back; an evaluation-cache hit (LEAF_*_HIT, set by k_search before the
// net phase) is consumed exactly like the evaluation it stands for.
template <int N, class PG, bool SYNTH>
__device__ __forceinline__ void step_lds(const DevState &d, int b, int lane, Edge *rows, unsigned *path, GameLds &gs,
                                         const float *lg, const float *hid, const double *sq_lds, int rootN, int do_select,
                                         int pl, int slast, int netid, int game, int ply, const Plane &bX, const Plane &bO,
                                         unsigned long long (&cnt)[4], const float *w2, float b2, int carried KS_PARAM)
{
#ifdef AZ_STAMPS
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
#define ST_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define ST_STAMP(i) do { } while (0)
#endif
    typedef TreeGeo<N> G;
    static_assert(G::CPL == 1, "the LDS tree is for boards of at most 64 cells");
    const int kind_raw = gs.leaf_kind, depth0 = gs.depth, rows0 = gs.rows_used, leaf_last = gs.leaf_last;
    const int kind = kind_raw == LEAF_EXPAND_HIT ? LEAF_EXPAND : (kind_raw == LEAF_ROOT_HIT ? LEAF_ROOT : kind_raw);
    Plane lme, lopp;
#pragma unroll
    for (int q = 0; q < 4; q++) { lme.w[q] = gs.leaf[q]; lopp.w[q] = gs.leaf[4 + q]; }
    // ---------------- stage 1: evaluation -> expand -> backup ----------------
    if (kind == LEAF_REUSE) {
        // subtree reuse: the root kept from the previous ply (its rows were loaded by k_search) is not evaluated again; only the
        // fresh Dirichlet sample is mixed in, with a new root's arithmetic (k_step's branch, az_tree.h)
        if (d.add_noise) {
            Plane occ;
#pragma unroll
            for (int q = 0; q < 4; q++) occ.w[q] = bX.w[q] | bO.w[q];
            const double *nz = d.noise + (size_t)game * d.noise_stride + d.noise_off[ply];
            if (lane < G::nn && !pl_get(occ, lane)) {
                const int rank = lane - pl_rank(occ, lane);
                Edge *e = rows + lane;
                const float scaled = d.one_minus_w * e->P;
                e->P = (float)((double)scaled + d.w_noise * nz[rank]);
            }
        }
        wave_mem_sync();
    } else if (kind != LEAF_NONE) {
        float v = 0.0f;
        if (kind == LEAF_ROOT || kind == LEAF_EXPAND) {
            float P;
            if (SYNTH) {
                unsigned hx = 0;
                if (lane < G::nn) {
                    unsigned code = pl_get(lme, lane) ? 1u : (pl_get(lopp, lane) ? 2u : 0u);
                    hx = az_fmix32((unsigned)lane * 3u + code + 0x9E3779B9u);
                }
                unsigned hs = wave_xor_u(hx);
                hs ^= az_fmix32(0x51ED270Bu + (unsigned)(leaf_last + 1));
                unsigned r = az_fmix32(hs + (unsigned)(lane + 1) * 0x9E3779B1u);
                P = (float)(((r >> 8) & 0xFFFFu) + 1u) * 0x1p-23f;
                int vv = (int)(az_fmix32(hs ^ 0x7F4A7C15u) & 0x1FFu);
                v = (float)(vv - 256) / 256.0f;
            } else {
                // controller.py:49 softmax over all n^2 logits (no legality mask), canonical wave order
                float x = -INFINITY;
                if (lane < G::nn) {
                    if (d.leaf_sym) {       // the net saw the leaf under symmetry t: board cell j sits at image cell sym_src(t^-1, j)
                        const int r = lane / N;
                        x = lg[sym_src(sym_inverse(gs.sym), r, lane - r * N, N)];
                    } else {
                        x = lg[lane];
                    }
                }
                const float mx = wave_max_f(x);
                P = 0.0f;
                float part = 0.0f;
                if (lane < G::nn) {
                    P = az_expf(x - mx);
                    part = part + P;
                }
                const float ssum = wave_sum_butterfly(part);
                P = P / ssum;
                ST_STAMP(10);        // softmax
                // value tail: value_fc2 + tanh (net.py:70), one k-ordered fma chain
                const float acc = value_fc2_chain_lds(hid, w2);    // w2, b2: value_fc2 of the game's net, put into LDS / a register once per ply
                v = az_tanhf(acc + b2);
                ST_STAMP(11);        // value tail
            }
            if (kind == LEAF_ROOT && d.add_noise) {
                // mcts.py:113-116; float32 multiply, float64 add, float32 store
                Plane occ;
#pragma unroll
                for (int q = 0; q < 4; q++) occ.w[q] = lme.w[q] | lopp.w[q];
                const double *nz = d.noise + (size_t)game * d.noise_stride + d.noise_off[ply];
                if (lane < G::nn && !pl_get(occ, lane)) {
                    int rank = lane - pl_rank(occ, lane);
                    float scaled = d.one_minus_w * P;
                    P = (float)((double)scaled + d.w_noise * nz[rank]);
                }
            }
            // mcts.py:50-64 expand: one edge per cell (occupied cells are never selected)
            const int row = kind == LEAF_ROOT ? 0 : rows0;
            if (lane < PG::ROWE) {
                Edge e;
                e.W = 0.0; e.P = P; e.N = 0; e.child = 0;
                rows[row * PG::ROWE + lane] = e;
            }
            if (lane == 0) gs.rows_used = row + 1;
            if (kind == LEAF_EXPAND) {
                const unsigned pe = path[depth0 - 1];
                if (lane == 0) rows[(pe >> 16) * PG::ROWE + (pe & 0xFFFFu)].child = (unsigned short)row;
            }
            ST_STAMP(12);            // root noise, expand
        }
        if (kind != LEAF_ROOT) {
            // mcts.py:132-134,141,76-82: value w.r.t. the side to move at the leaf, backed up with alternating sign
            const double value = kind == LEAF_EXPAND ? (double)v : (kind == LEAF_TERM_LOSS ? -1.0 : 0.0);
            for (int dd = lane; dd < depth0; dd += 64) {
                const unsigned pe = path[dd];
                Edge *e = rows + (pe >> 16) * PG::ROWE + (pe & 0xFFFFu);
                const double val = ((depth0 - 1 - dd) & 1) ? value : -value;
                e->N = (unsigned short)(e->N + 1);
                e->W = e->W + val;
            }
            cnt[0] += kind == LEAF_EXPAND ? 1ull : 0ull;
            cnt[1] += 1ull;
            cnt[2] += kind == LEAF_EXPAND ? 0ull : 1ull;
            cnt[3] += (unsigned long long)depth0;
        }
        wave_mem_sync();
        ST_STAMP(13);                // backup
    }
    // ---------------- stage 2: selection (mcts.py:124-129) ----------------
    if (!do_select || rootN < carried) {          // a retained root tops its visits up to S: idle until simulation `carried`
        if (lane == 0) gs.leaf_kind = LEAF_NONE;
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
        if (lane < G::nn && !pl_get(occ, lane)) {
            Edge e = rows[row * PG::ROWE + lane];
            double Q = e.N ? e.W / (double)e.N : 0.0;         // mcts.py:80 Q = W/N (0.0 while unvisited)
            best = Q + ((d.c_puct * (double)e.P) * sq) / (double)(1 + (int)e.N);
            bi = lane; bN = e.N; bC = e.child;
        }
        ST_STAMP(16);                // one level: row read, PUCT scores
        wave_argmax(best, bi);
        ST_STAMP(17);                // ... argmax
        const int a = __builtin_amdgcn_readfirstlane(bi);
        if (a < 0) { out_kind = LEAF_NONE; break; }
        const int child = __builtin_amdgcn_readlane(bC, a);
        const int an = __builtin_amdgcn_readlane(bN, a);
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
        ST_STAMP(18);                // ... move, terminal tests (of the levels that go on)
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) { gs.leaf[q] = me.w[q]; gs.leaf[4 + q] = opp.w[q]; }
        gs.leaf_last = last;
        gs.leaf_kind = out_kind;
        gs.depth = depth;
        if (d.leaf_sym) gs.sym = leaf_sym_of((int)(d.game_key0 + (unsigned)game), ply, rootN + 1);     // this leaf is evaluation rootN + 1 of the search (as k_step)
    }
    ST_STAMP(14);                    // selection: `depth` levels
#ifdef AZ_STAMPS
    st[15] += (unsigned long long)depth;
#endif
}

// policy_fc (net.py:65) and value_fc1 + ReLU (net.py:69) of the workgroup's GP boards in the canonical order of az_net.h
// (fc_chain_groups): one 16-output tile per wave, its FOUR chains in four independent accumulators (the MFMAs of different
// chains do not wait for each other), combined as ((p0 + p1) + (p2 + p3)) + bias.  Board rows beyond GP are zeros from
// registers; the weight fragments of a tile (4 QG groups) are requested up front from L2; the feature rows are read from
// LDS: featl row = [policy inputs | zeros up to 64 QGP][value inputs | zeros up to 64 QGV], stride NG::FSTR.
template <class NG>
struct FcPre {
    float4 bw[4][NG::QGMAX];       // the wave's tile: weight fragments of its four chains
    float bias;                    // bias of the lane's output
};
// the loads of fc_mfma, issued by the caller a phase early so that their L2 round trip is over when the FC phase starts
template <class PG, class NG>
__device__ __forceinline__ FcPre<NG> fc_prefetch(const NetWeights &w, int wave, int lane)
{
    FcPre<NG> r;
    const int tile = wave < NG::NTP + 4 ? wave : 0;
    const bool is_pol = tile < NG::NTP;
    const int QG = is_pol ? NG::QGP : NG::QGV;
    const float4 *wp4 = reinterpret_cast<const float4 *>(is_pol ? w.pf : w.vf) +
                        (size_t)(is_pol ? tile : tile - NG::NTP) * 4 * QG * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < NG::QGMAX; j++) r.bw[c][j] = j < QG ? wp4[(size_t)(c * QG + j) * 64] : float4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15;
    const int j = tile * 16 + r16, i = (tile - NG::NTP) * 16 + r16;
    r.bias = is_pol ? (j < NG::nn ? w.pfb[j] : 0.0f) : w.vfb[i];
    return r;
}
template <class PG, class NG>
__device__ __forceinline__ void fc_mfma(const FcPre<NG> &pre, const float *featl, float *logits_l, float *vhid_l, int wave, int lane,
                                        unsigned need)      // need: bit g = game g waits for this evaluation (the others' rows are left alone)
{
    const int tile = wave;
    if (tile >= NG::NTP + 4) return;
    const int q = lane >> 4, r16 = lane & 15;
    const bool is_pol = tile < NG::NTP;
    const int QG = is_pol ? NG::QGP : NG::QGV;
    const bool row_ok = r16 < PG::G;
    const float *ip = featl + (row_ok ? r16 : 0) * NG::FSTR + (is_pol ? 0 : NG::VOFFL) + q;
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NG::QGMAX; j++)
        if (j < QG) {
            float a[4][4];
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int e = 0; e < 4; e++) a[c][e] = row_ok ? ip[(c * QG + j) * 16 + e * 4] : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float4 bv = pre.bw[c][j];
                    acc[c] = mfma4(a[c][e], e == 0 ? bv.x : e == 1 ? bv.y : e == 2 ? bv.z : bv.w, acc[c]);
                }
        }
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
        const int g = q * 4 + rg;                      // board row of this accumulator register
        if (g < PG::G && ((need >> g) & 1u)) {
            const float r = (acc[0][rg] + acc[1][rg]) + (acc[2][rg] + acc[3][rg]);
            if (is_pol) {
                const int j = tile * 16 + r16;
                if (j < NG::nn) logits_l[g * PG::RW + j] = r + pre.bias;
            } else {
                const int i = (tile - NG::NTP) * 16 + r16;
                const float v = r + pre.bias;
                vhid_l[g * 64 + i] = v > 0.0f ? v : 0.0f;
            }
        }
    }
}

// One workgroup per CU (the LDS footprint allows no second one), i.e. two waves per SIMD: let the compiler use the
// registers that leaves (256 VGPRs) instead of spilling for an occupancy the kernel can never have.
// TS ("tile subsets", two games per workgroup): when only one of the games waits for the net, compute only the cell tiles that
// hold its cells.  Its own instantiation, chosen when the evaluation cache or subtree reuse is on (then half of the iterations
// look like that): three copies of the trunk made the default path 1.3 % slower, which has few such iterations.
// RES: the ResidualBlock net (stem + 3 blocks on two 64-channel images, 2 + 1 head channels; k_trunk_res): r0 / r1 hold its conv
// weights, w0 / w1 only the FC layers'.  The reference's trained ResidualBlock checkpoints are 5x5 nets.
struct NoWeights { };
template <int N, int GP, bool SYNTH, bool TS = false, bool RES = false>
__global__ __launch_bounds__(AZ_NW * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_search(DevState d, NetWeights w0, NetWeights w1, unsigned long long *dbg,
                                                                                                  std::conditional_t<RES, ResWeights, NoWeights> r0,
                                                                                                  std::conditional_t<RES, ResWeights, NoWeights> r1)
{
    // diagnostic builds only (-DAZ_STAMPS): s_memtime spent in each phase of the loop, summed over the ply, per workgroup
    // (slots 0-9: the loop's phases on waves 0 and 1; 10-18: inside the tree step, see step_lds)
#ifdef AZ_STAMPS
    unsigned long long ks_ph[24];
    for (int i = 0; i < 24; i++) ks_ph[i] = 0ull;
    unsigned long long ks_t = __builtin_amdgcn_s_memtime();
#define KS_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ks_ph[i] += t_ - ks_t; ks_t = t_; } while (0)
#else
#define KS_STAMP(i) do { } while (0)
#endif
    typedef PersistGeo<N, GP, RES> PG;
    typedef std::conditional_t<RES, ResGeo<N>, NetGeo<N>> NG;
    typedef TreeGeo<N> TG;
    constexpr int HK = RES ? 4 : 8;                      // k-groups of 16 channels the head convs run over (64 / 128 input channels)
    constexpr int HC = PG::PC + PG::VC;                  // head channels
    constexpr int NTH = AZ_NW * 64;
    static_assert(NTH == 512, "hd_lds is filled by one float4 per thread");
    __shared__ __attribute__((aligned(16))) float lds[PG::LDSF];
    __shared__ unsigned short wpos[PG::MR];
    __shared__ unsigned short cellof[PG::MR];
    __shared__ __attribute__((aligned(16))) float planes[4 * PG::CS];        // the encoded leaves [mover | opponent | last move | zeros][padded position]
    __shared__ unsigned short cell_at[PG::CS];                               // game-major cell of a padded position, 0xFFFF = padding
    __shared__ __attribute__((aligned(16))) float featl[GP * NG::FSTR];       // head-conv outputs: [policy inputs | zeros][value inputs | zeros] per board (fc_mfma)
    __shared__ float logits_l[GP * PG::RW];
    __shared__ __attribute__((aligned(16))) float vhid_l[GP * 64];
    __shared__ __attribute__((aligned(16))) float4 hd_lds[8 * 64];           // the head convs' weight fragments (read from L2 once per ply)
    __shared__ float hdb_lds[8];
    __shared__ __attribute__((aligned(16))) float w2_lds[GP * 64];           // value_fc2 weights of each game's net
    __shared__ unsigned path_l[GP][PG::PATH];
    __shared__ GameLds games[GP];
    __shared__ double sq_lds[1026];                       // np.sqrt(N + 1e-8), N = 0..S+1 (S <= 1024)
    __shared__ int wg_net;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];     // GP x R x ROWE edges
    Edge *rows_all = reinterpret_cast<Edge *>(dyn_lds);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * GP;
    const int S = d.S;
    for (int i = tid; i < S + 2; i += NTH) sq_lds[i] = d.sqrt_table[i];
    for (int i = tid; i < GP * NG::FSTR; i += NTH) featl[i] = 0.0f;
    for (int i = tid; i < PG::CS; i += NTH) {
        const int g = i / PG::PP, pp = i - g * PG::PP, r = pp / PG::PW - 1, c = pp % PG::PW - 1;
        cell_at[i] = (unsigned short)((g < GP && r >= 0 && r < N && c >= 0 && c < N) ? g * PG::nn + r * N + c : 0xFFFF);
        planes[3 * PG::CS + i] = 0.0f;
    }
    // The padding ring of the two activation images is zero and stays zero: the layers write cells only.  What destroys it is
    // conv3's [co][cell] image on top of them; the workgroup clears that again while the FC phase waits for its MFMA chains.
    {
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < (RES ? PG::LDSF : 96 * PG::CS) / 4; i += NTH) z[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    // tile tables (constant over the ply)
    for (int m = tid; m < PG::MR; m += NTH) {
        const int g = m / PG::nn, p = m - g * PG::nn, r = p / N, c = p - r * N;
        wpos[m] = (unsigned short)(m < PG::M ? g * PG::PP + (r + 1) * PG::PW + (c + 1) : PG::PW + 1);
        cellof[m] = (unsigned short)(m < PG::M ? m : 0xFFFF);
    }
    // per-game constants of the ply in registers of the game's wave (wave g < GP owns game b0 + g)
    const int gb = b0 + (wave < GP ? wave : 0);
    const bool mine = wave < GP && gb < d.B && d.s_status[gb] == SLOT_ACTIVE;
    int pl = 1, slast = -1, netid = 0, game = 0, ply = 0;
    Plane bX{}, bO{};
    if (mine) {
        pl = d.s_player[gb]; slast = d.s_last[gb]; netid = d.s_net[gb]; game = d.s_game[gb]; ply = d.s_ply[gb];
        bX = pl_load(d.board + (size_t)gb * 8); bO = pl_load(d.board + (size_t)gb * 8 + 4);
    }
    // subtree reuse: a root kept from the previous ply arrives as LEAF_REUSE with its compacted rows in HBM (k_reuse)
    const int kind0 = mine ? d.leaf_kind[gb] : LEAF_NONE;
    const int carried = (mine && d.reuse) ? d.carried[gb] : -1;
    const int rows0 = (mine && kind0 == LEAF_REUSE) ? d.rows_used[gb] : 0;
    if (rows0 > 0) {
        const Edge *src = d.edges + (size_t)gb * d.R * TG::RW;
        Edge *dst = rows_all + (size_t)wave * d.R * PG::ROWE;
        for (int r = 0; r < rows0; r++)
            if (lane < PG::ROWE) dst[r * PG::ROWE + lane] = src[(size_t)r * TG::RW + lane];
    }
    if (wave < GP && lane == 0) {
        GameLds &gs = games[wave];
        const int kind = kind0;                                    // k_begin staged the root as the pending leaf
        for (int q = 0; q < 8; q++) gs.leaf[q] = mine ? d.leaf[(size_t)gb * 8 + q] : 0ull;
        gs.leaf_last = mine ? d.leaf_last[gb] : -1;
        gs.sym = (mine && d.leaf_sym) ? d.leaf_sym[gb] : 0;          // k_begin: the root is evaluation 0
        gs.leaf_kind = kind == LEAF_ROOT ? LEAF_ROOT : (kind == LEAF_REUSE ? LEAF_REUSE : LEAF_NONE);
        gs.depth = 0;
        gs.rows_used = rows0;
    }
    if (tid == 0) wg_net = 0;
    __syncthreads();
    if (mine && lane == 0 && netid) atomicOr(&wg_net, 1);          // a workgroup's games share one net (arena: GP = 1)
    __syncthreads();
    if (!SYNTH) {                      // the head convs' weights stay in LDS for the ply (their L2 round trip sat between conv3 and the heads)
        const bool o = wg_net != 0;
        if constexpr (RES) {
            if (tid < HK * 64) hd_lds[tid] = reinterpret_cast<const float4 *>(o ? r1.hd : r0.hd)[tid];
            if (tid < 8) hdb_lds[tid] = tid < HC ? (o ? r1.hdb : r0.hdb)[tid] : 0.0f;
        } else {
            hd_lds[tid] = reinterpret_cast<const float4 *>(o ? w1.hd : w0.hd)[tid];          // 8 k-groups x 64 lanes = the 512 threads
            if (tid < 8) hdb_lds[tid] = tid < HC ? (o ? w1.hdb : w0.hdb)[tid] : 0.0f;
        }
    }
    unsigned long long cnt[4] = {0ull, 0ull, 0ull, 0ull};
    unsigned long long cache_lookups = 0ull, cache_hits = 0ull;
    // value_fc2 of the wave's game does not change over the ply: read once, not behind every evaluation
    const float b2 = (!SYNTH && mine) ? d.v2b[netid][0] : 0.0f;
    if (!SYNTH && mine) w2_lds[wave * 64 + lane] = d.v2w[netid][lane];        // read by the same wave only: no barrier needed beyond the loop's
    float *inA = lds, *inB = lds + 32 * PG::CS;

    KS_STAMP(0);
    for (int idx = 0; idx <= S; idx++) {
        bool any_eval = false;                 // does a game of the workgroup wait for the net?  Every thread works it out for itself.
        unsigned need = 0u;                    // ... and which: bit g = game g
        if (!SYNTH) {
            // opt-in evaluation cache (az_set_eval_cache): the game's wave looks its pending leaf up in the table in HBM; a hit
            // puts the net's outputs for it into the LDS rows and the iteration needs no net for this game -- none at all when
            // every game of the workgroup hits
            if (d.cache) {
                if (mine && leaf_needs_net(games[wave].leaf_kind)) {
                    GameLds &gs = games[wave];
                    Plane lme, lopp;
#pragma unroll
                    for (int q = 0; q < 4; q++) { lme.w[q] = gs.leaf[q]; lopp.w[q] = gs.leaf[4 + q]; }
                    float cx[TG::CPL], ch;
                    const bool hit = cache_lookup<N>(d, lme, lopp, gs.leaf_last, cache_net_key(netid, gs.sym), lane, cx, ch);
                    if (hit) {
                        cache_hit_store<N>(logits_l + wave * PG::RW, cx, lane, d.leaf_sym != nullptr, gs.sym);
                        vhid_l[wave * 64 + lane] = ch;
                        if (lane == 0) gs.leaf_kind = gs.leaf_kind == LEAF_ROOT ? LEAF_ROOT_HIT : LEAF_EXPAND_HIT;
                    }
                    cache_lookups += 1ull;
                    cache_hits += hit ? 1ull : 0ull;
                }
                __syncthreads();
            }
#pragma unroll
            for (int g = 0; g < GP; g++) need |= leaf_needs_net(games[g].leaf_kind) ? 1u << g : 0u;
            any_eval = need != 0u;
        }
        // without a net phase (whose barriers do it otherwise) this barrier keeps a game's wave from replacing its pending leaf
        // while another wave still looks at it
        if (SYNTH || !any_eval) __syncthreads();
        KS_STAMP(1);
        if (!SYNTH && any_eval) {
            NetWeights w;                              // field-wise select: a reference to one of two argument structs would put both into scratch
            {
                const bool o = wg_net != 0;
                w.c1 = o ? w1.c1 : w0.c1; w.c2 = o ? w1.c2 : w0.c2; w.c3 = o ? w1.c3 : w0.c3; w.hd = o ? w1.hd : w0.hd;
                w.pf = o ? w1.pf : w0.pf; w.vf = o ? w1.vf : w0.vf; w.c1b = o ? w1.c1b : w0.c1b; w.c2b = o ? w1.c2b : w0.c2b;
                w.c3b = o ? w1.c3b : w0.c3b; w.hdb = o ? w1.hdb : w0.hdb; w.pfb = o ? w1.pfb : w0.pfb; w.vfb = o ? w1.vfb : w0.vfb;
            }
            ConvPre<32, 4> pre2;
            if constexpr (!RES) pre2 = conv_prefetch<PG, 32, 64>(w.c2, wave, lane);     // conv2's first weights: asked for three phases early
            // ---- leaf encode + conv trunk, as trunk_group (az_net.h) on the LDS-resident leaves ----
            for (int i = tid; i < 3 * PG::CS; i += NTH) {       // games.py:86-129 encode: every position of the three planes, padding = 0
                const int c = i / PG::CS, pos = i - c * PG::CS;
                const int cell = cell_at[pos];
                float v = 0.0f;
                if (cell != 0xFFFF) {
                    const int g = cell / PG::nn, pi = cell - g * PG::nn;
                    const GameLds &gs = games[g];
                    int p = pi;                                  // board cell whose stone the net sees at image cell pi
                    if (d.leaf_sym) { const int r = pi / N; p = sym_src(gs.sym, r, pi - r * N, N); }
                    v = c == 2 ? (gs.leaf_last == p ? 1.0f : 0.0f) : (float)((gs.leaf[4 * c + (p >> 6)] >> (p & 63)) & 1ull);
                }
                planes[i] = v;
            }
            __syncthreads();
            KS_STAMP(2);
            // The conv trunk and the head convs over the cell tiles [mt_base, mt_base + MTL).  Two games share a workgroup, and often
            // only one of them waits for the net (the other's simulation ended on a terminal position, hit the cache, or idles under
            // subtree reuse): then only the tiles that hold its cells are computed -- 2 of 4 for the first game at 5x5, 3 of 4 for
            // the second.  Same chains for every cell that is computed, so nothing changes for the game that is evaluated.
            FcPre<NG> fpre;
            auto trunk = [&](auto mtl_c, const int mt_base) __attribute__((always_inline)) {
                constexpr int MTL = decltype(mtl_c)::value;
                if constexpr (RES) {
                    // stem + three residual blocks on the two 64-channel images, as k_trunk_res (az_net.h); the images' padding ring is
                    // never written, so it needs no clearing between evaluations
                    const bool o = wg_net != 0;
                    float *A = lds, *B = lds + 64 * PG::CS;
                    conv_layer<PG, 4, 64, CONV_OUT_PACKED, 4, MTL>(planes, A, o ? r1.stem : r0.stem, o ? r1.stemb : r0.stemb, wpos, cellof, wave, lane, 0, mt_base, MTL);
                    __syncthreads();
                    KS_STAMP(3);
#pragma unroll 1
                    for (int blk = 0; blk < 3; blk++) {        // (asking for a conv's first weights a layer early measured equal here: not done)
                        conv_layer<PG, 64, 64, CONV_OUT_PACKED, 4, MTL>(A, B, o ? r1.blk[2 * blk] : r0.blk[2 * blk], o ? r1.blkb[2 * blk] : r0.blkb[2 * blk], wpos,
                                                                        cellof, wave, lane, 0, mt_base, MTL);
                        __syncthreads();
                        conv_layer<PG, 64, 64, CONV_OUT_RESIDUAL, 4, MTL>(B, A, o ? r1.blk[2 * blk + 1] : r0.blk[2 * blk + 1],
                                                                          o ? r1.blkb[2 * blk + 1] : r0.blkb[2 * blk + 1], wpos, cellof, wave, lane, 0, mt_base, MTL);
                        __syncthreads();
                    }
                    KS_STAMP(5);
                } else {
                    conv_layer<PG, 4, 32, CONV_OUT_PACKED, 2, MTL>(planes, inA, w.c1, w.c1b, wpos, cellof, wave, lane, 0, mt_base, MTL);
                    __syncthreads();
                    KS_STAMP(3);
                    const ConvPre<64, 8> pre3 = conv_prefetch<PG, 64, 128>(w.c3, wave, lane);    // ... conv3's while conv2 runs
                    conv_layer<PG, 32, 64, CONV_OUT_PACKED, 4, MTL>(inA, inB, w.c2, w.c2b, wpos, cellof, wave, lane, 0, mt_base, MTL, 0, nullptr, nullptr, &pre2);
                    __syncthreads();
                    KS_STAMP(4);
                    conv_layer<PG, 64, 128, CONV_OUT3, 8, MTL>(inB, lds, w.c3, w.c3b, wpos, cellof, wave, lane, 0, mt_base, MTL, 0, nullptr, nullptr, &pre3);
                    __syncthreads();
                    KS_STAMP(5);
                }
                fpre = fc_prefetch<PG, NG>(w, wave, lane);      // the FC weights are on their way during the head convs
                // policy_conv and value_conv, 1x1 (plain: 128 -> 4 + 2 over conv3's [co][cell] image; ResidualBlock: 64 -> 2 + 1 over the
                // packed trunk image), into the LDS feature rows
                const int q = lane >> 4, r16 = lane & 15;
                const float4 *wp4 = hd_lds + lane;
                float hb[4];
#pragma unroll
                for (int rg = 0; rg < 4; rg++) hb[rg] = (q * 4 + rg) < HC ? hdb_lds[q * 4 + rg] : 0.0f;
                for (int lt = wave; lt < MTL; lt += AZ_NW) {
                    const int mt = mt_base + lt;
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (RES) {
                        const float4 *in4 = reinterpret_cast<const float4 *>(lds);
                        const int base = q * PG::CS + (int)wpos[mt * 16 + r16];
#pragma unroll
                        for (int cg = 0; cg < HK; cg++) {
                            const float4 a = in4[base + cg * 4 * PG::CS];
                            const float4 wq = wp4[cg * 64];
                            acc = mfma4(wq.x, a.x, acc);
                            acc = mfma4(wq.y, a.y, acc);
                            acc = mfma4(wq.z, a.z, acc);
                            acc = mfma4(wq.w, a.w, acc);
                        }
                    } else {
                        const float *ip = lds + q * PG::CS3 + (MTL < PG::MT ? lt : mt) * 16 + r16;     // a subset's conv3 image counts its columns from its first tile
#pragma unroll
                        for (int s4 = 0; s4 < HK; s4++) {
                            const float4 bq = wp4[s4 * 64];
                            acc = mfma4(bq.x, ip[(s4 * 16 + 0) * PG::CS3], acc);
                            acc = mfma4(bq.y, ip[(s4 * 16 + 4) * PG::CS3], acc);
                            acc = mfma4(bq.z, ip[(s4 * 16 + 8) * PG::CS3], acc);
                            acc = mfma4(bq.w, ip[(s4 * 16 + 12) * PG::CS3], acc);
                        }
                    }
                    const int cell = cellof[mt * 16 + r16];
                    if (cell != 0xFFFF) {
                        const int g = cell / PG::nn, p = cell - g * PG::nn;
#pragma unroll
                        for (int rg = 0; rg < 4; rg++) {
                            const int j = q * 4 + rg;     // head channel: policy_conv's, then value_conv's (net.py:64,69 flatten order)
                            if (j < HC) {
                                float v = acc[rg] + hb[rg];
                                featl[g * NG::FSTR + (j < PG::PC ? j * PG::nn : NG::VOFFL + (j - PG::PC) * PG::nn) + p] = v > 0.0f ? v : 0.0f;
                            }
                        }
                    }
                }
            };
            constexpr int TILES_A = (PG::nn + 15) / 16, BASE_B = PG::nn / 16, TILES_B = PG::MT - BASE_B;     // tiles of game 0 / of game 1
            if constexpr (TS && GP == 2 && TILES_A < PG::MT && TILES_B < PG::MT) {
                if (need == 1u) trunk(std::integral_constant<int, TILES_A>{}, 0);
                else if (need == 2u) trunk(std::integral_constant<int, TILES_B>{}, BASE_B);
                else trunk(std::integral_constant<int, PG::MT>{}, 0);
            } else {
                trunk(std::integral_constant<int, PG::MT>{}, 0);
            }
            __syncthreads();
            KS_STAMP(6);
            if constexpr (!RES) {
                float4 *z = reinterpret_cast<float4 *>(lds);
                for (int i = tid; i < (96 * PG::CS) / 4; i += NTH) z[i] = float4{0.f, 0.f, 0.f, 0.f};
            }
            if (!(AZ_SEARCH_SKIP & 1)) fc_mfma<PG, NG>(fpre, featl, logits_l, vhid_l, wave, lane, need);
            __syncthreads();
            KS_STAMP(7);
            // a fresh evaluation is remembered (the rows of a game that hit are the cache's: the FC layers leave them alone)
            if (d.cache && mine && leaf_needs_net(games[wave].leaf_kind)) {
                const GameLds &gs = games[wave];
                Plane lme, lopp;
#pragma unroll
                for (int q = 0; q < 4; q++) { lme.w[q] = gs.leaf[q]; lopp.w[q] = gs.leaf[4 + q]; }
                float cx[TG::CPL];
#pragma unroll
                for (int i = 0; i < TG::CPL; i++) {         // entries hold the logits in board order, as k_step's
                    const int j = lane + 64 * i, r = j / N;
                    cx[i] = j < TG::nn ? logits_l[wave * PG::RW + (d.leaf_sym ? sym_src(sym_inverse(gs.sym), r, j - r * N, N) : j)] : 0.0f;
                }
                cache_insert<N>(d, lme, lopp, gs.leaf_last, cache_net_key(netid, gs.sym), lane, cx, vhid_l[wave * 64 + lane]);
            }
        }
        // ---- tree step: wave g works on game g ----
        if (mine && (!(AZ_SEARCH_SKIP & 2) || idx == S))
            step_lds<N, PG, SYNTH>(d, gb, lane, rows_all + (size_t)wave * d.R * PG::ROWE, path_l[wave], games[wave],
                                   logits_l + wave * PG::RW, vhid_l + wave * 64, sq_lds, idx, idx < S ? 1 : 0, pl, slast, netid,
                                   game, ply, bX, bO, cnt, w2_lds + wave * 64, b2, carried KS_ARG);
        KS_STAMP(8);                 // the tree step of this wave's game (waves 0 .. GP - 1)
        __syncthreads();
        KS_STAMP(9);                 // ... and the wait for the other game's
    }
    // ---- hand the root row to k_move (visit counts -> pi -> move), counters to the host ----
    if (mine) {
        const Edge *root = rows_all + (size_t)wave * d.R * PG::ROWE;
        Edge *out = d.edges + (size_t)gb * d.R * TG::RW;
        const int nrows = d.reuse ? games[wave].rows_used : 1;     // subtree reuse: k_reuse picks the chosen child's subtree out of the whole tree
        for (int r = 0; r < nrows; r++)
            if (lane < PG::ROWE) out[(size_t)r * TG::RW + lane] = root[r * PG::ROWE + lane];
        if (lane == 0) {
            d.rows_used[gb] = games[wave].rows_used;
            d.leaf_kind[gb] = LEAF_NONE;
            unsigned long long *c = d.cnt + (size_t)gb * CNT_STRIDE;
            c[0] += cnt[0]; c[1] += cnt[1]; c[2] += cnt[2]; c[3] += cnt[3];
            c[4] += kind0 == LEAF_REUSE ? 1ull : 0ull;
            c[6] += cache_lookups; c[7] += cache_hits;
        }
    }
#ifdef AZ_STAMPS
    if (dbg && lane == 0 && wave < 2)
        for (int i = 0; i < 24; i++) dbg[((size_t)blockIdx.x * 2 + wave) * 32 + i] = ks_ph[i];
#endif
}
