// az_net.h -- the policy/value net forward as HIP kernels for gfx950.
//
//  k_trunk     : GomokuNet (net.py:55-72): leaf encode (games.py:86-129) -> conv1 -> conv2 -> conv3 -> policy/value
//                1x1 convs.  One workgroup per group of G boards, every activation resident in LDS: zero-padded
//                images, so the 3x3 im2col is "base + constant offset"; conv2/conv3 inputs in a packed layout read
//                with ds_read_b128; contraction on v_mfma_f32_16x16x4_f32 with the weights as the A operand.
//  k_trunk_res : the ResidualBlock variant (BASELINE config 5): stem + 3 residual blocks + 1x1 heads, same machinery.
//  k_fc        : policy_fc and value_fc1 as a batched GEMM over boards, same MFMA, one k-ordered chain per output.
// softmax / value_fc2 / tanh are fused into the tree kernel that consumes them (az_tree.h).
//
// Numerics: every output element is ONE k-ordered fp32 fma chain from +0 (what the f32 MFMA computes,
// bit for bit), bias added afterwards.  conv k = (ky*3+kx)*Cin + ci; FC k ascending.
#pragma once
#include "az_tree.h"
#include <type_traits>

#ifdef AZ_STAMPS
// diagnostic build only: per-workgroup phase time stamps (s_memtime) into a buffer no other code reads
// slots 0..5: s_memtime (shader clock ticks) at the phase boundaries; slots 14 / 15: s_memrealtime (the constant 100 MHz counter)
// at the first and the last stamp, so that the clock the chip HELD inside the kernel is (t5 - t0) / (rt5 - rt0) x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6)
#define AZ_STAMP(k) do { if (threadIdx.x == 0 && dbg) { dbg[(size_t)grp * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    if ((k) == 0) dbg[(size_t)grp * 16 + 14] = __builtin_amdgcn_s_memrealtime(); \
    if ((k) == 5) dbg[(size_t)grp * 16 + 15] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define AZ_STAMP(k) do { } while (0)
#endif

#ifndef AZ_NW
#define AZ_NW 8          // waves per trunk workgroup
#endif
#ifndef AZ_SEQ
#define AZ_SEQ 1         // board groups per trunk workgroup (0 = persistent grid stride); all variants measure the same
#endif
#ifndef AZ_PRIO_ALT
#define AZ_PRIO_ALT 2    // the waves sharing a SIMD take turns with the issue priority in the conv main loops: 2 = per tap (default), 1 = per
                         // 16-channel group (the same speed, two VGPRs over the budget at n = 15), 0 = off
#endif
#ifndef AZ_UNROLL_TAPS
#define AZ_UNROLL_TAPS 1     // the conv main loops with all nine taps unrolled (constant fragment offsets, no per-tap address arithmetic)
#endif
#ifndef AZ_NTW
#define AZ_NTW 1         // channel tiles per wave in the conv layers (1: 15 cell tiles per wave, no surplus tile)
#endif

struct NetWeights {
    // MFMA-fragment packed (see pack_* in az_engine.hip): [ntile][kstep/4][lane][4]
    const float *c1, *c2, *c3, *hd, *pf, *vf;
    const float *c1b, *c2b, *c3b, *hdb, *pfb, *vfb;   // biases (hdb: 4 policy_conv + 2 value_conv)
    const void *c1x[2], *c2x[2], *c3x[2], *hdx[2];    // the convs and head convs split into 16-bit fragments (az_net_emul.h), [0] bf16x3, [1] f16x2
};

// Canonical order of an FC output over K inputs (oracle: fc_dot): the inputs are taken in groups of 16 (four k-steps of
// v_mfma_f32_16x16x4_f32) and cut into FOUR contiguous blocks of fc_chain_groups(K) groups; each block is one k-ordered fma
// chain from +0, and the output is ((p0 + p1) + (p2 + p3)) + bias.  Four chains instead of one so that an output tile's
// contraction spreads over four wavefronts (k_fc) or four independent accumulators (k_search) -- a 15x15 policy output is
// 4 x 60 dependent MFMAs instead of 228.
__host__ __device__ constexpr int fc_chain_groups(int K) { return (((K + 15) / 16) + 3) / 4; }

__host__ __device__ constexpr int up16(int x) { return x + ((16 - (x % 32) + 32) % 32); }   // smallest y >= x, y == 16 (mod 32)

// Boards per trunk workgroup for an n x n board: as many as fit the LDS budget, at most 4 (so that 1024 games still
// give >= 256 workgroups).  chan = LDS floats per padded cell (96 plain net: 32+64 channels; 128 ResidualBlock net).
__host__ __device__ constexpr int trunk_lds_floats(int n, int g, int chan)
{
    const int cs = up16(g * (n + 2) * (n + 2));
    const int mr = ((g * n * n + 15) / 16) * 16;
    const int a = chan * cs, b = chan == 96 ? 128 * up16(mr) : 0;     // plain net: the conv3 image overlays the inputs
    return a > b ? a : b;
}
#ifndef AZ_G_BUDGET
#define AZ_G_BUDGET 36500      // LDS floats the packed images of one trunk workgroup may take (146 KB)
#endif
__host__ __device__ constexpr int pick_boards(int n, int chan, int budget)
{
    // powers of two only: 1024 x k games then split into whole rounds of 256 workgroups (3 boards per workgroup left
    // the last round of a 4096-game 9x9 launch one-third full)
    int g = 1;
    for (int t = 2; t <= 4; t *= 2)
        if (trunk_lds_floats(n, t, chan) <= budget) g = t;
    return g;
}

template <int N>
struct NetGeo {
    static constexpr int n = N, nn = N * N, PW = N + 2, PP = PW * PW;
    static constexpr int G = pick_boards(N, 96, AZ_G_BUDGET);            // boards per workgroup: 1 at n >= 12, 2 at 8-11, 4 below
    static constexpr int M = G * nn;                               // real GEMM columns (board cells)
    // 16-cell MFMA tiles.  n = 15: one tile = one board row + its right padding cell (contiguous in the padded
    // image, so the 16 lanes of a fragment hit 16 consecutive LDS banks); other sizes: 16 consecutive cells.
    static constexpr bool ROWT = (N == 15);
    static constexpr int MT = ROWT ? G * N : (M + 15) / 16, MR = MT * 16;
    static constexpr int CS = up16(G * PP);                        // channel stride of padded planes (floats), == 16 mod 32
    static constexpr int CS3 = up16(MR);                           // channel stride of the conv3 output image
    static constexpr int LDSF = (96 * CS > 128 * CS3) ? 96 * CS : 128 * CS3;
    static constexpr int RW = ((nn + 63) / 64) * 64;
    static constexpr int NW = AZ_NW;                               // waves per trunk workgroup
    // FC kernel
    static constexpr int PC = 4, VC = 2;                           // policy_conv / value_conv output channels (net.py:46,52)
    static constexpr int NTP = (nn + 15) / 16;                     // policy N-tiles
    static constexpr int FROW = (((PC + VC) * nn + 3) / 4) * 4;    // feature row in HBM: [0,PC*nn) policy, [PC*nn,(PC+VC)*nn) value, zero tail
    // FC layers (fc_chains below): K inputs = groups of 16 (four MFMA k-steps), cut into 4 chains of QG groups each
    static constexpr int QGP = fc_chain_groups(PC * nn), QGV = fc_chain_groups(VC * nn);
    static constexpr int QGMAX = QGP > QGV ? QGP : QGV;
        // LDS feature rows of the persistent search kernel: policy inputs, then value inputs, each padded to whole chains (zeros)
    static constexpr int VOFFL = 64 * QGP;
    static constexpr int FSTR0 = 64 * (QGP + QGV);
    static constexpr int FSTR = FSTR0 + ((4 - (FSTR0 % 32) + 32) % 32);   // LDS row stride, == 4 (mod 32), multiple of 4
};


// Geometry of the ResidualBlock variant (BASELINE config 5; topology from the reference's historical checkpoints,
// SURVEY.md §8c): 64-channel stem + 3 residual blocks, 2/1-channel heads.  Two 64-channel packed images live in LDS.
template <int N>
struct ResGeo {
    static constexpr int n = N, nn = N * N, PW = N + 2, PP = PW * PW;
    static constexpr int G = pick_boards(N, 128, 39500);           // boards per workgroup (LDS: 2 x 64 channels)
    static constexpr int M = G * nn;
    static constexpr bool ROWT = (N == 15);
    static constexpr int MT = ROWT ? G * N : (M + 15) / 16, MR = MT * 16;
    static constexpr int CS = up16(G * PP);
    static constexpr int CS3 = CS;                                 // unused (no conv3-style overlay)
    static constexpr int LDSF = 128 * CS;
    static constexpr int RW = ((nn + 63) / 64) * 64;
    static constexpr int NW = N == 15 ? 12 : 8;                    // 4 channel tiles x {3,2} cell-tile groups: 15 = 3 x 5 tiles, no surplus
    static constexpr int PC = 2, VC = 1;
    static constexpr int NTP = (nn + 15) / 16;
    static constexpr int FROW = (((PC + VC) * nn + 3) / 4) * 4;
    static constexpr int QGP = fc_chain_groups(PC * nn), QGV = fc_chain_groups(VC * nn);
    static constexpr int QGMAX = QGP > QGV ? QGP : QGV;
    // LDS feature rows of the persistent search kernel (as NetGeo)
    static constexpr int VOFFL = 64 * QGP;
    static constexpr int FSTR0 = 64 * (QGP + QGV);
    static constexpr int FSTR = FSTR0 + ((4 - (FSTR0 % 32) + 32) % 32);
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// LDS activation image of a conv input with C >= 16 channels ("packed"): channel ci = 16*cg + 4*e + q lives at
//   float index  (((cg*4 + q) * CS + pos) * 4 + e)
// so that ONE ds_read_b128 at (cg, q, pos) delivers the B-operand values of four consecutive k-steps
// (e = 0..3) of lane group q, and the 16 lanes of a group read 256 contiguous bytes (conflict-free; CS % 16 == 0).
template <class GEO>
__device__ __forceinline__ int pk_index(int ci, int pos)
{
    return ((((ci >> 4) * 4 + (ci & 3)) * GEO::CS + pos) << 2) + ((ci >> 2) & 3);
}
enum { CONV_OUT_PACKED = 0, CONV_OUT3 = 1, CONV_OUT_RESIDUAL = 2,    // RESIDUAL: relu(acc + bias + out[same index]) in place
       CONV_OUT3_RESIDUAL = 3,     // relu(acc + bias + res[packed index]) into the [co][cell] tile (the last block of a tile-split ResidualBlock trunk)
       CONV_OUT3_PK = 4 };         // conv3 of the fused trunk: the output packed by tile cell, float index ((cg*4 + q)*MR + m)*4 + e for channel
                                   // 16 cg + 4 e + q -- no padding ring (a 1x1 conv reads it), so that the heads' B operands of four k-steps are ONE ds_read_b128

// One conv layer on the workgroup's LDS image, computed as D[co][cell] = sum_k W[co][k] * X[k][cell]:
// the weight fragment is the MFMA A operand (row = output channel), the activation fragment the B operand
// (column = board cell), so an accumulator register holds 16 consecutive cells of one channel per 16 lanes.
// A wave owns NTW channel tiles x MTW cell tiles; every activation fragment read from LDS feeds NTW MFMAs.
// CIN == 4 (conv1) reads plain planes [ci][pos]; CIN >= 32 reads the packed image above.
// OUT3=false: relu(acc+bias) -> packed image of the next layer.  OUT3=true (conv3): barrier, then the
// [co][cell] image that overlays the (now dead) inputs.
// NTL / nt_base: the split (low-latency) kernels give one workgroup only NTL of the layer's channel tiles, starting
// at tile nt_base; the fused kernels use all of them.
// MTL / mt_base / mt_cnt: the tile-split kernels (k_tile) give one workgroup only the cell tiles mt_base .. mt_base + mt_cnt - 1
// (mt_cnt <= MTL); GO / out_pos_off: their packed output goes to an image of another geometry (the full board image in
// global memory) at positions shifted by out_pos_off; O3S: row stride of the [co][cell] output of CONV_OUT3, whose columns
// are counted from the workgroup's first tile.  The fused kernels use the defaults (all tiles, same geometry).
// The first weight fragments of a layer (the ones conv_layer asks for at its top), requested by the caller ahead of the layer.
// The persistent search kernel of the small boards does that one layer early: there a layer lasts 0.3-8 us, and the L2 round
// trip in front of its first MFMA was a visible part of it.  Same registers, same values: results cannot change.
template <int CIN, int NT>
struct ConvPre {
    static constexpr int NTW = (AZ_NTW <= NT) ? AZ_NTW : NT;
    static constexpr int NF = CIN == 4 ? 3 : CIN / 16;       // float4 fragments per channel tile: conv1's whole weights, one tap's otherwise
    float4 bw[NTW][NF];
};
template <class G, int CIN, int COUT, int NTL = COUT / 16>
__device__ __forceinline__ ConvPre<CIN, NTL> conv_prefetch(const float *__restrict__ wp, int wave, int lane, int nt_base = 0)
{
    typedef ConvPre<CIN, NTL> P;
    constexpr int NG = NTL / P::NTW;
    constexpr int KS4 = (9 * (CIN / 4) + 3) / 4;
    const int ng = wave % NG;
    P r;
#pragma unroll
    for (int t = 0; t < P::NTW; t++) {
        const float4 *w4 = reinterpret_cast<const float4 *>(wp) + (size_t)(nt_base + ng * P::NTW + t) * KS4 * 64 + lane;
#pragma unroll
        for (int j = 0; j < P::NF; j++) r.bw[t][j] = w4[(size_t)j * 64];
    }
    return r;
}

template <class G, int CIN, int COUT, int MODE, int NTL = COUT / 16, int MTL = G::MT, class GO = G, int O3S = G::CS3>
__device__ __forceinline__ void conv_layer(const float *in, float *out, const float *__restrict__ wp,
                                           const float *__restrict__ bias, const unsigned short *wpos,
                                           const unsigned short *cellof, int wave, int lane, int nt_base = 0,
                                           int mt_base = 0, int mt_cnt = MTL, int out_pos_off = 0, const float *res = nullptr,
                                           unsigned long long *lst = nullptr, const ConvPre<CIN, NTL> *pre = nullptr)
{
    // diagnostic builds only (-DAZ_STAMPS): per-wave time stamps of a layer's phases into lst[wave * 4 + k]
#ifdef AZ_STAMPS
#define AZ_LSTAMP(k) do { if (lst && lane == 0) lst[wave * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AZ_LSTAMP(k) do { } while (0)
#endif
    constexpr bool OUT3 = MODE == CONV_OUT3 || MODE == CONV_OUT3_RESIDUAL || MODE == CONV_OUT3_PK;
    constexpr bool SUBSET = MTL < G::MT;                       // a tile-split kernel: waves without a tile skip the layer
    constexpr int NT = NTL;                                    // channel tiles handled by this workgroup
    constexpr int NTW = (AZ_NTW <= NT) ? AZ_NTW : NT;          // channel tiles per wave
    constexpr int NG = NT / NTW;                               // channel-tile groups
    constexpr int MG = (G::NW / NG) > 0 ? (G::NW / NG) : 1;    // cell-tile groups
    // cell tiles per wave: the first MTL % MG cell-tile groups take one more than the others.  The layer's body is
    // instantiated for both counts and a wave runs the one that is its own (until round 3 every wave ran the larger count and
    // threw the surplus tile away: 16 instead of 15 tile chains per SIMD in conv2 at n = 15).
    constexpr int MT_FULL = MTL / MG, MT_REM = MTL % MG;
    constexpr int KST = CIN / 4;           // k-steps per tap
    constexpr int KS = 9 * KST;
    constexpr int KS4 = (KS + 3) / 4;
    const int ng = wave % NG, mg = wave / NG;       // wave is wave-uniform (readfirstlane) -> scalar control flow
    const int q = lane >> 4, r16 = lane & 15;

    auto body = [&](auto mtw_c) __attribute__((always_inline)) {
    constexpr int MTW = decltype(mtw_c)::value;
    if constexpr (MTW == 0) {
        if constexpr (OUT3) __syncthreads();       // a wave without a tile still meets the layer's barrier
    } else {
    AZ_LSTAMP(0);
    f32x4 acc[NTW][MTW];
#pragma unroll
    for (int t = 0; t < NTW; t++)
#pragma unroll
        for (int i = 0; i < MTW; i++) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float4 *wp4[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) wp4[t] = reinterpret_cast<const float4 *>(wp) + (size_t)(nt_base + ng * NTW + t) * KS4 * 64 + lane;
    // the epilogue's biases are requested before the main loop, so that their L2 round trip is not paid between two layers
    float bias_pre[NTW][4];
#pragma unroll
    for (int t = 0; t < NTW; t++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bias_pre[t][rg] = bias[(nt_base + ng * NTW + t) * 16 + q * 4 + rg];
    if (!SUBSET || mg < mt_cnt) {
    if constexpr (CIN == 4) {
        // conv1: 9 k-steps (one per tap), channels = {mover, opponent, last move, zero plane}, planes [ci][pos]
        int rb[MTW];
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            int mt = mg + i * MG;
            int m = (mt_base + (mt < mt_cnt ? mt : 0)) * 16 + r16;       // a surplus tile aliases the first one (computed, never written back)
            rb[i] = (int)wpos[m] - (G::PW + 1) + q * G::CS;
        }
        float bk[NTW][12];
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            float4 b0, b1, b2;
            if (pre) { b0 = pre->bw[t][0]; b1 = pre->bw[t][1]; b2 = pre->bw[t][2]; }
            else { b0 = wp4[t][0]; b1 = wp4[t][64]; b2 = wp4[t][128]; }
            const float tmp[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
            for (int j = 0; j < 12; j++) bk[t][j] = tmp[j];
        }
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int toff = (tap / 3) * G::PW + (tap % 3);
#pragma unroll
            for (int i = 0; i < MTW; i++) {
                const float a = in[rb[i] + toff];
#pragma unroll
                for (int t = 0; t < NTW; t++) acc[t][i] = mfma4(bk[t][tap], a, acc[t][i]);
            }
        }
    } else {
        constexpr int NQ = KST / 4;        // weight groups (= 16-channel groups) per tap: 2 for conv2, 4 for conv3
        // Software pipeline pinned with sched_group_barrier: while the 4*MTW*NTW MFMAs of group g issue, the MTW
        // ds_read_b128 of group g+1 are interleaved between them; the weight fragments of tap t+1 are fetched (L2)
        // at the top of tap t.
        const float4 *in4 = reinterpret_cast<const float4 *>(in);
        int ra[MTW];                       // running float4 index: (cg*4 + q)*CS + top-left position of the window
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            int mt = mg + i * MG;
            int m = (mt_base + (mt < mt_cnt ? mt : 0)) * 16 + r16;
            ra[i] = (int)wpos[m] - (G::PW + 1) + q * G::CS;
        }
        float4 a0[MTW], a1[MTW];
        float4 bw[NTW][NQ], bnx[NTW][NQ];
#if AZ_PRIO_ALT
        const int prio_turn = (wave >> 2) % (G::NW / 4);        // scalar: which of the waves sharing a SIMD this one is
#endif
#pragma unroll
        for (int t = 0; t < NTW; t++)
#pragma unroll
            for (int j = 0; j < NQ; j++) bw[t][j] = pre ? pre->bw[t][j] : wp4[t][(size_t)j * 64];
#pragma unroll
        for (int i = 0; i < MTW; i++) a0[i] = in4[ra[i]];
        // AZ_UNROLL_TAPS: the nine taps unrolled, so that every fragment address is the tile's base register + a constant (the
        // ds_read's 16-bit immediate: at most ((NQ - 1) 4 CS + 2 PW + 2) x 16 B = 58.9 KB at n = 15) and the weight registers
        // of consecutive taps are renamed instead of copied.  The rolled loop kept a running index per tile and recomputed
        // the bases every tap: 0.82 vector instructions per MFMA in conv3 -- and the float32 MFMA leaves the SIMD's vector
        // issue so little room that those showed one for one: 35.2 cycles per MFMA instead of 32 (measured with the LDS
        // reads AND the weight fetches taken out: still 35).
#if AZ_UNROLL_TAPS
#pragma unroll
#endif
        for (int tap = 0; tap < 9; tap++) {
            const int tn = tap + 1 < 9 ? tap + 1 : tap;
#pragma unroll
            for (int t = 0; t < NTW; t++)
#pragma unroll
                for (int j = 0; j < NQ; j++) bnx[t][j] = wp4[t][(size_t)(tn * NQ + j) * 64];
#pragma unroll
            for (int sq = 0; sq < NQ; sq++) {
#if AZ_PRIO_ALT
                // The waves that share a SIMD (w, w + 4, ...) run the same program; left alone, one of them wins the issue
                // arbitration every time, finishes the layer early and leaves its partner to run the rest alone -- and ONE
                // wave issues an MFMA only every ~44 cycles (measured: conv3 at n = 15, waves 4-7 done after 102.6 k cycles,
                // waves 0-3 after 151.6 k; the matrix pipe needs 138.2 k).  Taking turns with the priority keeps them
                // level, so that the pipe has two streams to draw from until the end.
                if (AZ_PRIO_ALT == 1 || sq == 0) {      // 1: turns change every 16-channel group, 2: every tap
                    if (((AZ_PRIO_ALT == 1 ? tap * NQ + sq : tap) % (G::NW / 4)) == prio_turn) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
#endif
                float4 *cur = (sq & 1) ? a1 : a0;
                float4 *nxt = (sq & 1) ? a0 : a1;
                // float4 offset of the NEXT group's fragments from the tile's base: next 16-channel group of this tap, or the
                // first group of the next tap (the last group of all reads its own again: never used)
                const int gtap = sq + 1 < NQ ? tap : tn, gsq = sq + 1 < NQ ? sq + 1 : (tap + 1 < 9 ? 0 : sq);
                const int noff = gsq * 4 * G::CS + (gtap / 3) * G::PW + (gtap % 3);
#pragma unroll
                for (int i = 0; i < MTW; i++) nxt[i] = in4[ra[i] + noff];
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int i = 0; i < MTW; i++)
#pragma unroll
                        for (int t = 0; t < NTW; t++) {
                            const float4 wv = bw[t][sq];
                            const float we = e == 0 ? wv.x : e == 1 ? wv.y : e == 2 ? wv.z : wv.w;
                            const float ae = e == 0 ? cur[i].x : e == 1 ? cur[i].y : e == 2 ? cur[i].z : cur[i].w;
                            acc[t][i] = mfma4(we, ae, acc[t][i]);
                        }
#pragma unroll
                for (int i = 0; i < MTW; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NTW, 0);   // 4*NTW MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // 1 LDS read (b128)
                }
            }
#pragma unroll
            for (int t = 0; t < NTW; t++)
#pragma unroll
                for (int j = 0; j < NQ; j++) bw[t][j] = bnx[t][j];
        }
#if AZ_PRIO_ALT
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    }
    AZ_LSTAMP(1);
    if constexpr (OUT3) __syncthreads();   // every wave has finished reading the conv3 input image
    AZ_LSTAMP(2);
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int nt = nt_base + ng * NTW + t;
        float bco[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bco[rg] = bias_pre[t][rg];
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            const int mt = mg + i * MG;
            if (mt < mt_cnt) {
                const int m = (mt_base + mt) * 16 + r16;
                const bool valid = cellof[m] != 0xFFFFu;
                const int pos = (int)wpos[m] + out_pos_off;
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int co = nt * 16 + q * 4 + rg;      // = 16*cg + 4*e + q' with cg = nt, e = q, q' = rg
                    float v = acc[t][i][rg] + bco[rg];
                    v = v > 0.0f ? v : 0.0f;
                    if constexpr (MODE == CONV_OUT3_PK) {
                        // co = 16 nt + 4 q + rg  ->  plane (nt, rg), component q: the 64 lanes of a store cover 256 contiguous bytes
                        out[(((nt * 4 + rg) * G::MR + m) << 2) + q] = v;
                    } else if constexpr (MODE == CONV_OUT3_RESIDUAL) {
                        float r = acc[t][i][rg] + bco[rg] + (valid ? res[pk_index<GO>(co, pos)] : 0.0f);
                        out[co * O3S + mt * 16 + r16 + (SUBSET ? 0 : mt_base * 16)] = r > 0.0f ? r : 0.0f;
                    } else if constexpr (OUT3) out[co * O3S + mt * 16 + r16 + (SUBSET ? 0 : mt_base * 16)] = v;
                    else if constexpr (MODE == CONV_OUT_RESIDUAL) {
                        if (valid) {                          // net block: relu(bn2(conv2(h)) + x), x updated in place
                            const int oi = pk_index<GO>(co, pos);
                            float r = acc[t][i][rg] + bco[rg] + out[oi];
                            out[oi] = r > 0.0f ? r : 0.0f;
                        }
                    } else if (valid) out[pk_index<GO>(co, pos)] = v;
                }
            }
        }
    }
    AZ_LSTAMP(3);
    }
    };
    if constexpr (MT_REM == 0) body(std::integral_constant<int, MT_FULL>{});
    else if (mg < MT_REM) body(std::integral_constant<int, MT_FULL + 1>{});
    else body(std::integral_constant<int, MT_FULL>{});
}

// policy_conv (128->4) and value_conv (128->2), 1x1 (net.py:64,69): D[head channel][cell] over the float32 conv3 image
// [co][cell] (stride CS3) in LDS, 32 k-steps on the f32 MFMA; relu(acc + bias) -> feature rows in HBM.
template <class G>
__device__ __forceinline__ void trunk_heads(const DevState &d, const NetWeights &w, int net_id, float *__restrict__ feat,
                                            const float *lds, const unsigned short *cellof, int b0, int wave, int lane)
{
    const int q = lane >> 4, r16 = lane & 15;
    const float4 *wp4 = reinterpret_cast<const float4 *>(w.hd) + lane;
    float hb[4];
#pragma unroll
    for (int rg = 0; rg < 4; rg++) hb[rg] = (q * 4 + rg) < 6 ? w.hdb[q * 4 + rg] : 0.0f;
    constexpr int HT = (G::MT + AZ_NW - 1) / AZ_NW;     // tiles per wave
    f32x4 acc[HT];
    // the conv3 image is packed by tile cell (CONV_OUT3_PK): lane (q, r16) reads the float4 of plane (s4, q) at its cell -- the B
    // operands of k-steps 4 s4 .. 4 s4 + 3 (channels 16 s4 + 4 e + q)
    const float4 *ip[HT];
#pragma unroll
    for (int i = 0; i < HT; i++) {
        const int mt = wave + AZ_NW * i;
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ip[i] = reinterpret_cast<const float4 *>(lds) + q * G::MR + (mt < G::MT ? mt : 0) * 16 + r16;
    }
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
        const float4 bq = wp4[s4 * 64];
        const float be[4] = {bq.x, bq.y, bq.z, bq.w};
        float4 av[HT];
#pragma unroll
        for (int i = 0; i < HT; i++) av[i] = ip[i][s4 * 4 * G::MR];
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int i = 0; i < HT; i++) acc[i] = mfma4(be[e], e == 0 ? av[i].x : e == 1 ? av[i].y : e == 2 ? av[i].z : av[i].w, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < HT; i++) {
        const int mt = wave + AZ_NW * i;
        if (mt < G::MT) {
            const int cell = cellof[mt * 16 + r16];
            if (cell != 0xFFFF) {
                const int g = cell / G::nn, p = cell - g * G::nn;
                const int b = b0 + g;
                if (b < d.B && d.s_net[b] == net_id) {
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) {
                        const int j = q * 4 + rg;     // head channel: 0-3 policy_conv, 4-5 value_conv (net.py:64,69 flatten order)
                        if (j < 6) {
                            float v = acc[i][rg] + hb[rg];
                            feat[(size_t)b * G::FROW + j * G::nn + p] = v > 0.0f ? v : 0.0f;
                        }
                    }
                }
            }
        }
    }
}

// The work of one board group.  When this is inlined into a loop over groups the compiler hoists the layers'
// loop-invariant address arithmetic across iterations, runs out of VGPRs (256 + scratch) and the kernel slows down by
// 20 %; callers in a loop therefore pass a thread id made opaque per iteration (asm volatile), which pins that
// arithmetic inside the iteration.  (Out-of-line calls are worse: the call ABI forces spills.)
template <int N>
__device__ __forceinline__ void trunk_group(const DevState &d, const NetWeights &w, int net_id, float *__restrict__ feat,
                                                      unsigned long long *dbg, int grp, float *lds, unsigned short *wpos,
                                                      unsigned short *cellof, int *any_active_p, int tid)
{
    typedef NetGeo<N> G;
    int &any_active = *any_active_p;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = grp * G::G;
    AZ_STAMP(0);
    if (tid == 0) any_active = 0;
    __syncthreads();
    if (tid < G::G) {
        int b = b0 + tid;
        if (b < d.B) {
            int kind = d.leaf_kind[b];
            if ((kind == LEAF_ROOT || kind == LEAF_EXPAND) && d.s_status[b] == SLOT_ACTIVE && d.s_net[b] == net_id)
                atomicOr(&any_active, 1);
        }
    }
    float *inA = lds;                  // 32 channels (conv1 out / conv2 in)
    float *inB = lds + 32 * G::CS;     // 64 channels (conv2 out / conv3 in); first 4 channels hold the input planes
    // Every thread requests the leaf words of its cells (games.py:86-129 encode: ch0 = side to move, ch1 = opponent, ch2 = last
    // action, ch3 = zeros) before it is known whether the group has anything to evaluate: the loads are in flight while both
    // padded images are zeroed (the padding ring must read as 0).
    constexpr int EPT = (G::MR + AZ_NW * 64 - 1) / (AZ_NW * 64);
    int e_pos[EPT];
    bool e_me[EPT], e_op[EPT], e_last[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int m = tid + e * AZ_NW * 64;
        int pos = G::PW + 1, cell = 0xFFFF;
        if (m < G::MR) {
            if constexpr (G::ROWT) {
                const int t = m >> 4, c = m & 15, g = t / N, r = t - g * N;
                pos = g * G::PP + (r + 1) * G::PW + (c + 1);                 // c == N is the right padding cell of the row
                cell = c < N ? g * G::nn + r * N + c : 0xFFFF;
            } else {
                const int g = m / G::nn, p = m - g * G::nn, r = p / N, c = p - r * N;
                pos = m < G::M ? g * G::PP + (r + 1) * G::PW + (c + 1) : G::PW + 1;
                cell = m < G::M ? m : 0xFFFF;
            }
            wpos[m] = (unsigned short)pos;
            cellof[m] = (unsigned short)cell;
        }
        e_pos[e] = pos;
        e_me[e] = e_op[e] = e_last[e] = false;
        if (cell != 0xFFFF) {
            const int g = cell / G::nn, p = cell - g * G::nn;
            const int b = b0 + g;
            if (b < d.B) {
                const u64 *lf = d.leaf + (size_t)b * 8;
                const int ps = sym_cell(d.leaf_sym, b, p, N);
                e_me[e] = (lf[ps >> 6] >> (ps & 63)) & 1ull;
                e_op[e] = (lf[4 + (ps >> 6)] >> (ps & 63)) & 1ull;
                e_last[e] = d.leaf_last[b] == ps;
            }
        }
    }
    {
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < (96 * G::CS) / 4; i += AZ_NW * 64) z[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (!any_active) return;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        if (e_me[e]) inB[e_pos[e]] = 1.0f;
        if (e_op[e]) inB[G::CS + e_pos[e]] = 1.0f;
        if (e_last[e]) inB[2 * G::CS + e_pos[e]] = 1.0f;
    }
    __syncthreads();
    AZ_STAMP(1);
    conv_layer<G, 4, 32, CONV_OUT_PACKED>(inB, inA, w.c1, w.c1b, wpos, cellof, wave, lane);
    __syncthreads();
    // the input planes lived in the first 3 planes of inB; clear them before conv2's packed output lands there
    // (in packed coordinates some of those floats are padding-ring cells that conv2 never writes)
    for (int i = tid; i < 3 * G::CS; i += AZ_NW * 64) inB[i] = 0.0f;
    __syncthreads();
    AZ_STAMP(2);
    conv_layer<G, 32, 64, CONV_OUT_PACKED>(inA, inB, w.c2, w.c2b, wpos, cellof, wave, lane);
    __syncthreads();
    AZ_STAMP(3);
#ifdef AZ_STAMPS
    unsigned long long *lst3 = dbg ? dbg + (size_t)d.B * 16 + 4096 * 32 + (size_t)grp * 32 : nullptr;      // behind k_fc's stamps
#else
    unsigned long long *lst3 = nullptr;
#endif
    conv_layer<G, 64, 128, CONV_OUT3_PK>(inB, lds, w.c3, w.c3b, wpos, cellof, wave, lane, 0, 0, G::MT, 0, nullptr, lst3);
    __syncthreads();
    AZ_STAMP(4);
    trunk_heads<G>(d, w, net_id, feat, lds, cellof, b0, wave, lane);
    AZ_STAMP(5);
}

template <int N>
__global__ __launch_bounds__(AZ_NW * 64) void k_trunk(DevState d, NetWeights w, int net_id, float *__restrict__ feat,
                                               unsigned long long *dbg)
{
    typedef NetGeo<N> G;
    __shared__ __attribute__((aligned(16))) float lds[G::LDSF];
    __shared__ unsigned short wpos[G::MR];     // centre position of tile cell m in the padded image
    __shared__ unsigned short cellof[G::MR];   // g*nn + cell index, 0xFFFF for a junk lane
    __shared__ int any_active;
    const int ngroups = (d.B + G::G - 1) / G::G;
#if AZ_SEQ == 0
    // persistent: one workgroup per CU walks the groups with a grid stride
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        trunk_group<N>(d, w, net_id, feat, dbg, grp, lds, wpos, cellof, &any_active, tid);
        __syncthreads();
    }
#else
#pragma unroll 1
    for (int it = 0; it < AZ_SEQ; it++) {
        const int grp = blockIdx.x * AZ_SEQ + it;
        if (grp >= ngroups) break;
        int tid = threadIdx.x;
        if (AZ_SEQ > 1) asm volatile("" : "+v"(tid));
        trunk_group<N>(d, w, net_id, feat, dbg, grp, lds, wpos, cellof, &any_active, tid);
        if (it + 1 < AZ_SEQ) __syncthreads();      // the LDS image is reused by the next group
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Split trunk (GomokuNet): the low-latency path for launches with few pending boards (episode tails, the arena,
// single-position search).  The fused k_trunk keeps a board on ONE CU (~95 us of paced MFMAs at n = 15); here every
// layer is its own launch and a board's channel tiles are spread over 2 / 4 / 8 workgroups, the activations crossing
// HBM/L2 in the same packed images (scratch per board group, padding ring zeroed once at allocation).  Same fma
// chains, so the results are bit-identical to k_trunk.
//   stage 1: encode + conv1   grid (groups, 2)      -> img1  (32 channels, packed)
//   stage 2: conv2            grid (groups, 4)      -> img2  (64 channels, packed)
//   stage 3: conv3            grid (groups, 8)      -> img3  ([co][cell], stride CS3)
//   stage 4: 1x1 heads        grid (groups)         -> feature rows
// ------------------------------------------------------------------------------------------------
template <int N>
struct SplitGeo {
    typedef NetGeo<N> G;
    static constexpr int IMG1 = 32 * G::CS, IMG2 = 64 * G::CS, IMG3 = 128 * G::CS3;
    static constexpr int PER_GROUP = IMG1 + IMG2 + IMG3;      // floats of scratch per board group
};

template <int N>
__device__ __forceinline__ bool split_prologue(const DevState &d, int net_id, int b0, unsigned short *wpos,
                                               unsigned short *cellof, int *any_active, int tid)
{
    typedef NetGeo<N> G;
    if (tid == 0) *any_active = 0;
    __syncthreads();
    if (tid < G::G) {
        int b = b0 + tid;
        if (b < d.B) {
            int kind = d.leaf_kind[b];
            if ((kind == LEAF_ROOT || kind == LEAF_EXPAND) && d.s_status[b] == SLOT_ACTIVE && d.s_net[b] == net_id)
                atomicOr(any_active, 1);
        }
    }
    for (int m = tid; m < G::MR; m += G::NW * 64) {
        int pos, cell;
        if constexpr (G::ROWT) {
            const int t = m >> 4, c = m & 15, g = t / N, r = t - g * N;
            pos = g * G::PP + (r + 1) * G::PW + (c + 1);
            cell = c < N ? g * G::nn + r * N + c : 0xFFFF;
        } else {
            const int g = m / G::nn, p = m - g * G::nn, r = p / N, c = p - r * N;
            pos = m < G::M ? g * G::PP + (r + 1) * G::PW + (c + 1) : G::PW + 1;
            cell = m < G::M ? m : 0xFFFF;
        }
        wpos[m] = (unsigned short)pos;
        cellof[m] = (unsigned short)cell;
    }
    __syncthreads();
    return *any_active != 0;
}

template <int N, int STAGE>
__global__ __launch_bounds__(AZ_NW * 64) void k_split(DevState d, NetWeights w, int net_id, float *__restrict__ scratch,
                                                      float *__restrict__ feat)
{
    typedef NetGeo<N> G;
    typedef SplitGeo<N> SG;
    constexpr int NTH = G::NW * 64;
    constexpr int LDSF = STAGE == 1 ? 4 * G::CS : (STAGE == 2 ? 32 * G::CS : (STAGE == 3 ? 64 * G::CS : 128 * G::CS3));
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    __shared__ unsigned short wpos[G::MR];
    __shared__ unsigned short cellof[G::MR];
    __shared__ int any_active;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.x, b0 = grp * G::G;
    if (!split_prologue<N>(d, net_id, b0, wpos, cellof, &any_active, tid)) return;
    float *img1 = scratch + (size_t)grp * SG::PER_GROUP, *img2 = img1 + SG::IMG1, *img3 = img2 + SG::IMG2;
    if constexpr (STAGE == 1) {
        for (int i = tid; i < LDSF; i += NTH) lds[i] = 0.0f;
        __syncthreads();
        for (int m = tid; m < G::MR; m += NTH) {          // games.py:86-129 encode
            const int cell = cellof[m];
            if (cell != 0xFFFF) {
                const int g = cell / G::nn, p = cell - g * G::nn;
                const int b = b0 + g;
                if (b < d.B) {
                    const u64 *lf = d.leaf + (size_t)b * 8;
                    const int pos = wpos[m];
                    const int ps = sym_cell(d.leaf_sym, b, p, G::n);
                    if ((lf[ps >> 6] >> (ps & 63)) & 1ull) lds[pos] = 1.0f;
                    if ((lf[4 + (ps >> 6)] >> (ps & 63)) & 1ull) lds[G::CS + pos] = 1.0f;
                    if (d.leaf_last[b] == ps) lds[2 * G::CS + pos] = 1.0f;
                }
            }
        }
        __syncthreads();
        conv_layer<G, 4, 32, CONV_OUT_PACKED, 1>(lds, img1, w.c1, w.c1b, wpos, cellof, wave, lane, blockIdx.y);
    } else if constexpr (STAGE == 2 || STAGE == 3) {
        const float4 *src = reinterpret_cast<const float4 *>(STAGE == 2 ? img1 : img2);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < LDSF / 4; i += NTH) dst[i] = src[i];
        __syncthreads();
        if constexpr (STAGE == 2) conv_layer<G, 32, 64, CONV_OUT_PACKED, 1>(lds, img2, w.c2, w.c2b, wpos, cellof, wave, lane, blockIdx.y);
        else conv_layer<G, 64, 128, CONV_OUT3, 1>(lds, img3, w.c3, w.c3b, wpos, cellof, wave, lane, blockIdx.y);
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(img3);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < LDSF / 4; i += NTH) dst[i] = src[i];
        __syncthreads();
        const int q = lane >> 4, r16 = lane & 15;
        const float4 *wp4 = reinterpret_cast<const float4 *>(w.hd) + lane;
        float hb[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) hb[rg] = (q * 4 + rg) < 6 ? w.hdb[q * 4 + rg] : 0.0f;
        for (int mt = wave; mt < G::MT; mt += G::NW) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const float *ip = lds + q * G::CS3 + mt * 16 + r16;
#pragma unroll
            for (int s4 = 0; s4 < 8; s4++) {
                const float4 bq = wp4[s4 * 64];
                acc = mfma4(bq.x, ip[(s4 * 16 + 0) * G::CS3], acc);
                acc = mfma4(bq.y, ip[(s4 * 16 + 4) * G::CS3], acc);
                acc = mfma4(bq.z, ip[(s4 * 16 + 8) * G::CS3], acc);
                acc = mfma4(bq.w, ip[(s4 * 16 + 12) * G::CS3], acc);
            }
            const int cell = cellof[mt * 16 + r16];
            if (cell != 0xFFFF) {
                const int g = cell / G::nn, p = cell - g * G::nn;
                const int b = b0 + g;
                if (b < d.B && d.s_net[b] == net_id) {
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) {
                        const int j = q * 4 + rg;
                        if (j < 6) {
                            float v = acc[rg] + hb[rg];
                            feat[(size_t)b * G::FROW + j * G::nn + p] = v > 0.0f ? v : 0.0f;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Tile-split trunk (GomokuNet): the low-latency path, round 3.  k_split above spreads a board over workgroups by CHANNEL
// tiles, so every layer is a launch (each needs all channels of its input), conv3 runs 2 x 144 MFMAs per wave and the
// 1x1 heads need a fourth launch that reads the whole 123 KB conv3 image back.  Measured at one pending board
// (profiles/r03_*): 4.6 + 9.0 + 15.2 + 8.6 us of kernels per evaluation, launch gaps ~0.3 us each -- the gaps are not the
// cost, the kernels are.  Here a board is spread by CELL tiles (one 16-cell MFMA tile = one board row at n = 15):
//   stage A  grid (groups, MT): encode the rows around tile t, conv1 on the <= 4 tiles conv2 needs (recomputed per
//            workgroup: 9 k-steps), conv2 on tile t (4 waves x 72 MFMAs)        -> img2, the packed 64-channel board image
//   stage B  grid (groups, MT): the rows of img2 around tile t -> conv3 on tile t (8 waves x 144 MFMAs: all 128 channels of
//            the tile's cells are in THIS workgroup) -> 1x1 heads of the tile (32 MFMAs) -> feature rows
// Two launches instead of four, no conv3 image, 16-24 KB of LDS per workgroup (several per CU).  Every output element is
// the same k-ordered fma chain as in k_trunk -- which workgroup computes it does not enter -- so the results are
// bit-identical (every parity test runs on this path too).
// ------------------------------------------------------------------------------------------------
template <class F, int NWT>
struct TileGeoT {
    static constexpr int N = F::n, PW = F::PW, PP = F::PP, MT = F::MT, nn = F::nn, M = F::M;
    // padded-image position of tile lane m (the centre of its 3x3 window), -1 for a lane without a cell to compute
    __host__ __device__ static constexpr int centre(int m)
    {
        if (F::ROWT) {
            const int t = m >> 4, c = m & 15, g = t / N, r = t - g * N;
            return g * PP + (r + 1) * PW + (c + 1);          // c == N: the row's right padding cell (computed, never stored)
        }
        if (m >= M) return -1;
        const int g = m / nn, p = m - g * nn, r = p / N, c = p - r * N;
        return g * PP + (r + 1) * PW + (c + 1);
    }
    __host__ __device__ static constexpr int lo_of(int t)
    {
        int v = 1 << 30;
        for (int i = 0; i < 16; i++) { const int c = centre(16 * t + i); if (c >= 0 && c < v) v = c; }
        return v;
    }
    __host__ __device__ static constexpr int hi_of(int t)
    {
        int v = -1;
        for (int i = 0; i < 16; i++) { const int c = centre(16 * t + i); if (c > v) v = c; }
        return v;
    }
    // first-conv tiles stage A computes for tile t: every tile with a cell next to a cell of t (same board group)
    __host__ __device__ static constexpr int c1_lo(int t)
    {
        if (F::ROWT) return (t % N) > 0 ? t - 1 : t;
        const int a = 16 * t - (N + 1);
        return a < 0 ? 0 : a >> 4;
    }
    __host__ __device__ static constexpr int c1_hi(int t)
    {
        if (F::ROWT) return (t % N) < N - 1 ? t + 1 : t;
        const int b = (16 * t + 15 + N + 1) >> 4;
        return b > MT - 1 ? MT - 1 : b;
    }
    // origins (first padded-image position held in LDS) and extents of the two kinds of image
    __host__ __device__ static constexpr int org_b(int t) { return lo_of(t) - (PW + 1); }
    __host__ __device__ static constexpr int org_a(int t) { return lo_of(c1_lo(t)) - (PW + 1); }
    __host__ __device__ static constexpr int span_b()
    {
        int v = 0;
        for (int t = 0; t < MT; t++) { const int e = hi_of(t) + (PW + 1) - org_b(t) + 1; if (e > v) v = e; }
        return v;
    }
    __host__ __device__ static constexpr int span_a()
    {
        int v = 0;
        for (int t = 0; t < MT; t++) { const int e = hi_of(c1_hi(t)) + (PW + 1) - org_a(t) + 1; if (e > v) v = e; }
        return v;
    }
    __host__ __device__ static constexpr int c1_tiles()
    {
        int v = 0;
        for (int t = 0; t < MT; t++) { const int e = c1_hi(t) - c1_lo(t) + 1; if (e > v) v = e; }
        return v;
    }
    static constexpr int CSA = ((span_a() + 15) / 16) * 16, CSB = ((span_b() + 15) / 16) * 16;   // plane strides: multiples of 16 (conflict-free b128 reads)
    static constexpr int MTL1 = c1_tiles();
    static_assert(MTL1 <= 4, "stage A covers the first conv with at most 4 cell tiles per channel tile");
    // what conv_layer sees of the geometry in each stage: the full board's tiling, the local image's stride
    struct A { static constexpr int NW = NWT, MT = F::MT, PW = F::PW, CS = CSA, CS3 = 16; };
    struct B { static constexpr int NW = NWT, MT = F::MT, PW = F::PW, CS = CSB, CS3 = 16; };

    // relative-position tables of one workgroup (origin org, image stride csl) + the group's activity flag; false = nothing to do
    __device__ static __forceinline__ bool prologue(const DevState &d, int net_id, int b0, int org, int csl, unsigned short *wpos,
                                                    unsigned short *cellof, int *any_active, int tid, int nth)
    {
        if (tid == 0) *any_active = 0;
        __syncthreads();
        if (tid < F::G) {
            const int b = b0 + tid;
            if (b < d.B) {
                const int kind = d.leaf_kind[b];
                if ((kind == LEAF_ROOT || kind == LEAF_EXPAND) && d.s_status[b] == SLOT_ACTIVE && d.s_net[b] == net_id)
                    atomicOr(any_active, 1);
            }
        }
        for (int m = tid; m < F::MR; m += nth) {
            const int c = centre(m);
            int cell = 0xFFFF;
            if (c >= 0) {
                if constexpr (F::ROWT) { if ((m & 15) < N) cell = (m >> 4) * N + (m & 15); }
                else cell = m;
            }
            const int rel = c - org;
            // lanes outside this workgroup's image (other tiles, junk lanes) get a harmless in-range position: they are
            // either never used or computed and thrown away
            wpos[m] = (unsigned short)((c >= 0 && rel >= PW + 1 && rel + PW + 1 < csl) ? rel : PW + 1);
            cellof[m] = (unsigned short)cell;
        }
        __syncthreads();
        return *any_active != 0;
    }
    // games.py:86-129 encode of the cells inside an image of stride csl starting at org: planes [mover | opponent | last move | 0]
    __device__ static __forceinline__ void encode(const DevState &d, int b0, int org, int csl, float *planes, const unsigned short *cellof,
                                                  int tid, int nth)
    {
        for (int m = tid; m < F::MR; m += nth) {
            const int cell = cellof[m];
            const int rel = centre(m) - org;
            if (cell != 0xFFFF && rel >= 0 && rel < csl) {
                const int g = cell / nn, p = cell - g * nn;
                const int b = b0 + g;
                if (b < d.B) {
                    const u64 *lf = d.leaf + (size_t)b * 8;
                    const int ps = sym_cell(d.leaf_sym, b, p, N);
                    if ((lf[ps >> 6] >> (ps & 63)) & 1ull) planes[rel] = 1.0f;
                    if ((lf[4 + (ps >> 6)] >> (ps & 63)) & 1ull) planes[csl + rel] = 1.0f;
                    if (d.leaf_last[b] == ps) planes[2 * csl + rel] = 1.0f;
                }
            }
        }
    }
    // the positions org .. org + CSB of the 16 planes (cg, q) of a packed 64-channel board image (stride F::CS) into LDS
    __device__ static __forceinline__ void load_rows(const float *img, int org, float *lds, int tid, int nth)
    {
        const float4 *src = reinterpret_cast<const float4 *>(img);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < 16 * CSB; i += nth) {
            const int plane = i / CSB, rel = i - plane * CSB, pos = org + rel;
            dst[i] = (pos >= 0 && pos < F::CS) ? src[plane * F::CS + pos] : float4{0.f, 0.f, 0.f, 0.f};
        }
    }
};
template <int N>
struct TileGeo : TileGeoT<NetGeo<N>, AZ_NW> {
    typedef TileGeoT<NetGeo<N>, AZ_NW> T;
    static constexpr int LDSA = 36 * T::CSA, LDSB = 64 * T::CSB + 128 * 16;
};

template <int N, int STAGE>
__global__ __launch_bounds__(AZ_NW * 64) void k_tile(DevState d, NetWeights w, int net_id, float *__restrict__ scratch,
                                                     float *__restrict__ feat)
{
    typedef NetGeo<N> F;
    typedef TileGeo<N> TG;
    typedef SplitGeo<N> SG;
    constexpr int NTH = F::NW * 64;
    __shared__ __attribute__((aligned(16))) float lds[STAGE == 1 ? TG::LDSA : TG::LDSB];
    __shared__ unsigned short wpos[F::MR];      // centre of tile lane m RELATIVE to this workgroup's image origin
    __shared__ unsigned short cellof[F::MR];
    __shared__ int any_active;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.x, t = blockIdx.y, b0 = grp * F::G;
    const int org = STAGE == 1 ? TG::org_a(t) : TG::org_b(t);
    if constexpr (STAGE == 1) {
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < TG::LDSA / 4; i += NTH) z[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    if (!TG::prologue(d, net_id, b0, org, STAGE == 1 ? TG::CSA : TG::CSB, wpos, cellof, &any_active, tid, NTH)) return;
    float *img2 = scratch + (size_t)grp * SG::PER_GROUP + SG::IMG1;      // the packed 64-channel board image (padding ring zeroed once)
    if constexpr (STAGE == 1) {
        float *planes = lds, *img1 = lds + 4 * TG::CSA;
        TG::encode(d, b0, org, TG::CSA, planes, cellof, tid, NTH);
        __syncthreads();
        const int tlo = TG::c1_lo(t), cnt = TG::c1_hi(t) - tlo + 1;
        conv_layer<typename TG::A, 4, 32, CONV_OUT_PACKED, 2, TG::MTL1>(planes, img1, w.c1, w.c1b, wpos, cellof, wave, lane, 0, tlo, cnt);
        __syncthreads();
        conv_layer<typename TG::A, 32, 64, CONV_OUT_PACKED, 4, 1, F>(img1, img2, w.c2, w.c2b, wpos, cellof, wave, lane, 0, t, 1, org);
    } else {
        TG::load_rows(img2, org, lds, tid, NTH);              // the rows of the conv2 image this tile's windows touch
        __syncthreads();
        float *out3 = lds + 64 * TG::CSB;                     // [co][16 cells of the tile]
        conv_layer<typename TG::B, 64, 128, CONV_OUT3, 8, 1, typename TG::B, 16>(lds, out3, w.c3, w.c3b, wpos, cellof, wave, lane, 0, t, 1);
        __syncthreads();
        if (wave == 0) {                                      // policy_conv (128->4) and value_conv (128->2) of the tile's cells
            const int q = lane >> 4, r16 = lane & 15;
            const float4 *wp4 = reinterpret_cast<const float4 *>(w.hd) + lane;
            float hb[4];
#pragma unroll
            for (int rg = 0; rg < 4; rg++) hb[rg] = (q * 4 + rg) < 6 ? w.hdb[q * 4 + rg] : 0.0f;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const float *ip = out3 + q * 16 + r16;
#pragma unroll
            for (int s4 = 0; s4 < 8; s4++) {
                const float4 bq = wp4[s4 * 64];
                acc = mfma4(bq.x, ip[(s4 * 16 + 0) * 16], acc);
                acc = mfma4(bq.y, ip[(s4 * 16 + 4) * 16], acc);
                acc = mfma4(bq.z, ip[(s4 * 16 + 8) * 16], acc);
                acc = mfma4(bq.w, ip[(s4 * 16 + 12) * 16], acc);
            }
            const int cell = cellof[t * 16 + r16];
            if (cell != 0xFFFF) {
                const int g = cell / F::nn, p = cell - g * F::nn;
                const int b = b0 + g;
                if (b < d.B && d.s_net[b] == net_id) {
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) {
                        const int j = q * 4 + rg;
                        if (j < 6) {
                            float v = acc[rg] + hb[rg];
                            feat[(size_t)b * F::FROW + j * F::nn + p] = v > 0.0f ? v : 0.0f;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ResidualBlock variant: stem conv(4->64)+BN+ReLU, 3 x {conv+BN+ReLU, conv+BN, +skip, ReLU}, 1x1 heads (2 policy
// channels, 1 value channel) + BN + ReLU.  Eval-mode BatchNorm is folded into the conv weights/biases by the host
// layer.  Two 64-channel packed images in LDS: A holds the block input/output (updated in place by the skip add),
// B the intermediate.
// ------------------------------------------------------------------------------------------------
struct ResWeights {
    const float *stem, *stemb;
    const float *blk[6], *blkb[6];     // res1.conv1, res1.conv2, res2.conv1, ... (MFMA-fragment packed), folded biases
    const float *hd, *hdb;             // policy_conv (2) + value_conv (1) rows of one 16-row tile, folded biases [3]
    const void *stemx[2], *blkx[2][6], *hdx[2];   // stem, the six 64 -> 64 convs and the head rows split into 16-bit fragments (az_net_emul.h), [0] bf16x3, [1] f16x2
};

template <int N>
__global__ __launch_bounds__(ResGeo<N>::NW * 64) void k_trunk_res(DevState d, ResWeights w, int net_id, float *__restrict__ feat)
{
    typedef ResGeo<N> G;
    constexpr int NTH = G::NW * 64;
    __shared__ __attribute__((aligned(16))) float lds[G::LDSF];
    __shared__ unsigned short wpos[G::MR];
    __shared__ unsigned short cellof[G::MR];
    __shared__ int any_active;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * G::G;
    if (tid == 0) any_active = 0;
    __syncthreads();
    if (tid < G::G) {
        int b = b0 + tid;
        if (b < d.B) {
            int kind = d.leaf_kind[b];
            if ((kind == LEAF_ROOT || kind == LEAF_EXPAND) && d.s_status[b] == SLOT_ACTIVE && d.s_net[b] == net_id)
                atomicOr(&any_active, 1);
        }
    }
    __syncthreads();
    if (!any_active) return;
    float *A = lds, *B = lds + 64 * G::CS;
    {
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < G::LDSF / 4; i += NTH) z[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    for (int m = tid; m < G::MR; m += NTH) {
        int pos, cell;
        if constexpr (G::ROWT) {
            const int t = m >> 4, c = m & 15, g = t / N, r = t - g * N;
            pos = g * G::PP + (r + 1) * G::PW + (c + 1);
            cell = c < N ? g * G::nn + r * N + c : 0xFFFF;
        } else {
            const int g = m / G::nn, p = m - g * G::nn, r = p / N, c = p - r * N;
            pos = m < G::M ? g * G::PP + (r + 1) * G::PW + (c + 1) : G::PW + 1;
            cell = m < G::M ? m : 0xFFFF;
        }
        wpos[m] = (unsigned short)pos;
        cellof[m] = (unsigned short)cell;
    }
    __syncthreads();
    for (int m = tid; m < G::MR; m += NTH) {          // games.py:86-129 encode into plain planes at the start of B
        const int cell = cellof[m];
        if (cell != 0xFFFF) {
            const int g = cell / G::nn, p = cell - g * G::nn;
            const int b = b0 + g;
            if (b < d.B) {
                const u64 *lf = d.leaf + (size_t)b * 8;
                const int pos = wpos[m];
                const int ps = sym_cell(d.leaf_sym, b, p, N);
                if ((lf[ps >> 6] >> (ps & 63)) & 1ull) B[pos] = 1.0f;
                if ((lf[4 + (ps >> 6)] >> (ps & 63)) & 1ull) B[G::CS + pos] = 1.0f;
                if (d.leaf_last[b] == ps) B[2 * G::CS + pos] = 1.0f;
            }
        }
    }
    __syncthreads();
    conv_layer<G, 4, 64, CONV_OUT_PACKED>(B, A, w.stem, w.stemb, wpos, cellof, wave, lane);
    __syncthreads();
    for (int i = tid; i < 3 * G::CS; i += NTH) B[i] = 0.0f;     // the planes would alias padding cells of the packed image
    __syncthreads();
#pragma unroll 1
    for (int blk = 0; blk < 3; blk++) {
        conv_layer<G, 64, 64, CONV_OUT_PACKED>(A, B, w.blk[2 * blk], w.blkb[2 * blk], wpos, cellof, wave, lane);
        __syncthreads();
        conv_layer<G, 64, 64, CONV_OUT_RESIDUAL>(B, A, w.blk[2 * blk + 1], w.blkb[2 * blk + 1], wpos, cellof, wave, lane);
        __syncthreads();
    }
    // heads: D[head channel][cell] over the packed trunk image A, 16 k-steps (64 channels)
    {
        const int q = lane >> 4, r16 = lane & 15;
        const float4 *wp4 = reinterpret_cast<const float4 *>(w.hd) + lane;
        const float4 *in4 = reinterpret_cast<const float4 *>(A);
        float hb[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) hb[rg] = (q * 4 + rg) < (G::PC + G::VC) ? w.hdb[q * 4 + rg] : 0.0f;
        for (int mt = wave; mt < G::MT; mt += G::NW) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const int base = q * G::CS + (int)wpos[mt * 16 + r16];
#pragma unroll
            for (int cg = 0; cg < 4; cg++) {
                const float4 a = in4[base + cg * 4 * G::CS];
                const float4 wq = wp4[cg * 64];
                acc = mfma4(wq.x, a.x, acc);
                acc = mfma4(wq.y, a.y, acc);
                acc = mfma4(wq.z, a.z, acc);
                acc = mfma4(wq.w, a.w, acc);
            }
            const int cell = cellof[mt * 16 + r16];
            if (cell != 0xFFFF) {
                const int g = cell / G::nn, p = cell - g * G::nn;
                const int b = b0 + g;
                if (b < d.B && d.s_net[b] == net_id) {
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) {
                        const int j = q * 4 + rg;     // 0-1 policy_conv, 2 value_conv
                        if (j < G::PC + G::VC) {
                            float v = acc[rg] + hb[rg];
                            feat[(size_t)b * G::FROW + j * G::nn + p] = v > 0.0f ? v : 0.0f;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split trunk for the ResidualBlock net: the low-latency path (few pending boards: the arena, episode tails, single
// searches), like k_split for GomokuNet.  Every conv is its own launch and a board group's four 16-channel tiles go to
// four workgroups; the 64-channel packed images A (block input/output) and B (intermediate) cross HBM/L2 in the scratch
// of the group (padding ring zeroed once at allocation, never written).  Same fma chains as k_trunk_res: bit-identical.
//   KIND 0: encode + stem             grid (groups, 4) -> A
//   KIND 1: block conv1               grid (groups, 4)    A -> B
//   KIND 2: block conv2 + skip, ReLU  grid (groups, 4)    B -> A (in place: a workgroup reads and writes only its own channel tile of A)
//   KIND 3: 1x1 heads                 grid (groups)       A -> feature rows
// ------------------------------------------------------------------------------------------------
template <int N>
struct ResSplitGeo {
    typedef ResGeo<N> G;
    static constexpr int IMG = 64 * G::CS;
    static constexpr int PER_GROUP = 2 * IMG;                 // floats of scratch per board group
};

template <class G>
__device__ __forceinline__ bool split_prologue_g(const DevState &d, int net_id, int b0, unsigned short *wpos,
                                                 unsigned short *cellof, int *any_active, int tid)
{
    if (tid == 0) *any_active = 0;
    __syncthreads();
    if (tid < G::G) {
        int b = b0 + tid;
        if (b < d.B) {
            int kind = d.leaf_kind[b];
            if (leaf_needs_net(kind) && d.s_status[b] == SLOT_ACTIVE && d.s_net[b] == net_id) atomicOr(any_active, 1);
        }
    }
    for (int m = tid; m < G::MR; m += G::NW * 64) {
        int pos, cell;
        if constexpr (G::ROWT) {
            const int t = m >> 4, c = m & 15, g = t / G::n, r = t - g * G::n;
            pos = g * G::PP + (r + 1) * G::PW + (c + 1);
            cell = c < G::n ? g * G::nn + r * G::n + c : 0xFFFF;
        } else {
            const int g = m / G::nn, p = m - g * G::nn, r = p / G::n, c = p - r * G::n;
            pos = m < G::M ? g * G::PP + (r + 1) * G::PW + (c + 1) : G::PW + 1;
            cell = m < G::M ? m : 0xFFFF;
        }
        wpos[m] = (unsigned short)pos;
        cellof[m] = (unsigned short)cell;
    }
    __syncthreads();
    return *any_active != 0;
}

template <int N, int KIND>
__global__ __launch_bounds__(ResGeo<N>::NW * 64) void k_split_res(DevState d, const float *__restrict__ wp, const float *__restrict__ bias,
                                                                  int net_id, float *__restrict__ scratch, float *__restrict__ feat)
{
    typedef ResGeo<N> G;
    constexpr int NTH = G::NW * 64;
    constexpr int LDSF = KIND == 0 ? 4 * G::CS : 64 * G::CS;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    __shared__ unsigned short wpos[G::MR];
    __shared__ unsigned short cellof[G::MR];
    __shared__ int any_active;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.x, b0 = grp * G::G;
    if (!split_prologue_g<G>(d, net_id, b0, wpos, cellof, &any_active, tid)) return;
    float *A = scratch + (size_t)grp * ResSplitGeo<N>::PER_GROUP, *B = A + ResSplitGeo<N>::IMG;
    if constexpr (KIND == 0) {
        for (int i = tid; i < LDSF; i += NTH) lds[i] = 0.0f;
        __syncthreads();
        for (int m = tid; m < G::MR; m += NTH) {          // games.py:86-129 encode
            const int cell = cellof[m];
            if (cell != 0xFFFF) {
                const int g = cell / G::nn, p = cell - g * G::nn;
                const int b = b0 + g;
                if (b < d.B) {
                    const u64 *lf = d.leaf + (size_t)b * 8;
                    const int pos = wpos[m];
                    const int ps = sym_cell(d.leaf_sym, b, p, G::n);
                    if ((lf[ps >> 6] >> (ps & 63)) & 1ull) lds[pos] = 1.0f;
                    if ((lf[4 + (ps >> 6)] >> (ps & 63)) & 1ull) lds[G::CS + pos] = 1.0f;
                    if (d.leaf_last[b] == ps) lds[2 * G::CS + pos] = 1.0f;
                }
            }
        }
        __syncthreads();
        conv_layer<G, 4, 64, CONV_OUT_PACKED, 1>(lds, A, wp, bias, wpos, cellof, wave, lane, blockIdx.y);
    } else {
        const float4 *src = reinterpret_cast<const float4 *>(KIND == 2 ? B : A);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < LDSF / 4; i += NTH) dst[i] = src[i];
        __syncthreads();
        if constexpr (KIND == 1) conv_layer<G, 64, 64, CONV_OUT_PACKED, 1>(lds, B, wp, bias, wpos, cellof, wave, lane, blockIdx.y);
        else if constexpr (KIND == 2) conv_layer<G, 64, 64, CONV_OUT_RESIDUAL, 1>(lds, A, wp, bias, wpos, cellof, wave, lane, blockIdx.y);
        else {
            // heads: D[head channel][cell] over the packed trunk image, 16 k-steps (64 channels); wp = packed head rows, bias = folded biases [3]
            const int q = lane >> 4, r16 = lane & 15;
            const float4 *wp4 = reinterpret_cast<const float4 *>(wp) + lane;
            const float4 *in4 = reinterpret_cast<const float4 *>(lds);
            float hb[4];
#pragma unroll
            for (int rg = 0; rg < 4; rg++) hb[rg] = (q * 4 + rg) < (G::PC + G::VC) ? bias[q * 4 + rg] : 0.0f;
            for (int mt = wave; mt < G::MT; mt += G::NW) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                const int base = q * G::CS + (int)wpos[mt * 16 + r16];
#pragma unroll
                for (int cg = 0; cg < 4; cg++) {
                    const float4 a = in4[base + cg * 4 * G::CS];
                    const float4 wq = wp4[cg * 64];
                    acc = mfma4(wq.x, a.x, acc);
                    acc = mfma4(wq.y, a.y, acc);
                    acc = mfma4(wq.z, a.z, acc);
                    acc = mfma4(wq.w, a.w, acc);
                }
                const int cell = cellof[mt * 16 + r16];
                if (cell != 0xFFFF) {
                    const int g = cell / G::nn, p = cell - g * G::nn;
                    const int b = b0 + g;
                    if (b < d.B && d.s_net[b] == net_id) {
#pragma unroll
                        for (int rg = 0; rg < 4; rg++) {
                            const int j = q * 4 + rg;     // 0-1 policy_conv, 2 value_conv
                            if (j < G::PC + G::VC) {
                                float v = acc[rg] + hb[rg];
                                feat[(size_t)b * G::FROW + j * G::nn + p] = v > 0.0f ? v : 0.0f;
                            }
                        }
                    }
                }
            }
        }
    }
}

// Tile-split trunk of the ResidualBlock variant: like k_tile, a board spread by CELL tiles, 4 waves per workgroup (the four
// 16-channel tiles of a 64-channel conv on one 16-cell tile: 144 dependent MFMAs each), grid (groups, MT).  Six launches
// instead of k_split_res's eight:
//   KIND 0  encode + stem on the tiles around t (recomputed per workgroup: 9 k-steps) -> x of tile t to image A; res1.conv1 on tile t -> image B
//   KIND 1  rows of B around t -> conv2 of a block on tile t, relu(. + x) with x read and written in place in A
//   KIND 2  rows of A around t -> conv1 of the next block on tile t -> B
//   KIND 3  the last conv2: relu(. + x) stays in LDS as [channel][16 cells] -> 1x1 heads of the tile -> feature rows
// Same fma chains as k_trunk_res: bit-identical.
template <int N>
struct ResTileGeo : TileGeoT<ResGeo<N>, 4> {
    typedef TileGeoT<ResGeo<N>, 4> T;
    static constexpr int LDS0 = 68 * T::CSA, LDSC = 64 * T::CSB + 64 * 16;
};

template <int N, int KIND>
__global__ __launch_bounds__(256) void k_tile_res(DevState d, const float *__restrict__ wp, const float *__restrict__ bias,
                                                  const float *__restrict__ wp2, const float *__restrict__ bias2, int net_id,
                                                  float *__restrict__ scratch, float *__restrict__ feat)
{
    typedef ResGeo<N> F;
    typedef ResTileGeo<N> TG;
    constexpr int NTH = 256;
    __shared__ __attribute__((aligned(16))) float lds[KIND == 0 ? TG::LDS0 : TG::LDSC];
    __shared__ unsigned short wpos[F::MR];
    __shared__ unsigned short cellof[F::MR];
    __shared__ int any_active;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.x, t = blockIdx.y, b0 = grp * F::G;
    const int org = KIND == 0 ? TG::org_a(t) : TG::org_b(t);
    if constexpr (KIND == 0) {
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < TG::LDS0 / 4; i += NTH) z[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    if (!TG::prologue(d, net_id, b0, org, KIND == 0 ? TG::CSA : TG::CSB, wpos, cellof, &any_active, tid, NTH)) return;
    float *A = scratch + (size_t)grp * ResSplitGeo<N>::PER_GROUP, *B = A + ResSplitGeo<N>::IMG;
    if constexpr (KIND == 0) {
        float *planes = lds, *x = lds + 4 * TG::CSA;          // x: the stem's output around tile t, packed, local geometry
        TG::encode(d, b0, org, TG::CSA, planes, cellof, tid, NTH);
        __syncthreads();
        const int tlo = TG::c1_lo(t), cnt = TG::c1_hi(t) - tlo + 1;
        conv_layer<typename TG::A, 4, 64, CONV_OUT_PACKED, 4, TG::MTL1>(planes, x, wp, bias, wpos, cellof, wave, lane, 0, tlo, cnt);
        __syncthreads();
        // x of tile t also goes to the board image A: the skip operand of res1 (read in place by KIND 1)
        for (int i = tid; i < 64 * 16; i += NTH) {
            const int co = i >> 4, m = t * 16 + (i & 15);
            if (cellof[m] != 0xFFFFu) {
                const int rel = wpos[m];
                A[pk_index<F>(co, rel + org)] = x[pk_index<typename TG::A>(co, rel)];
            }
        }
        conv_layer<typename TG::A, 64, 64, CONV_OUT_PACKED, 4, 1, F>(x, B, wp2, bias2, wpos, cellof, wave, lane, 0, t, 1, org);
    } else {
        TG::load_rows(KIND == 2 ? A : B, org, lds, tid, NTH);
        __syncthreads();
        if constexpr (KIND == 1) conv_layer<typename TG::B, 64, 64, CONV_OUT_RESIDUAL, 4, 1, F>(lds, A, wp, bias, wpos, cellof, wave, lane, 0, t, 1, org);
        else if constexpr (KIND == 2) conv_layer<typename TG::B, 64, 64, CONV_OUT_PACKED, 4, 1, F>(lds, B, wp, bias, wpos, cellof, wave, lane, 0, t, 1, org);
        else {
            float *xt = lds + 64 * TG::CSB;                   // [channel][16 cells of the tile]: the trunk's output
            conv_layer<typename TG::B, 64, 64, CONV_OUT3_RESIDUAL, 4, 1, F, 16>(lds, xt, wp, bias, wpos, cellof, wave, lane, 0, t, 1, org, A);
            __syncthreads();
            if (wave == 0) {                                  // policy_conv (64->2) and value_conv (64->1), BatchNorm folded: wp2 = packed head rows, bias2[3]
                const int q = lane >> 4, r16 = lane & 15;
                const float4 *wp4 = reinterpret_cast<const float4 *>(wp2) + lane;
                float hb[4];
#pragma unroll
                for (int rg = 0; rg < 4; rg++) hb[rg] = (q * 4 + rg) < (F::PC + F::VC) ? bias2[q * 4 + rg] : 0.0f;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                const float *ip = xt + q * 16 + r16;
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
                    const float4 bq = wp4[s4 * 64];
                    acc = mfma4(bq.x, ip[(s4 * 16 + 0) * 16], acc);
                    acc = mfma4(bq.y, ip[(s4 * 16 + 4) * 16], acc);
                    acc = mfma4(bq.z, ip[(s4 * 16 + 8) * 16], acc);
                    acc = mfma4(bq.w, ip[(s4 * 16 + 12) * 16], acc);
                }
                const int cell = cellof[t * 16 + r16];
                if (cell != 0xFFFF) {
                    const int g = cell / F::nn, p = cell - g * F::nn;
                    const int b = b0 + g;
                    if (b < d.B && d.s_net[b] == net_id) {
#pragma unroll
                        for (int rg = 0; rg < 4; rg++) {
                            const int j = q * 4 + rg;     // 0-1 policy_conv, 2 value_conv
                            if (j < F::PC + F::VC) {
                                float v = acc[rg] + hb[rg];
                                feat[(size_t)b * F::FROW + j * F::nn + p] = v > 0.0f ? v : 0.0f;
                            }
                        }
                    }
                }
            }
        }
    }
}

// policy_fc (net.py:65) and value_fc1 + ReLU (net.py:69): one workgroup = 16 boards x (TPW x R) 16-output tiles of one
// layer; it works through them in R rounds of TPW tiles, four waves per tile = the four chains of the canonical order
// (fc_chain_groups above).  grid = (ceil(B / 16), ceil(NTP / (TPW R)) policy workgroups + ceil(4 / (TPW R)) value workgroups).
//   * the chain's weight fragments (QG x dwordx4 per lane, L2) are requested first, before anything else, and the next
//     round's while the current round computes;
//   * the features the layer needs (policy or value part of the 16 boards' rows, only boards with a pending evaluation)
//     are staged into LDS once, as [group of 16 inputs][q][board] float4: a thread loads 16 consecutive inputs of one board
//     and stores them transposed 4 x 4, so that lane (q, board) reads the A operands of a group's four MFMAs (inputs
//     16 g + 4 e + q, e = 0..3) with ONE conflict-free ds_read_b128;
//   * QG x 4 dependent MFMAs per wave and round, partial sums combined through LDS in the canonical order.
// <1, 1> (4 waves, one per SIMD, 19 workgroups per board row at n = 15) is the latency shape: few rows, many CUs free.
// <2, 4> (8 waves, 3 workgroups per row) is the throughput shape: with every CU busy under another lane's trunk what the
// kernel costs is the number of CU slots it has to win and the times it stages a row's features (3 instead of 19);
// measured in the 4-lane pipeline at 15x15: one tile per workgroup 158.4 ms per ply, <4, 1> 157.6, trunk-bound floor 155.1.
template <class G, int TPW, int R>
__global__ __launch_bounds__(TPW * 256, (R > 1 && TPW <= 2) ? 2 * TPW : 1) void k_fc(DevState d, NetWeights w, int net_id, const float *__restrict__ feat,
                                                  unsigned long long *dbgfc)
{
    constexpr int NTH = TPW * 256, TW = TPW * R;       // threads, tiles per workgroup
    constexpr int NPW = (G::NTP + TW - 1) / TW;        // policy workgroups per board row
    __shared__ float4 ft4[4 * G::QGMAX * 64];          // [group][q][board]
    __shared__ float4 part[2 * TPW * 3 * 64];          // partial sums of chains 1..3, per tile and lane; double-buffered over the rounds
    __shared__ unsigned fc_active;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chain = wv & 3, tl = wv >> 2;
    const int mb = blockIdx.x * 16;
    const bool is_pol = (int)blockIdx.y < NPW;
    const int tile0 = is_pol ? blockIdx.y * TW + tl : G::NTP + ((int)blockIdx.y - NPW) * TW + tl;   // this wave's tile in round 0
    const int tile_end = is_pol ? G::NTP : G::NTP + 4;
#ifdef AZ_STAMPS
#define FC_STAMP(k) do { if (dbgfc && (threadIdx.x & 63) == 0) dbgfc[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + ((threadIdx.x >> 6) & 7)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FC_STAMP(k) do { } while (0)
#endif
    FC_STAMP(0);
    const int QG = is_pol ? G::QGP : G::QGV;           // groups per chain
    const int K = is_pol ? G::PC * G::nn : G::VC * G::nn;
    const int koff = is_pol ? 0 : G::PC * G::nn;
    const int q = lane >> 4, r16 = lane & 15;
    // a chain's weight fragments: [tile][4 QG groups, zero-padded][lane] float4 (pack_fc)
    const float4 *wbase = reinterpret_cast<const float4 *>(is_pol ? w.pf : w.vf) + (size_t)chain * QG * 64 + lane;
    const int tsub = is_pol ? 0 : G::NTP;
    float4 wf[G::QGMAX];
    {
        const float4 *wp4 = wbase + (size_t)((tile0 < tile_end ? tile0 : tile_end - 1) - tsub) * 4 * QG * 64;
#pragma unroll
        for (int j = 0; j < G::QGMAX; j++) wf[j] = j < QG ? wp4[(size_t)j * 64] : float4{0.f, 0.f, 0.f, 0.f};
    }
    // the epilogues' biases, one per round, requested now so that no round waits for one
    float bias_r[R];
#pragma unroll
    for (int rd = 0; rd < R; rd++) {
        const int tile = tile0 + rd * TPW, o = (tile - tsub) * 16 + r16;
        bias_r[rd] = (chain == 0 && tile < tile_end && (!is_pol || o < G::nn)) ? (is_pol ? w.pfb[o] : w.vfb[o]) : 0.0f;
    }
    // boards of this row with a pending evaluation (episode tails, terminal leaves, cache hits leave rows idle)
    if (tid == 0) fc_active = 0u;
    __syncthreads();
    if (tid < 16) {
        const int b = mb + tid;
        if (b < d.B) {
            const int kind = d.leaf_kind[b];
            if ((kind == LEAF_ROOT || kind == LEAF_EXPAND) && d.s_status[b] == SLOT_ACTIVE && d.s_net[b] == net_id)
                atomicOr(&fc_active, 1u << tid);
        }
    }
    __syncthreads();
    const unsigned active = fc_active;
    if (!active) return;
    // stage: unit u = (group g, board i); inputs beyond K (chain padding) and idle boards are zeros
    const bool vec4 = (koff & 3) == 0;                 // the ResidualBlock net's value inputs start at 2 n^2: 8-byte aligned for odd n
    for (int u = tid; u < 4 * QG * 16; u += NTH) {
        const int g = u >> 4, i = u & 15;
        float x[16];
#pragma unroll
        for (int e = 0; e < 16; e++) x[e] = 0.0f;
        const int k0 = 16 * g;
        if (((active >> i) & 1u) && k0 < K) {
            const float *src = feat + (size_t)(mb + i) * G::FROW + koff + k0;
            if (vec4) {
#pragma unroll
                for (int a = 0; a < 4; a++)
                    if (k0 + 4 * a < K) {
                        const float4 v = reinterpret_cast<const float4 *>(src)[a];
                        x[4 * a] = v.x; x[4 * a + 1] = v.y; x[4 * a + 2] = v.z; x[4 * a + 3] = v.w;
                    }
            } else {
#pragma unroll
                for (int a = 0; a < 8; a++)
                    if (k0 + 2 * a < K) {
                        const float2 v = reinterpret_cast<const float2 *>(src)[a];
                        x[2 * a] = v.x; x[2 * a + 1] = v.y;
                    }
            }
#pragma unroll
            for (int e = 0; e < 16; e++) x[e] = k0 + e < K ? x[e] : 0.0f;     // the row's next part / tail is not this layer's input
        }
#pragma unroll
        for (int qq = 0; qq < 4; qq++) ft4[(g * 4 + qq) * 16 + i] = float4{x[qq], x[4 + qq], x[8 + qq], x[12 + qq]};
    }
    __syncthreads();
    FC_STAMP(1);
    const float4 *fp = ft4 + (chain * QG * 4 + q) * 16 + r16;
#pragma unroll
    for (int rd = 0; rd < R; rd++) {
        const int tile = tile0 + rd * TPW;
        const bool has_tile = tile < tile_end;
        // the next round's fragments are requested into the registers of this round's as soon as those have been used: a
        // whole round (4 QG MFMAs) of flight time, no second register set (two workgroups fit a CU: one stages while one computes)
        const int tn = tile + TPW < tile_end ? tile + TPW : tile_end - 1;
        const float4 *wnext = wbase + (size_t)(tn - tsub) * 4 * QG * 64;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < G::QGMAX; j++)
            if (j < QG) {
                const float4 af = fp[j * 64];
                const float4 wv = wf[j];
                if (R > 1 && rd + 1 < R) wf[j] = wnext[(size_t)j * 64];
                if (has_tile) {
                    acc = mfma4(af.x, wv.x, acc);
                    acc = mfma4(af.y, wv.y, acc);
                    acc = mfma4(af.z, wv.z, acc);
                    acc = mfma4(af.w, wv.w, acc);
                }
            }
        float4 *pr = part + (rd & 1) * TPW * 3 * 64;
        if (chain) pr[(tl * 3 + chain - 1) * 64 + lane] = float4{acc[0], acc[1], acc[2], acc[3]};
        __syncthreads();
        if (!chain && has_tile) {
            const float4 p1 = pr[(tl * 3) * 64 + lane], p2 = pr[(tl * 3 + 1) * 64 + lane], p3 = pr[(tl * 3 + 2) * 64 + lane];
            const float r[4] = {(acc[0] + p1.x) + (p2.x + p3.x), (acc[1] + p1.y) + (p2.y + p3.y),
                                (acc[2] + p1.z) + (p2.z + p3.z), (acc[3] + p1.w) + (p2.w + p3.w)};
            // only slots with a pending evaluation are written: the row of a slot whose evaluation came from the cache (or
            // that waits on a terminal leaf) must stay as it is
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                const int i = q * 4 + rg, b = mb + i;
                if ((active >> i) & 1u) {
                    if (is_pol) {
                        const int j = tile * 16 + r16;
                        if (j < G::nn) d.logits[(size_t)b * G::RW + j] = r[rg] + bias_r[rd];
                    } else {
                        const int o = (tile - G::NTP) * 16 + r16;
                        const float v = r[rg] + bias_r[rd];
                        d.vhid[(size_t)b * 64 + o] = v > 0.0f ? v : 0.0f;
                    }
                }
            }
        }
    }
    FC_STAMP(2);
}
