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

#ifdef AZ_STAMPS
// diagnostic build only: per-workgroup phase time stamps (s_memtime) into a buffer no other code reads
#define AZ_STAMP(k) do { if (threadIdx.x == 0 && dbg) dbg[(size_t)grp * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AZ_STAMP(k) do { } while (0)
#endif

#ifndef AZ_NW
#define AZ_NW 8          // waves per trunk workgroup
#endif
#ifndef AZ_SEQ
#define AZ_SEQ 1         // board groups per trunk workgroup (0 = persistent grid stride); all variants measure the same
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
    static constexpr int KSP = (PC * nn + 3) / 4;                  // policy k-steps
    static constexpr int KSV = (VC * nn + 3) / 4;                  // value_fc1 k-steps
    static constexpr int FROW = (((PC + VC) * nn + 3) / 4) * 4;    // feature row in HBM: [0,PC*nn) policy, [PC*nn,(PC+VC)*nn) value, zero tail
    static constexpr int KS4P_PAD = (((KSP + 3) / 4 + 15) / 16) * 16;   // policy weight groups padded to chunks of 16
    static constexpr int KS4V_PAD = (((KSV + 3) / 4 + 15) / 16) * 16;
    static constexpr int FNEED = (KS4P_PAD * 16 > PC * nn + KS4V_PAD * 16) ? KS4P_PAD * 16 : PC * nn + KS4V_PAD * 16;
    static constexpr int FSTR0 = FROW > FNEED ? FROW : FNEED;     // the padded k-steps read (zero) LDS beyond the row
    static constexpr int FSTR = FSTR0 + ((4 - (FSTR0 % 32) + 32) % 32);   // LDS row stride, == 4 (mod 32), multiple of 4
    static constexpr int FCW = 8;                                  // waves per k_fc workgroup (one 16x16 output tile each)
    static constexpr int NSPLIT = (NTP + 4 + FCW - 1) / FCW;       // workgroups per 16-board row: (B/16)*NSPLIT <= 256 -> one round
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
    static constexpr int KSP = (PC * nn + 3) / 4;
    static constexpr int KSV = (VC * nn + 3) / 4;
    static constexpr int FROW = (((PC + VC) * nn + 3) / 4) * 4;
    static constexpr int KS4P_PAD = (((KSP + 3) / 4 + 15) / 16) * 16;
    static constexpr int KS4V_PAD = (((KSV + 3) / 4 + 15) / 16) * 16;
    static constexpr int FNEED = (KS4P_PAD * 16 > PC * nn + KS4V_PAD * 16) ? KS4P_PAD * 16 : PC * nn + KS4V_PAD * 16;
    static constexpr int FSTR0 = FROW > FNEED ? FROW : FNEED;
    static constexpr int FSTR = FSTR0 + ((4 - (FSTR0 % 32) + 32) % 32);
    static constexpr int FCW = 8;
    static constexpr int NSPLIT = (NTP + 4 + FCW - 1) / FCW;
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
enum { CONV_OUT_PACKED = 0, CONV_OUT3 = 1, CONV_OUT_RESIDUAL = 2 };   // RESIDUAL: relu(acc + bias + out[same index]) in place

// One conv layer on the workgroup's LDS image, computed as D[co][cell] = sum_k W[co][k] * X[k][cell]:
// the weight fragment is the MFMA A operand (row = output channel), the activation fragment the B operand
// (column = board cell), so an accumulator register holds 16 consecutive cells of one channel per 16 lanes.
// A wave owns NTW channel tiles x MTW cell tiles; every activation fragment read from LDS feeds NTW MFMAs.
// CIN == 4 (conv1) reads plain planes [ci][pos]; CIN >= 32 reads the packed image above.
// OUT3=false: relu(acc+bias) -> packed image of the next layer.  OUT3=true (conv3): barrier, then the
// [co][cell] image that overlays the (now dead) inputs.
// NTL / nt_base: the split (low-latency) kernels give one workgroup only NTL of the layer's channel tiles, starting
// at tile nt_base; the fused kernels use all of them.
template <class G, int CIN, int COUT, int MODE, int NTL = COUT / 16>
__device__ __forceinline__ void conv_layer(const float *in, float *out, const float *__restrict__ wp,
                                           const float *__restrict__ bias, const unsigned short *wpos,
                                           const unsigned short *cellof, int wave, int lane, int nt_base = 0)
{
    constexpr bool OUT3 = MODE == CONV_OUT3;
    constexpr int NT = NTL;                                    // channel tiles handled by this workgroup
    constexpr int NTW = (AZ_NTW <= NT) ? AZ_NTW : NT;          // channel tiles per wave
    constexpr int NG = NT / NTW;                               // channel-tile groups
    constexpr int MG = (G::NW / NG) > 0 ? (G::NW / NG) : 1;    // cell-tile groups
    constexpr int MTW = (G::MT + MG - 1) / MG;
    constexpr int KST = CIN / 4;           // k-steps per tap
    constexpr int KS = 9 * KST;
    constexpr int KS4 = (KS + 3) / 4;
    const int ng = wave % NG, mg = wave / NG;       // wave is wave-uniform (readfirstlane) -> scalar control flow
    const int q = lane >> 4, r16 = lane & 15;

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
    if constexpr (CIN == 4) {
        // conv1: 9 k-steps (one per tap), channels = {mover, opponent, last move, zero plane}, planes [ci][pos]
        int rb[MTW];
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            int mt = mg + i * MG;
            int m = (mt < G::MT ? mt : 0) * 16 + r16;       // a surplus tile aliases tile 0 (computed, never written back)
            rb[i] = (int)wpos[m] - (G::PW + 1) + q * G::CS;
        }
        float bk[NTW][12];
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            float4 b0 = wp4[t][0], b1 = wp4[t][64], b2 = wp4[t][128];
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
            int m = (mt < G::MT ? mt : 0) * 16 + r16;
            ra[i] = (int)wpos[m] - (G::PW + 1) + q * G::CS;
        }
        float4 a0[MTW], a1[MTW];
        float4 bw[NTW][NQ], bnx[NTW][NQ];
#pragma unroll
        for (int t = 0; t < NTW; t++)
#pragma unroll
            for (int j = 0; j < NQ; j++) bw[t][j] = wp4[t][(size_t)j * 64];
#pragma unroll
        for (int i = 0; i < MTW; i++) a0[i] = in4[ra[i]];
        for (int tap = 0; tap < 9; tap++) {
            const int tn = tap + 1 < 9 ? tap + 1 : tap;
#pragma unroll
            for (int t = 0; t < NTW; t++)
#pragma unroll
                for (int j = 0; j < NQ; j++) bnx[t][j] = wp4[t][(size_t)(tn * NQ + j) * 64];
            // float4-index step from the last group of this tap to the first group of the next tap
            const int dnext = ((tn / 3) * G::PW + (tn % 3)) - ((tap / 3) * G::PW + (tap % 3)) - (NQ - 1) * 4 * G::CS;
#pragma unroll
            for (int sq = 0; sq < NQ; sq++) {
                float4 *cur = (sq & 1) ? a1 : a0;
                float4 *nxt = (sq & 1) ? a0 : a1;
                const int dstep = sq + 1 < NQ ? 4 * G::CS : dnext;
#pragma unroll
                for (int i = 0; i < MTW; i++) { ra[i] += dstep; nxt[i] = in4[ra[i]]; }
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
    }
    if constexpr (OUT3) __syncthreads();   // every wave has finished reading the conv3 input image
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int nt = nt_base + ng * NTW + t;
        float bco[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bco[rg] = bias_pre[t][rg];
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            const int mt = mg + i * MG;
            if (mt < G::MT) {
                const int m = mt * 16 + r16;
                const bool valid = cellof[m] != 0xFFFFu;
                const int pos = wpos[m];
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const int co = nt * 16 + q * 4 + rg;      // = 16*cg + 4*e + q' with cg = nt, e = q, q' = rg
                    float v = acc[t][i][rg] + bco[rg];
                    v = v > 0.0f ? v : 0.0f;
                    if constexpr (OUT3) out[co * G::CS3 + m] = v;
                    else if constexpr (MODE == CONV_OUT_RESIDUAL) {
                        if (valid) {                          // net block: relu(bn2(conv2(h)) + x), x updated in place
                            const int oi = pk_index<G>(co, pos);
                            float r = acc[t][i][rg] + bco[rg] + out[oi];
                            out[oi] = r > 0.0f ? r : 0.0f;
                        }
                    } else if (valid) out[pk_index<G>(co, pos)] = v;
                }
            }
        }
    }
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
    const float *ip[HT];
#pragma unroll
    for (int i = 0; i < HT; i++) {
        const int mt = wave + AZ_NW * i;
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ip[i] = lds + q * G::CS3 + (mt < G::MT ? mt : 0) * 16 + r16;
    }
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
        const float4 bq = wp4[s4 * 64];
        const float be[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int i = 0; i < HT; i++) acc[i] = mfma4(be[e], ip[i][(s4 * 16 + 4 * e) * G::CS3], acc[i]);
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
    conv_layer<G, 64, 128, CONV_OUT3>(inB, lds, w.c3, w.c3b, wpos, cellof, wave, lane);
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

// policy_fc (net.py:65) and value_fc1 + ReLU (net.py:69) for 16 boards per workgroup row.
template <class G>
__global__ __launch_bounds__(G::FCW * 64) void k_fc(DevState d, NetWeights w, int net_id, const float *__restrict__ feat,
                                                    unsigned long long *dbgfc)
{
    __shared__ __attribute__((aligned(16))) float ft[16 * G::FSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mb = blockIdx.x * 16;
    const int tile = blockIdx.y * G::FCW + wave;
    // stage the 16 boards' feature rows (16-byte vectors; rows are FROW floats, 16-B aligned)
#ifdef AZ_STAMPS
#define FC_STAMP(k) do { if (dbgfc && (threadIdx.x & 63) == 0) dbgfc[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FC_STAMP(k) do { } while (0)
#endif
    FC_STAMP(0);
    // nothing to do for a row of 16 boards without a pending evaluation (episode tails, terminal leaves)
    {
        __shared__ int fc_active;
        if (tid == 0) fc_active = 0;
        __syncthreads();
        if (tid < 16) {
            const int b = mb + tid;
            if (b < d.B) {
                const int kind = d.leaf_kind[b];
                if ((kind == LEAF_ROOT || kind == LEAF_EXPAND) && d.s_status[b] == SLOT_ACTIVE && d.s_net[b] == net_id)
                    atomicOr(&fc_active, 1);
            }
        }
        __syncthreads();
        if (!fc_active) return;
    }
    // the LDS tail [FROW, FSTR) is zeroed: the padded k-steps of the last weight group read it (times zero weights)
    constexpr int V = G::FROW / 4, VS = G::FSTR / 4;
    static_assert(G::FNEED <= G::FSTR, "feature tile too narrow");
    static_assert(16 * G::FSTR * 4 <= 150 * 1024, "k_fc feature tile exceeds LDS");
    constexpr int NTH = G::FCW * 64;
    constexpr int U = (16 * VS + NTH - 1) / NTH;
    float4 tmp[U];
#pragma unroll
    for (int u = 0; u < U; u++) {            // all loads in flight before the first LDS store
        const int idx = tid + NTH * u;
        const int i = idx / VS, c = idx - i * VS;
        const int b = mb + i;
        tmp[u] = float4{0.f, 0.f, 0.f, 0.f};
        if (idx < 16 * VS && b < d.B && c < V) tmp[u] = reinterpret_cast<const float4 *>(feat + (size_t)b * G::FROW)[c];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int idx = tid + NTH * u;
        const int i = idx / VS, c = idx - i * VS;
        if (idx < 16 * VS) reinterpret_cast<float4 *>(ft + i * G::FSTR)[c] = tmp[u];
    }
    __syncthreads();
    FC_STAMP(1);
    if (tile >= G::NTP + 4) return;
    const int q = lane >> 4, r16 = lane & 15;
    const bool is_pol = tile < G::NTP;
    const int KS = is_pol ? G::KSP : G::KSV;
    const int KS4 = (KS + 3) / 4;
    constexpr int CH = 8;                              // weight groups (of 4 k-steps) per chunk
    const int NCH = (KS4 + CH - 1) / CH;               // the packed weights are zero-padded to whole chunks of 16 groups
    const float4 *wp4 = reinterpret_cast<const float4 *>(is_pol ? w.pf : w.vf) +
                        (size_t)(is_pol ? tile : tile - G::NTP) * (is_pol ? G::KS4P_PAD : G::KS4V_PAD) * 64 + lane;
    const float *ip = ft + r16 * G::FSTR + (is_pol ? 0 : G::PC * G::nn) + q;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // One k-ordered MFMA chain.  Both operand streams are double-buffered a whole chunk (32 MFMAs) ahead: the
    // weight fragments come from L2, the feature fragments from LDS; the sched_barrier pins the loads above the chain.
    float4 bcur[CH], bnxt[CH];
    float acur[CH][4], anxt[CH][4];
#pragma unroll
    for (int j = 0; j < CH; j++) {
        bcur[j] = wp4[(size_t)j * 64];
#pragma unroll
        for (int e = 0; e < 4; e++) acur[j][e] = ip[j * 16 + e * 4];
    }
    for (int c = 0; c < NCH; c++) {
        const int cn = c + 1 < NCH ? c + 1 : c;
        const float *ipn = ip + cn * CH * 16;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            bnxt[j] = wp4[(size_t)(cn * CH + j) * 64];
#pragma unroll
            for (int e = 0; e < 4; e++) anxt[j][e] = ipn[j * 16 + e * 4];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; j++) {
            acc = mfma4(acur[j][0], bcur[j].x, acc);
            acc = mfma4(acur[j][1], bcur[j].y, acc);
            acc = mfma4(acur[j][2], bcur[j].z, acc);
            acc = mfma4(acur[j][3], bcur[j].w, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; j++) {
            bcur[j] = bnxt[j];
#pragma unroll
            for (int e = 0; e < 4; e++) acur[j][e] = anxt[j][e];
        }
    }
    FC_STAMP(2);
    // only slots with a pending evaluation are written: the row of a slot whose evaluation came from the cache (or that
    // waits on a terminal leaf) must stay as it is
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
        int b = mb + q * 4 + rg;
        if (b < d.B && d.s_net[b] == net_id && leaf_needs_net(d.leaf_kind[b]) && d.s_status[b] == SLOT_ACTIVE) {
            if (is_pol) {
                int j = tile * 16 + r16;
                if (j < G::nn) d.logits[(size_t)b * G::RW + j] = acc[rg] + w.pfb[j];
            } else {
                int i = (tile - G::NTP) * 16 + r16;
                float v = acc[rg] + w.vfb[i];
                d.vhid[(size_t)b * 64 + i] = v > 0.0f ? v : 0.0f;
            }
        }
    }
}
