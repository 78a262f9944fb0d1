// az_net.h -- GomokuNet forward (net.py:55-72) as two HIP kernels for gfx950.
//
//  k_trunk : leaf encode (games.py:86-129) -> conv1 -> conv2 -> conv3 -> policy/value 1x1 convs, one
//            workgroup per group of G boards, every activation resident in LDS (zero-padded planes, so
//            the 3x3 im2col is "base + constant offset"), contraction on v_mfma_f32_16x16x4_f32.
//  k_fc    : policy_fc and value_fc1 as a batched GEMM over boards, same MFMA.
// softmax / value_fc2 / tanh are fused into the tree kernel that consumes them (az_tree.h).
//
// Numerics: every output element is ONE k-ordered fp32 fma chain from +0 (what the f32 MFMA computes,
// bit for bit), bias added afterwards.  conv k = (ky*3+kx)*Cin + ci; FC k ascending.
#pragma once
#include "az_tree.h"

struct NetWeights {
    // MFMA-fragment packed (see pack_* in az_engine.hip): [ntile][kstep/4][lane][4]
    const float *c1, *c2, *c3, *hd, *pf, *vf;
    const float *c1b, *c2b, *c3b, *hdb, *pfb, *vfb;   // biases (hdb: 4 policy_conv + 2 value_conv)
};

__host__ __device__ constexpr int up17(int x) { return x + ((17 - (x % 32) + 32) % 32); }

template <int N>
struct NetGeo {
    static constexpr int n = N, nn = N * N, PW = N + 2, PP = PW * PW;
    static constexpr int G = N == 15 ? 1 : (N == 9 ? 3 : 7);     // boards per workgroup
    static constexpr int M = G * nn;                               // real GEMM rows (board cells)
    static constexpr int MT = (M + 15) / 16, MR = MT * 16;         // 16-row MFMA tiles
    static constexpr int CS = up17(G * PP);                        // channel stride of padded planes (floats)
    static constexpr int CS3 = up17(MR);                           // channel stride of the conv3 output image
    static constexpr int LDSF = (96 * CS > 128 * CS3) ? 96 * CS : 128 * CS3;
    static constexpr int RW = ((nn + 63) / 64) * 64;
    // FC kernel
    static constexpr int NTP = (nn + 15) / 16;                     // policy N-tiles
    static constexpr int KSP = nn;                                 // policy k-steps (4nn / 4)
    static constexpr int KSV = (2 * nn + 3) / 4;                   // value_fc1 k-steps
    static constexpr int FSTR = up17(6 * nn + 4);                  // LDS row stride of the feature tile
    static constexpr int NSPLIT = (NTP + 4 + 3) / 4;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One conv layer on the workgroup's LDS image.  OUT3=false: write relu(acc+bias) into the next padded image.
// OUT3=true (conv3): barrier, then write into the [co][m] image that overlays the (now dead) inputs.
template <int N, int CIN, int COUT, bool OUT3>
__device__ __forceinline__ void conv_layer(const float *in, float *out, const float *__restrict__ wp,
                                           const float *__restrict__ bias, const unsigned short *wpos, int wave,
                                           int lane)
{
    typedef NetGeo<N> G;
    constexpr int NT = COUT / 16;          // N tiles
    constexpr int MG = 8 / NT;             // M groups (8 waves)
    constexpr int MTW = (G::MT + MG - 1) / MG;
    constexpr int KST = CIN / 4;           // k-steps per tap
    constexpr int KS = 9 * KST;
    constexpr int KS4 = (KS + 3) / 4;
    const int nt = wave % NT, mg = wave / NT;
    const int q = lane >> 4, r16 = lane & 15;

    int rb[MTW];
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        int mt = mg + i * MG;
        int m = (mt < G::MT ? mt : 0) * 16 + r16;
        rb[i] = (int)wpos[m] - (G::PW + 1) + q * G::CS;
    }
    f32x4 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float4 *wp4 = reinterpret_cast<const float4 *>(wp) + (size_t)nt * KS4 * 64 + lane;
    if constexpr (CIN == 4) {
        // conv1: 9 k-steps (one per tap), channels = {mover, opponent, last move, zero plane}
        float4 b0 = wp4[0], b1 = wp4[64], b2 = wp4[128];
        float bk[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int toff = (tap / 3) * G::PW + (tap % 3);
#pragma unroll
            for (int i = 0; i < MTW; i++)
                if (mg + i * MG < G::MT) acc[i] = mfma4(in[rb[i] + toff], bk[tap], acc[i]);
        }
    } else {
        constexpr int NQ = KST / 4;        // float4 weight groups per tap
        float4 bq = wp4[0];
        for (int tap = 0; tap < 9; tap++) {
            const int toff = (tap / 3) * G::PW + (tap % 3);
            for (int sq = 0; sq < NQ; sq++) {
                const int it = tap * NQ + sq;
                const float4 bn = wp4[(size_t)(it + 1 < 9 * NQ ? it + 1 : it) * 64];
                const float *ip = in + toff + sq * 16 * G::CS;
                const float be[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
#pragma unroll
                    for (int i = 0; i < MTW; i++)
                        if (mg + i * MG < G::MT) acc[i] = mfma4(ip[rb[i] + e * 4 * G::CS], be[e], acc[i]);
                }
                bq = bn;
            }
        }
    }
    const int co = nt * 16 + r16;
    const float bco = bias[co];
    if constexpr (OUT3) __syncthreads();   // every wave has finished reading the conv3 input image
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        int mt = mg + i * MG;
        if (mt < G::MT) {
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                int m = mt * 16 + q * 4 + rg;
                float v = acc[i][rg] + bco;
                v = v > 0.0f ? v : 0.0f;
                if constexpr (OUT3) out[co * G::CS3 + m] = v;
                else if (m < G::M) out[co * G::CS + wpos[m]] = v;
            }
        }
    }
}

template <int N>
__global__ __launch_bounds__(512) void k_trunk(DevState d, NetWeights w, int net_id, float *__restrict__ pol_feat,
                                               float *__restrict__ val_feat)
{
    typedef NetGeo<N> G;
    __shared__ __attribute__((aligned(16))) float lds[G::LDSF];
    __shared__ unsigned short wpos[G::MR];
    __shared__ int any_active;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    float *inA = lds;                  // 32 channels (conv1 out / conv2 in)
    float *inB = lds + 32 * G::CS;     // 64 channels (conv2 out / conv3 in); first 4 channels hold the input planes
    // zero both padded images (the padding ring must read as 0)
    {
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < (96 * G::CS) / 4; i += 512) z[i] = float4{0.f, 0.f, 0.f, 0.f};
        if (tid < (96 * G::CS) % 4) lds[(96 * G::CS) - 1 - tid] = 0.0f;
    }
    for (int m = tid; m < G::MR; m += 512) {
        int g = m / G::nn, p = m - g * G::nn;
        int r = p / N, c = p - r * N;
        wpos[m] = (unsigned short)(m < G::M ? g * G::PP + (r + 1) * G::PW + (c + 1) : G::PW + 1);
    }
    __syncthreads();
    // games.py:86-129 encode: ch0 = side to move, ch1 = opponent, ch2 = last action, ch3 = zeros
    for (int m = tid; m < G::M; m += 512) {
        int g = m / G::nn, p = m - g * G::nn;
        int b = b0 + g;
        if (b < d.B) {
            const u64 *lf = d.leaf + (size_t)b * 8;
            bool me = (lf[p >> 6] >> (p & 63)) & 1ull;
            bool op = (lf[4 + (p >> 6)] >> (p & 63)) & 1ull;
            int pos = wpos[m];
            if (me) inB[pos] = 1.0f;
            if (op) inB[G::CS + pos] = 1.0f;
            if (d.leaf_last[b] == p) inB[2 * G::CS + pos] = 1.0f;
        }
    }
    __syncthreads();
    conv_layer<N, 4, 32, false>(inB, inA, w.c1, w.c1b, wpos, wave, lane);
    __syncthreads();
    conv_layer<N, 32, 64, false>(inA, inB, w.c2, w.c2b, wpos, wave, lane);
    __syncthreads();
    conv_layer<N, 64, 128, true>(inB, lds, w.c3, w.c3b, wpos, wave, lane);
    __syncthreads();
    // policy_conv (128->4) and value_conv (128->2), 1x1, as one 16-column MFMA GEMM over the conv3 image
    {
        const int q = lane >> 4, r16 = lane & 15;
        const float4 *wp4 = reinterpret_cast<const float4 *>(w.hd) + lane;
        const float hb = r16 < 6 ? w.hdb[r16] : 0.0f;
        for (int mt = wave; mt < G::MT; mt += 8) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const float *ip = lds + q * G::CS3 + mt * 16 + r16;
#pragma unroll
            for (int s4 = 0; s4 < 8; s4++) {
                float4 bq = wp4[s4 * 64];
                acc = mfma4(ip[(s4 * 16 + 0) * G::CS3], bq.x, acc);
                acc = mfma4(ip[(s4 * 16 + 4) * G::CS3], bq.y, acc);
                acc = mfma4(ip[(s4 * 16 + 8) * G::CS3], bq.z, acc);
                acc = mfma4(ip[(s4 * 16 + 12) * G::CS3], bq.w, acc);
            }
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                int m = mt * 16 + q * 4 + rg;
                if (m < G::M && r16 < 6) {
                    int g = m / G::nn, p = m - g * G::nn;
                    int b = b0 + g;
                    if (b < d.B && d.s_net[b] == net_id) {
                        float v = acc[rg] + hb;
                        v = v > 0.0f ? v : 0.0f;
                        if (r16 < 4) pol_feat[(size_t)b * 4 * G::nn + r16 * G::nn + p] = v;
                        else val_feat[(size_t)b * 2 * G::nn + (r16 - 4) * G::nn + p] = v;
                    }
                }
            }
        }
    }
}

// policy_fc (net.py:65) and value_fc1 + ReLU (net.py:69) for 16 boards per workgroup row.
template <int N>
__global__ __launch_bounds__(256) void k_fc(DevState d, NetWeights w, int net_id, const float *__restrict__ pol_feat,
                                            const float *__restrict__ val_feat)
{
    typedef NetGeo<N> G;
    __shared__ __attribute__((aligned(16))) float ft[16 * G::FSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mb = blockIdx.x * 16;
    const int tile = blockIdx.y * 4 + wave;
    // stage the 16 boards' features: [i][0..4nn) policy features, [4nn..6nn) value features, zero tail
    for (int idx = tid; idx < 16 * G::FSTR; idx += 256) {
        int i = idx / G::FSTR, k = idx - i * G::FSTR;
        int b = mb + i;
        float v = 0.0f;
        if (b < d.B) {
            if (k < 4 * G::nn) v = pol_feat[(size_t)b * 4 * G::nn + k];
            else if (k < 6 * G::nn) v = val_feat[(size_t)b * 2 * G::nn + (k - 4 * G::nn)];
        }
        ft[idx] = v;
    }
    __syncthreads();
    if (tile >= G::NTP + 4) return;
    const int q = lane >> 4, r16 = lane & 15;
    const bool is_pol = tile < G::NTP;
    const int KS = is_pol ? G::KSP : G::KSV;
    const int KS4 = (KS + 3) / 4;
    const float4 *wp4 = reinterpret_cast<const float4 *>(is_pol ? w.pf : w.vf) +
                        (size_t)(is_pol ? tile : tile - G::NTP) * KS4 * 64 + lane;
    const float *ip = ft + r16 * G::FSTR + (is_pol ? 0 : 4 * G::nn) + q;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bq = wp4[0];
    for (int s4 = 0; s4 < KS4; s4++) {
        const float4 bn = wp4[(size_t)(s4 + 1 < KS4 ? s4 + 1 : s4) * 64];
        const int s = s4 * 4;
        acc = mfma4(ip[(s + 0) * 4], bq.x, acc);
        if (s + 1 < KS) acc = mfma4(ip[(s + 1) * 4], bq.y, acc);
        if (s + 2 < KS) acc = mfma4(ip[(s + 2) * 4], bq.z, acc);
        if (s + 3 < KS) acc = mfma4(ip[(s + 3) * 4], bq.w, acc);
        bq = bn;
    }
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
        int b = mb + q * 4 + rg;
        if (b < d.B && d.s_net[b] == net_id) {
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
