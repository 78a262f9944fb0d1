// az_net_bf3.h -- opt-in fp32-EMULATING conv trunk for GomokuNet (net.py:55-72) on the bf16 matrix cores of gfx950.
//
// The f32 MFMA (v_mfma_f32_16x16x4_f32, az_net.h) runs at the vector rate; the bf16 MFMA (v_mfma_f32_16x16x32_bf16) moves
// 16 x the K per cycle.  Here every conv operand x is split into three bfloat16 parts, x = hi + mid + lo (8 + 8 + 8
// mantissa bits, each part the round-to-nearest bf16 of what the previous parts left over), and a product w * a is
// accumulated in float32 from the six largest of the nine cross products:
//     w_lo a_hi + w_hi a_lo + w_mid a_mid + w_mid a_hi + w_hi a_mid + w_hi a_hi        (dropped: <= 2^-24 relative)
// i.e. 6 bf16 MFMAs per 16x16 tile and 32 k instead of 8 f32 MFMAs: a 2.67 x ceiling over k_trunk at fp32-like accuracy.
// NOT the canonical fp order: results agree with the oracle within a tolerance (tests/test_bf16x3_gpu.py: logits 2e-5,
// P 1e-6, value 2e-6 -- the tolerances the build already grants against the Python reference's torch numbers), not bit for
// bit, so the mode is never the default (az_set_trunk_mode).  conv1 (K = 36, 1 % of the work) and the 1x1 head convs stay
// on the f32 MFMA; conv2 and conv3 (99 %) run here.
//
// LDS images: an activation image is a list of PLANES of CS 16-byte slots; plane (split * C/8 + ci/8), slot = position in
// the zero-padded board image, 8 consecutive channels per slot -- exactly the B fragment of one lane (k = 8 (lane >> 4) + j),
// so one ds_read_b128 per split feeds an MFMA, and the 16 lanes of a fragment row read 16 consecutive slots (CS % 16 == 0:
// conflict-free).  The 32-channel image (conv1 out) occupies planes 0..11 of the region the 64-channel image (conv2 out,
// 24 planes) later overlays; the padding rings coincide, so they are zeroed once per board.  conv2 and conv3 keep their
// outputs in accumulators until every wave has finished reading the inputs (one barrier), then write over them.
#pragma once
#include "az_net.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef AZ_BF3_NTW
#define AZ_BF3_NTW 1          // output-channel tiles per wave in conv3 (1: a wave owns one channel tile x all cell tiles)
#endif

__device__ __forceinline__ f32x4 mfma_bf(const uint4 &a, const uint4 &b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

enum { BF3_OUT_IMAGE = 0, BF3_OUT3 = 1 };

// One 3x3 conv layer, D[co][cell] = sum_k W[co][k] X[k][cell], k = tap * CIN + ci, on the bf16 MFMA with both operands
// split three ways.  A wave owns NTW channel tiles x MTW cell tiles; per K-block of 32 (one tap, 32 channels) and cell
// tile it reads three activation fragments from LDS (prefetched one tile ahead) and issues 6 * NTW MFMAs; the weight
// fragments (packed [tile][K-block][split][lane][8] at az_load_weights) come from L2 one K-block ahead.
// MODE BF3_OUT_IMAGE: barrier, relu(acc + bias) split into the C = COUT image at out;  BF3_OUT3: barrier, relu(acc + bias)
// as float32 [co][cell] (stride CS3) for the head convs.
template <class G, int CIN, int COUT, int MODE, int NTW_>
__device__ __forceinline__ void conv_layer_bf3(const uint4 *in, void *out, const uint4 *__restrict__ wp,
                                               const float *__restrict__ bias, const unsigned short *wpos,
                                               const unsigned short *cellof, int wave, int lane)
{
    constexpr int NT = COUT / 16;
    constexpr int NTW = NTW_ <= NT ? NTW_ : NT;
    constexpr int NG = NT / NTW;
    constexpr int MG = (G::NW / NG) > 0 ? (G::NW / NG) : 1;
    constexpr int MTW = (G::MT + MG - 1) / MG;
    constexpr int NCG = CIN / 8;               // 8-channel planes per split
    constexpr int KBT = CIN / 32;              // K-blocks per tap
    constexpr int KB = 9 * KBT;
    static_assert(NG * MG == G::NW, "wave grid does not cover the workgroup");
    const int ng = wave % NG, mg = wave / NG;
    const int q = lane >> 4, r16 = lane & 15;

    f32x4 acc[NTW][MTW];
#pragma unroll
    for (int t = 0; t < NTW; t++)
#pragma unroll
        for (int i = 0; i < MTW; i++) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int ra[MTW];                               // slot of the window's top-left corner in plane q
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        const int mt = mg + i * MG;
        const int m = (mt < G::MT ? mt : 0) * 16 + r16;        // a surplus tile aliases tile 0 (computed, never written back)
        ra[i] = (int)wpos[m] - (G::PW + 1) + q * G::CS;
    }
    const uint4 *wl[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) wl[t] = wp + (size_t)(ng * NTW + t) * KB * 3 * 64 + lane;
    uint4 wc[NTW][3], wn[NTW][3];
#pragma unroll
    for (int t = 0; t < NTW; t++)
#pragma unroll
        for (int s = 0; s < 3; s++) wc[t][s] = wl[t][s * 64];
    uint4 a0[3], a1[3];
#pragma unroll
    for (int s = 0; s < 3; s++) a0[s] = in[ra[0] + s * NCG * G::CS];
    for (int tap = 0; tap < 9; tap++) {
        const int toff = (tap / 3) * G::PW + (tap % 3);
        const int tn = tap + 1 < 9 ? tap + 1 : tap;
        const int toffn = (tn / 3) * G::PW + (tn % 3);
#pragma unroll
        for (int sq = 0; sq < KBT; sq++) {
            const int kb = tap * KBT + sq;
            const int kn = kb + 1 < KB ? kb + 1 : kb;
#pragma unroll
            for (int t = 0; t < NTW; t++)
#pragma unroll
                for (int s = 0; s < 3; s++) wn[t][s] = wl[t][(size_t)(kn * 3 + s) * 64];
            const int off = sq * 4 * G::CS + toff;
            const int offn = sq + 1 < KBT ? (sq + 1) * 4 * G::CS + toff : toffn;      // first tile of the next K-block
#pragma unroll
            for (int i = 0; i < MTW; i++) {
                uint4 *cur = (i & 1) ? a1 : a0;
                uint4 *nxt = (i & 1) ? a0 : a1;
                const int an = i + 1 < MTW ? ra[i + 1] + off : ra[0] + offn;
#pragma unroll
                for (int s = 0; s < 3; s++) nxt[s] = in[an + s * NCG * G::CS];
#pragma unroll
                for (int t = 0; t < NTW; t++) {
                    acc[t][i] = mfma_bf(wc[t][2], cur[0], acc[t][i]);       // small terms first
                    acc[t][i] = mfma_bf(wc[t][0], cur[2], acc[t][i]);
                    acc[t][i] = mfma_bf(wc[t][1], cur[1], acc[t][i]);
                    acc[t][i] = mfma_bf(wc[t][1], cur[0], acc[t][i]);
                    acc[t][i] = mfma_bf(wc[t][0], cur[1], acc[t][i]);
                    acc[t][i] = mfma_bf(wc[t][0], cur[0], acc[t][i]);
                }
            }
            if constexpr ((MTW & 1) != 0) {        // an odd number of tiles leaves the prefetched fragments in a1
#pragma unroll
                for (int s = 0; s < 3; s++) a0[s] = a1[s];
            }
#pragma unroll
            for (int i = 0; i < MTW; i++)
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTW, 0);   // 2*NTW MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // 1 LDS read (b128)
                }
#pragma unroll
            for (int t = 0; t < NTW; t++)
#pragma unroll
                for (int s = 0; s < 3; s++) wc[t][s] = wn[t][s];
        }
    }
    __syncthreads();                           // every wave has finished reading the input image: the output overlays it
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int nt = ng * NTW + t;
        float bco[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bco[rg] = bias[nt * 16 + q * 4 + rg];
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            const int mt = mg + i * MG;
            if (mt < G::MT) {
                const int m = mt * 16 + r16;
                float v[4];
#pragma unroll
                for (int rg = 0; rg < 4; rg++) {
                    const float x = acc[t][i][rg] + bco[rg];
                    v[rg] = x > 0.0f ? x : 0.0f;
                }
                if constexpr (MODE == BF3_OUT3) {
                    float *o = reinterpret_cast<float *>(out);
#pragma unroll
                    for (int rg = 0; rg < 4; rg++) o[(nt * 16 + q * 4 + rg) * G::CS3 + m] = v[rg];
                } else {
                    if (cellof[m] != 0xFFFFu) store_bf3<G, COUT>(reinterpret_cast<uint2 *>(out), nt * 16 + q * 4, wpos[m], v);
                }
            }
        }
    }
}

template <int N>
__global__ __launch_bounds__(AZ_NW * 64) void k_trunk_bf3(DevState d, NetWeights w, int net_id, float *__restrict__ feat,
                                                          unsigned long long *dbg)
{
    typedef NetGeo<N> G;
    constexpr int NTH = AZ_NW * 64;
    __shared__ __attribute__((aligned(16))) float lds[G::LDSF];
    __shared__ unsigned short wpos[G::MR];
    __shared__ unsigned short cellof[G::MR];
    __shared__ int any_active;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.x, b0 = grp * G::G;
    AZ_STAMP(0);
    if (tid == 0) any_active = 0;
    __syncthreads();
    if (tid < G::G) {
        const int b = b0 + tid;
        if (b < d.B) {
            const int kind = d.leaf_kind[b];
            if (leaf_needs_net(kind) && d.s_status[b] == SLOT_ACTIVE && d.s_net[b] == net_id) atomicOr(&any_active, 1);
        }
    }
    __syncthreads();
    if (!any_active) return;
    // region X = 24 planes of CS slots (96 * CS floats): conv1 out in planes 0..11, conv2 out in planes 0..23; the float32
    // input planes of conv1 are the bytes of plane 12
    float *inP = lds + 48 * G::CS;
    {
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < (96 * G::CS) / 4; i += NTH) z[i] = float4{0.f, 0.f, 0.f, 0.f};
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
    for (int m = tid; m < G::MR; m += NTH) {          // games.py:86-129 encode
        const int cell = cellof[m];
        if (cell != 0xFFFF) {
            const int g = cell / G::nn, p = cell - g * G::nn;
            const int b = b0 + g;
            if (b < d.B) {
                const u64 *lf = d.leaf + (size_t)b * 8;
                const int pos = wpos[m];
                if ((lf[p >> 6] >> (p & 63)) & 1ull) inP[pos] = 1.0f;
                if ((lf[4 + (p >> 6)] >> (p & 63)) & 1ull) inP[G::CS + pos] = 1.0f;
                if (d.leaf_last[b] == p) inP[2 * G::CS + pos] = 1.0f;
            }
        }
    }
    __syncthreads();
    AZ_STAMP(1);
    conv_layer<G, 4, 32, CONV_OUT_BF3>(inP, lds, w.c1, w.c1b, wpos, cellof, wave, lane);
    __syncthreads();
    for (int i = tid; i < 3 * G::CS; i += NTH) inP[i] = 0.0f;      // plane 12 is part of conv2's output image
    AZ_STAMP(2);
    conv_layer_bf3<G, 32, 64, BF3_OUT_IMAGE, 1>(reinterpret_cast<const uint4 *>(lds), lds, reinterpret_cast<const uint4 *>(w.c2x),
                                                w.c2b, wpos, cellof, wave, lane);
    __syncthreads();
    AZ_STAMP(3);
    conv_layer_bf3<G, 64, 128, BF3_OUT3, AZ_BF3_NTW>(reinterpret_cast<const uint4 *>(lds), lds, reinterpret_cast<const uint4 *>(w.c3x),
                                                     w.c3b, wpos, cellof, wave, lane);
    __syncthreads();
    AZ_STAMP(4);
    trunk_heads<G>(d, w, net_id, feat, lds, cellof, b0, wave, lane);
    AZ_STAMP(5);
}
