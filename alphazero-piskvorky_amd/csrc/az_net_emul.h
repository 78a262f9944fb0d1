// az_net_emul.h -- opt-in fp32-EMULATING conv trunks (GomokuNet, net.py:55-72, and the ResidualBlock variant) on the 16-bit
// matrix cores of gfx950.
//
// The f32 MFMA (v_mfma_f32_16x16x4_f32, az_net.h) runs at the vector rate; the 16-bit MFMAs (v_mfma_f32_16x16x32_bf16 / _f16)
// move 16 x the K per cycle.  Every conv operand is split into 16-bit parts and a product w * a is accumulated in float32
// from the largest cross products.  Two schemes (az_set_trunk_mode):
//   AZ_TRUNK_BF16X3  x = hi + mid + lo, three bfloat16 parts (8 + 8 + 8 mantissa bits, each the round-to-nearest bf16 of what
//                    the previous parts left over); six of the nine cross products:
//                        w_lo a_hi + w_mid a_hi + w_hi a_hi + w_mid a_mid + w_hi a_mid + w_hi a_lo     (dropped: <= 2^-24 relative)
//                    6 MFMAs per 16x16 tile and 32 k against 8 f32 MFMAs: a 2.67 x ceiling.  Keeps float32's exponent range.
//   AZ_TRUNK_F16X2   x = hi + lo / 2048, two float16 parts (11 + 11 mantissa bits; lo is stored scaled by 2^11 so that it never
//                    falls into float16's subnormals); three of the four cross products, w_hi a_hi into one accumulator and
//                    w_lo a_hi + w_hi a_lo into a second one that is folded in as acc + accx / 2048 (dropped: 2^-22 relative):
//                    3 MFMAs per tile, a 5.3 x ceiling.  float16's range: activations are saturated at 65504 (+inf too; a NaN
//                    activation stays NaN and reaches the priors, as in the float32 and bf16x3 trunks) and weights must
//                    stay below it (checked at az_load_weights) -- a net beyond that belongs to the other modes.
// NOT the canonical fp order: results agree with the oracle within a tolerance (tests/test_emulated_trunk_gpu.py: logits 2e-5,
// P 1e-6, value 2e-6 -- the tolerances the build already grants against the Python reference's torch numbers), not bit for
// bit, so neither is ever the default.  Every conv and the 1x1 head convs run on the 16-bit MFMA (the first conv's 0 / 1
// input planes are exact in one part, so only its weights are split).
//
// LDS images: an activation image is a list of PLANES of CS 16-byte slots; plane (NS * (ci/8) + part), slot = position in the
// zero-padded board image, 8 consecutive channels per slot -- exactly the B fragment of one lane (k = 8 (lane >> 4) + j), so
// one ds_read_b128 per part feeds an MFMA, and the 16 lanes of a fragment row read 16 consecutive slots (CS % 16 == 0:
// conflict-free).  The 32-channel image (conv1 out) occupies the first planes of the region the 64-channel image (conv2 out)
// later overlays; the padding rings coincide, so they are zeroed once per board.  Every conv keeps its outputs in
// accumulators until every wave has finished reading the inputs (one barrier), then writes over them.
//
// A wave owns ONE 16-channel tile of a layer and its share of the 16-cell tiles (all of them in conv3).  Measured at n = 15
// (bf16x3) and rejected: two channel tiles per wave (halves the LDS reads, 6 % surplus cell tiles: 49.8 us per 256 boards
// against 47.6), the MFMAs of two or three cell tiles interleaved (49.2 / 49.7 us).
#pragma once
#include "az_net.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

enum { EMUL_BF16X3 = 1, EMUL_F16X2 = 2 };      // = AZ_TRUNK_BF16X3, AZ_TRUNK_F16X2 (include/az_engine.h)

// two floats -> packed bf16 pair (round to nearest even; element 0 in the low half)
__device__ __forceinline__ unsigned pk_bf16(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xFFFF0000u); }
// two floats -> packed float16 pair (round to nearest even)
__device__ __forceinline__ unsigned pk_f16(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, f16x2));
}
__device__ __forceinline__ f32x2 unpk_f16(unsigned p) { return __builtin_convertvector(__builtin_bit_cast(f16x2, p), f32x2); }

// What a scheme is made of: NS parts per operand (= planes per channel group, fragments per K-block), NP MFMA products per
// tile and K-block with their (weight part, activation part, accumulator) triples in issue order -- the activation's first
// part, the fragment read first, is used first --, the split of four consecutive channels, and the two MFMAs.
template <int SCH> struct Emul;
template <> struct Emul<EMUL_BF16X3> {
    static constexpr int NS = 3, NP = 6, NACC = 1;
    __device__ static constexpr int ws(int p) { constexpr int t[6] = {2, 1, 0, 1, 0, 0}; return t[p]; }
    __device__ static constexpr int as(int p) { constexpr int t[6] = {0, 0, 0, 1, 1, 2}; return t[p]; }
    __device__ static constexpr int acc(int) { return 0; }
    __device__ static __forceinline__ void split(const float (&v)[4], uint2 (&s)[3])
    {
        const unsigned h0 = pk_bf16(v[0], v[1]), h1 = pk_bf16(v[2], v[3]);
        const float r0 = v[0] - bf_lo(h0), r1 = v[1] - bf_hi(h0), r2 = v[2] - bf_lo(h1), r3 = v[3] - bf_hi(h1);     // exact
        const unsigned m0 = pk_bf16(r0, r1), m1 = pk_bf16(r2, r3);
        const unsigned l0 = pk_bf16(r0 - bf_lo(m0), r1 - bf_hi(m0)), l1 = pk_bf16(r2 - bf_lo(m1), r3 - bf_hi(m1));
        s[0] = uint2{h0, h1};
        s[1] = uint2{m0, m1};
        s[2] = uint2{l0, l1};
    }
    __device__ static __forceinline__ f32x4 mfma32(const uint4 &a, const uint4 &b, f32x4 c)      // k = 8 (lane >> 4) + j
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x4 mfma16(const uint2 &a, const uint2 &b, f32x4 c)      // k = 4 (lane >> 4) + j
    {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ float fold(float a, float) { return a; }
    static constexpr unsigned ONE = 0x3F80u;       // 1.0 in bfloat16
};
template <> struct Emul<EMUL_F16X2> {
    static constexpr int NS = 2, NP = 3, NACC = 2;
    __device__ static constexpr int ws(int p) { constexpr int t[3] = {0, 1, 0}; return t[p]; }      // w_hi a_hi | w_lo a_hi, w_hi a_lo
    __device__ static constexpr int as(int p) { constexpr int t[3] = {0, 0, 1}; return t[p]; }
    __device__ static constexpr int acc(int p) { return p == 0 ? 0 : 1; }
    __device__ static __forceinline__ void split(const float (&v)[4], uint2 (&s)[2])
    {
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = v[i] > 65504.0f ? 65504.0f : v[i];         // float16's range (inputs are post-ReLU: >= 0); +inf saturates, NaN stays NaN like in the other trunks
        const unsigned h0 = pk_f16(x[0], x[1]), h1 = pk_f16(x[2], x[3]);
        const f32x2 f0 = unpk_f16(h0), f1 = unpk_f16(h1);
        // the remainder is exact in float32; scaled by 2^11 it is a normal float16 whenever it matters
        const unsigned l0 = pk_f16((x[0] - f0[0]) * 2048.0f, (x[1] - f0[1]) * 2048.0f);
        const unsigned l1 = pk_f16((x[2] - f1[0]) * 2048.0f, (x[3] - f1[1]) * 2048.0f);
        s[0] = uint2{h0, h1};
        s[1] = uint2{l0, l1};
    }
    __device__ static __forceinline__ f32x4 mfma32(const uint4 &a, const uint4 &b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x4 mfma16(const uint2 &a, const uint2 &b, f32x4 c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ float fold(float a, float ax) { return a + ax * (1.0f / 2048.0f); }
    static constexpr unsigned ONE = 0x3C00u;       // 1.0 in float16
};

// v[0..3] = channels co .. co+3 (co % 4 == 0) at padded position pos -> the part planes of channel group co / 8
template <class G, int SCH>
__device__ __forceinline__ void store_parts(uint2 *img, int co, int pos, const float (&v)[4])
{
    typedef Emul<SCH> E;
    uint2 s[E::NS];
    E::split(v, s);
    uint2 *p = img + (((co >> 3) * E::NS * G::CS + pos) << 1) + ((co >> 2) & 1);    // 8-byte units: slot * 2 + half
#pragma unroll
    for (int i = 0; i < E::NS; i++) p[G::CS * 2 * i] = s[i];
}

// What a wave needs from global memory before its first MFMA of a layer, requested long before (at the top of the kernel
// or a layer ahead) so that no L2 round trip sits on the critical path between the layers.
template <int SCH>
struct EmulPre {
    uint4 c1w[2][Emul<SCH>::NS];   // conv1: the two K-blocks of the wave's channel tile, one fragment per part
    float b1[4], b2[4], b3[4];   // biases of the wave's channel tile in conv1 / conv2 / conv3, rows 4 (lane >> 4) .. + 3
    uint2 hw[Emul<SCH>::NS];     // head-conv fragments of the wave's conv3 tile, one per part
};

// The first conv (4 -> COUT, K = 36 = 9 taps x 4 planes: GomokuNet's conv1, the ResidualBlock net's stem) on the 16-bit MFMA.
// Its input -- the encode planes of games.py:86-129 -- is 0 / 1, exact in one 16-bit part, so only the weights are split: NS
// products per tile and K-block, K padded to 64.  The input image is [position][4 planes] of 16-bit values (8 bytes per
// padded position); lane (q, cell) holds k = 8 q + j = taps 2 q and 2 q + 1 of K-block 0 (two 8-byte reads at the taps'
// offsets) and tap 8 of K-block 1 (q = 0 only).  w: [tile][K-block][part][lane][8], zero beyond k = 36 (requested by the
// caller before the prologue).  Output split into the COUT-channel image; keep != nullptr also returns the wave's tiles as
// float32 (the skip connection's operand of the first residual block).
template <class G, int COUT, int SCH>
__device__ __forceinline__ void conv_first_emul(const uint2 *in, uint2 *out, const uint4 (&w)[2][Emul<SCH>::NS], const float (&b1)[4],
                                                const unsigned short *wpos, const unsigned short *cellof, int wave, int lane,
                                                f32x4 *keep = nullptr)
{
    typedef Emul<SCH> E;
    constexpr int NG = COUT / 16, MG = G::NW / NG, MTW = (G::MT + MG - 1) / MG;
    static_assert(NG * MG == G::NW, "wave grid does not cover the workgroup");
    const int ng = wave % NG, mg = wave / NG;
    const int q = lane >> 4, r16 = lane & 15;
    const int t0 = 2 * q, t1 = 2 * q + 1;
    const int off0 = (t0 / 3) * G::PW + (t0 % 3), off1 = (t1 / 3) * G::PW + (t1 % 3), off8 = 2 * G::PW + 2;
    f32x4 acc[E::NACC][MTW];
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        const int mt = mg + i * MG;
        const int m = (mt < G::MT ? mt : 0) * 16 + r16;        // a surplus tile aliases tile 0 (computed, never written back)
        const int base = (int)wpos[m] - (G::PW + 1);
        const uint2 a0 = in[base + off0], a1 = in[base + off1], a8 = in[base + off8];
        const uint4 f0 = uint4{a0.x, a0.y, a1.x, a1.y};
        const uint4 f1 = q == 0 ? uint4{a8.x, a8.y, 0u, 0u} : uint4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int a = 0; a < E::NACC; a++) acc[a][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = E::NS - 1; s >= 0; s--) {                  // small parts first; part s > 0 of the float16 scheme feeds the scaled accumulator
            const int ai = (E::NACC > 1 && s > 0) ? 1 : 0;
            acc[ai][i] = E::mfma32(w[0][s], f0, acc[ai][i]);
            acc[ai][i] = E::mfma32(w[1][s], f1, acc[ai][i]);
        }
    }
#pragma unroll
    for (int i = 0; i < MTW; i++) {
        const int mt = mg + i * MG;
        float v[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            const float x = E::fold(acc[0][i][rg], acc[E::NACC - 1][i][rg]) + b1[rg];
            v[rg] = x > 0.0f ? x : 0.0f;
        }
        if (keep) keep[i] = f32x4{v[0], v[1], v[2], v[3]};
        if (mt < G::MT) {
            const int m = mt * 16 + r16;
            if (cellof[m] != 0xFFFFu) store_parts<G, SCH>(out, ng * 16 + q * 4, wpos[m], v);
        }
    }
}

// the weight fragments of K-block kb of the wave's channel tile: packed [tile][K-block][part][lane][8 x 16 bit]
template <int KB, int NS>
__device__ __forceinline__ void emul_weights(const uint4 *__restrict__ wp, int tile, int kb, int lane, uint4 (&w)[NS])
{
    const uint4 *p = wp + ((size_t)(tile * KB + kb) * NS) * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; s++) w[s] = p[s * 64];
}

// Issue order of one K-block, pinned with sched_group_barrier: per cell tile one LDS read (a fragment of the NEXT tile)
// behind each of the first NS MFMAs, then the other NP - NS MFMAs.
template <int NR>
__device__ __forceinline__ void emul_sched_pairs()
{
    if constexpr (NR > 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 LDS read (b128)
        emul_sched_pairs<NR - 1>();
    }
}
template <int TILES, int NS, int NP>
__device__ __forceinline__ void emul_sched_tiles()
{
    if constexpr (TILES > 0) {
        emul_sched_pairs<NS>();
        if constexpr (NP > NS) __builtin_amdgcn_sched_group_barrier(0x008, NP - NS, 0);
        emul_sched_tiles<TILES - 1, NS, NP>();
    }
}

enum { EMUL_OUT_IMAGE = 0, EMUL_OUT_HEADS = 1,      // where the result goes
       EMUL_SKIP = 2,                                // relu(acc + bias + keep[tile]): the residual block's skip connection, from registers
       EMUL_KEEP = 4 };                              // keep[tile] = the layer's output (float32), the next block's skip operand

// One 3x3 conv layer, D[co][cell] = sum_k W[co][k] X[k][cell], k = tap * CIN + ci, on the 16-bit MFMA with both operands
// split into NS parts.  A wave owns one channel tile x MTW cell tiles; per K-block of 32 (one tap, 32 channels) and cell tile
// it reads NS activation fragments from LDS (a tile ahead) and issues NP MFMAs; the weight fragments of the next K-block
// are requested from L2 before the block's MFMAs (w0 = the first block's, requested by the caller a layer ahead).
// MODE EMUL_OUT_IMAGE: barrier, relu(acc + bias) split into the COUT-channel image at out.
// MODE EMUL_OUT_HEADS (conv3, every wave owns all cell tiles): the 1x1 head convs (net.py:64,69) straight from the
// accumulators -- a 16-channel x 16-cell result tile is, lane for lane, the B operand of the 16x16x16 MFMA
// (k = 4 (lane >> 4) + register), so each wave multiplies its channel tile with its 16 columns of the head weights
// (hw: [tile][part][lane][4 x 16 bit], rows = the NH head channels) and leaves partial sums [channel tile][head][cell] at out
// after the barrier; the caller adds the channel tiles up.
// the priority turns of the main loop apply unless the geometry opts out (ResGeoEmul::EMUL_PRIO_TURNS = false)
template <class G, class = void> struct emul_prio_turns : std::true_type {};
template <class G> struct emul_prio_turns<G, std::enable_if_t<!G::EMUL_PRIO_TURNS>> : std::false_type {};

template <class G, int CIN, int COUT, int MODE, int NH, int SCH>
__device__ __forceinline__ void conv_layer_emul(const uint4 *in, void *out, const uint4 *__restrict__ wp,
                                                const uint4 (&w0)[Emul<SCH>::NS], const float (&bco)[4],
                                                const uint2 (&hw)[Emul<SCH>::NS], const unsigned short *wpos,
                                                const unsigned short *cellof, int wave, int lane, f32x4 *keep = nullptr)
{
    typedef Emul<SCH> E;
    constexpr int NS = E::NS;
    constexpr int NG = COUT / 16;              // channel tiles = wave columns
    constexpr int MG = (G::NW / NG) > 0 ? (G::NW / NG) : 1;
    constexpr int MTW = (G::MT + MG - 1) / MG;
    constexpr int KBT = CIN / 32;              // K-blocks per tap
    constexpr int KB = 9 * KBT;
    static_assert(NG * MG == G::NW, "wave grid does not cover the workgroup");
    const int ng = wave % NG, mg = wave / NG;
    const int q = lane >> 4, r16 = lane & 15;

    f32x4 acc[E::NACC][MTW];
    int ra[MTW];                               // slot of the window's top-left corner in the lane's channel group (first part's plane)
#pragma unroll
    for (int i = 0; i < MTW; i++) {
#pragma unroll
        for (int a = 0; a < E::NACC; a++) acc[a][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int mt = mg + i * MG;
        const int m = (mt < G::MT ? mt : 0) * 16 + r16;        // a surplus tile aliases tile 0 (computed, never written back)
        ra[i] = (int)wpos[m] - (G::PW + 1) + q * NS * G::CS;
    }
    uint4 wc[NS], wn[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) wc[s] = w0[s];
    uint4 fa[2][NS];
#pragma unroll
    for (int s = 0; s < NS; s++) fa[0][s] = in[ra[0] + s * G::CS];
    for (int tap = 0; tap < 9; tap++) {      // rolled: unrolling the taps here measured nothing (the K-block loop inside is already clean)
        const int toff = (tap / 3) * G::PW + (tap % 3);
        const int tn = tap + 1 < 9 ? tap + 1 : tap;
        const int toffn = (tn / 3) * G::PW + (tn % 3);
#pragma unroll
        for (int sq = 0; sq < KBT; sq++) {
            const int kb = tap * KBT + sq;
#if AZ_PRIO_ALT
            // the waves sharing a SIMD take turns with the issue priority, K-block by K-block (az_net.h conv_layer: left alone,
            // one of them wins every arbitration and leaves its partner to finish the layer alone); measured: f16x2 8.24 -> 8.52 M
            // expansions/s, bf16x3 unchanged
            if constexpr (emul_prio_turns<G>::value) {
                if ((kb % (G::NW / 4)) == ((wave >> 2) % (G::NW / 4))) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
#endif
            emul_weights<KB, NS>(wp, ng, kb + 1 < KB ? kb + 1 : kb, lane, wn);
            __builtin_amdgcn_sched_barrier(0);         // the next K-block's weights are requested before this block's MFMAs, not after
            const int off = sq * 4 * NS * G::CS + toff;    // a K-block = 4 channel groups x NS part planes
            const int offn = sq + 1 < KBT ? (sq + 1) * 4 * NS * G::CS + toff : toffn;      // the next K-block
#pragma unroll
            for (int i = 0; i < MTW; i++) {
                uint4 *cur = fa[i & 1], *nxt = fa[(i & 1) ^ 1];
                const int an = i + 1 < MTW ? ra[i + 1] + off : ra[0] + offn;
#pragma unroll
                for (int s = 0; s < NS; s++) nxt[s] = in[an + s * G::CS];
#pragma unroll
                for (int p = 0; p < E::NP; p++) acc[E::acc(p)][i] = E::mfma32(wc[E::ws(p)], cur[E::as(p)], acc[E::acc(p)][i]);
            }
            if constexpr ((MTW & 1) != 0) {            // an odd number of tiles leaves the prefetched fragments in fa[1]
#pragma unroll
                for (int s = 0; s < NS; s++) fa[0][s] = fa[1][s];
            }
            emul_sched_tiles<MTW, NS, E::NP>();
#pragma unroll
            for (int s = 0; s < NS; s++) wc[s] = wn[s];
        }
    }
#if AZ_PRIO_ALT
    if constexpr (emul_prio_turns<G>::value) __builtin_amdgcn_s_setprio(0);
#endif
    // relu(acc + bias [+ skip]) of the wave's tiles
    float v[MTW][4];
#pragma unroll
    for (int i = 0; i < MTW; i++) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            float x = E::fold(acc[0][i][rg], acc[E::NACC - 1][i][rg]) + bco[rg];
            if constexpr ((MODE & EMUL_SKIP) != 0) x = x + keep[i][rg];
            v[i][rg] = x > 0.0f ? x : 0.0f;
        }
        if constexpr ((MODE & EMUL_KEEP) != 0) keep[i] = f32x4{v[i][0], v[i][1], v[i][2], v[i][3]};
    }
    if constexpr ((MODE & EMUL_OUT_HEADS) != 0) {
        f32x4 hacc[MTW];
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            uint2 xs[NS];
            E::split(v[i], xs);
            f32x4 h[E::NACC];
#pragma unroll
            for (int a = 0; a < E::NACC; a++) h[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < E::NP; p++) h[E::acc(p)] = E::mfma16(hw[E::ws(p)], xs[E::as(p)], h[E::acc(p)]);
#pragma unroll
            for (int rg = 0; rg < 4; rg++) hacc[i][rg] = E::fold(h[0][rg], h[E::NACC - 1][rg]);
        }
        __syncthreads();                       // every wave has finished reading the input image: the partial sums overlay it
        float *o = reinterpret_cast<float *>(out) + ng * NH * G::MR;
        // result rows = head channels: lanes q = 0 hold channels 0..3, q = 1 channels 4..7 (rows NH..15 are padding)
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            const int mt = mg + i * MG;
            if (mt < G::MT && q < 2) {
                const int m = mt * 16 + r16;
#pragma unroll
                for (int rg = 0; rg < 4; rg++)
                    if (q * 4 + rg < NH) o[(q * 4 + rg) * G::MR + m] = hacc[i][rg];
            }
        }
    } else {
        __syncthreads();                       // every wave has finished reading the input image: the output overlays it
#pragma unroll
        for (int i = 0; i < MTW; i++) {
            const int mt = mg + i * MG;
            if (mt < G::MT) {
                const int m = mt * 16 + r16;
                if (cellof[m] != 0xFFFFu) store_parts<G, SCH>(reinterpret_cast<uint2 *>(out), ng * 16 + q * 4, wpos[m], v[i]);
            }
        }
    }
}

template <int N, int SCH>
__global__ __launch_bounds__(AZ_NW * 64) void k_trunk_emul(DevState d, NetWeights w, int net_id, float *__restrict__ feat,
                                                           unsigned long long *dbg)
{
    typedef NetGeo<N> G;
    typedef Emul<SCH> E;
    constexpr int NTH = AZ_NW * 64, NS = E::NS;
    constexpr int XF = 32 * NS * G::CS;          // floats of the 64-channel image: 8 channel groups x NS planes of CS 16-byte slots
    static_assert(XF <= G::LDSF, "the split image does not fit the trunk's LDS");
#ifndef AZ_EXPERIMENT     // experiment builds of the float32 trunk with another wave count never run this kernel
    static_assert(AZ_NW == 8, "the emulated trunk is laid out for 8 waves: 2 / 4 / 8 channel tiles in conv1 / conv2 / conv3");
    static_assert(AZ_NW * 6 * G::MR <= G::LDSF, "head-conv partial sums do not fit the trunk's LDS");
#endif
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
    // ---- requests that do not depend on anything: the wave's conv1 fragments, biases, head fragments, conv2's first K-block
    EmulPre<SCH> pre;
    uint4 w2[NS], w3[NS];
    const uint4 *c2x = reinterpret_cast<const uint4 *>(w.c2x[SCH - 1]), *c3x = reinterpret_cast<const uint4 *>(w.c3x[SCH - 1]);
    {
        const int q = lane >> 4;
        emul_weights<2, NS>(reinterpret_cast<const uint4 *>(w.c1x[SCH - 1]), wave % 2, 0, lane, pre.c1w[0]);
        emul_weights<2, NS>(reinterpret_cast<const uint4 *>(w.c1x[SCH - 1]), wave % 2, 1, lane, pre.c1w[1]);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            pre.b1[rg] = w.c1b[(wave % 2) * 16 + q * 4 + rg];
            pre.b2[rg] = w.c2b[(wave % 4) * 16 + q * 4 + rg];
            pre.b3[rg] = w.c3b[wave * 16 + q * 4 + rg];
        }
        const uint2 *hx = reinterpret_cast<const uint2 *>(w.hdx[SCH - 1]) + (size_t)wave * NS * 64 + lane;
#pragma unroll
        for (int s = 0; s < NS; s++) pre.hw[s] = hx[s * 64];
        emul_weights<9, NS>(c2x, wave % 4, 0, lane, w2);
    }
    // region X = 8 NS planes of CS slots: conv1 out in the first 4 NS planes, conv2 out in all of them; conv1's input image
    // ([position][4 planes] of 16-bit values, 8 bytes per position) is the first half of plane 4 NS
    uint2 *inP = reinterpret_cast<uint2 *>(lds + 16 * NS * G::CS);
    // Every thread requests the leaf words of its cells (games.py:86-129 encode) before it is known whether the group has
    // anything to evaluate: the loads are in flight while the image is zeroed.
    constexpr int EPT = (G::MR + NTH - 1) / NTH;
    int e_pos[EPT];
    bool e_me[EPT], e_op[EPT], e_last[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int m = tid + e * NTH;
        int pos = G::PW + 1, cell = 0xFFFF;
        if (m < G::MR) {
            if constexpr (G::ROWT) {
                const int t = m >> 4, c = m & 15, g = t / N, r = t - g * N;
                pos = g * G::PP + (r + 1) * G::PW + (c + 1);             // c == N is the right padding cell of the row
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
        for (int i = tid; i < XF / 4; i += NTH) z[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (!any_active) return;
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if (e_me[e] || e_op[e] || e_last[e])
            inP[e_pos[e]] = uint2{(e_me[e] ? E::ONE : 0u) | (e_op[e] ? E::ONE << 16 : 0u), e_last[e] ? E::ONE : 0u};
    __syncthreads();
    AZ_STAMP(1);
    conv_first_emul<G, 32, SCH>(inP, reinterpret_cast<uint2 *>(lds), pre.c1w, pre.b1, wpos, cellof, wave, lane);
    emul_weights<18, NS>(c3x, wave, 0, lane, w3);       // conv3's first K-block, a layer ahead
    __syncthreads();
    for (int i = tid; i < G::CS; i += NTH) inP[i] = uint2{0u, 0u};      // the input image's bytes are part of conv2's output image
    AZ_STAMP(2);
    conv_layer_emul<G, 32, 64, EMUL_OUT_IMAGE, 6, SCH>(reinterpret_cast<const uint4 *>(lds), lds, c2x, w2, pre.b2, pre.hw, wpos, cellof, wave, lane);
    __syncthreads();
    AZ_STAMP(3);
    // conv3 with the 1x1 head convs fused into its epilogue: partial sums [wave][head channel][cell] in LDS, added up here
    conv_layer_emul<G, 64, 128, EMUL_OUT_HEADS, 6, SCH>(reinterpret_cast<const uint4 *>(lds), lds, c3x, w3, pre.b3, pre.hw, wpos, cellof, wave, lane);
    __syncthreads();
    AZ_STAMP(4);
    for (int o = tid; o < 6 * G::MR; o += NTH) {
        const int j = o / G::MR, m = o - j * G::MR;
        const int cell = cellof[m];
        if (cell != 0xFFFF) {
            const int g = cell / G::nn, p = cell - g * G::nn;
            const int b = b0 + g;
            if (b < d.B && d.s_net[b] == net_id) {
                float v = lds[o];
#pragma unroll
                for (int wv = 1; wv < AZ_NW; wv++) v = v + lds[wv * 6 * G::MR + o];
                v = v + w.hdb[j];
                feat[(size_t)b * G::FROW + j * G::nn + p] = v > 0.0f ? v : 0.0f;      // net.py:64,69 flatten order
            }
        }
    }
    AZ_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// The ResidualBlock variant (BASELINE config 5; k_trunk_res in az_net.h) with the same emulation: stem on the f32 MFMA, the
// six 64 -> 64 convs and the 1x1 heads on the bf16 MFMA.  ONE 64-channel split image (24 planes) serves every layer: a conv
// keeps its result in accumulators until the barrier and writes it over its own input, and the skip connection's operand
// x never needs LDS -- every conv gives a wave the same channel tile x cell tiles, so the wave keeps its tiles of x as
// float32 registers (`keep`) from the epilogue that produced them to the epilogue that adds them (relu(conv2(h) + b + x)).
// First K-block fragments and biases of layer l + 1 are requested before layer l runs.
// ------------------------------------------------------------------------------------------------
#ifndef AZ_RES_EMUL_NW
#define AZ_RES_EMUL_NW 8       // waves per workgroup of the emulated ResidualBlock trunk (0 = as k_trunk_res: 12 at n = 15).  Measured at n = 15:
                              // 8 waves (4 channel tiles x 2 cell groups of 8 / 7 tiles, 256 VGPRs) 107.3 us per 256 boards, 12 waves (x 3 groups
                              // of 5, no surplus tile, 168 VGPRs and a 5-dword spill) 109.6 us
#endif
template <int N>
struct ResGeoEmul : ResGeo<N> {
    static constexpr int NW = AZ_RES_EMUL_NW ? AZ_RES_EMUL_NW : ResGeo<N>::NW;
    static constexpr bool EMUL_PRIO_TURNS = false;    // measured: the turns cost this kernel 4 % (f16x2 is at the register limit: they spill)
};

template <int N, int SCH>
__global__ __launch_bounds__(ResGeoEmul<N>::NW * 64) void k_trunk_res_emul(DevState d, ResWeights w, int net_id, float *__restrict__ feat)
{
    typedef ResGeoEmul<N> G;
    typedef Emul<SCH> E;
    constexpr int NTH = G::NW * 64, NS = E::NS;
    constexpr int NG = 4, MG = G::NW / NG, MTW = (G::MT + MG - 1) / MG, NH = G::PC + G::VC;
    constexpr int XF = 32 * NS * G::CS;          // floats of the 64-channel image
    static_assert(XF + 2 * G::CS <= G::LDSF && NG * NH * G::MR <= XF, "LDS layout of the emulated ResidualBlock trunk");
    __shared__ __attribute__((aligned(16))) float lds[G::LDSF];
    __shared__ unsigned short wpos[G::MR];
    __shared__ unsigned short cellof[G::MR];
    __shared__ int any_active;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * G::G;
    const int ng = wave % NG, q = lane >> 4;
    if (tid == 0) any_active = 0;
    __syncthreads();
    if (tid < G::G) {
        const int b = b0 + tid;
        if (b < d.B) {
            const int kind = d.leaf_kind[b];
            if (leaf_needs_net(kind) && d.s_status[b] == SLOT_ACTIVE && d.s_net[b] == net_id) atomicOr(&any_active, 1);
        }
    }
    // requests that depend on nothing: stem fragments and bias, the first conv's first K-block and bias
    uint4 c1w[2][NS];
    float bcur[4], bnxt[4];
    uint2 hw[NS];
    uint4 wcur[NS], wnxt[NS];
    const void *const *blkx = w.blkx[SCH - 1];
    {
        emul_weights<2, NS>(reinterpret_cast<const uint4 *>(w.stemx[SCH - 1]), ng, 0, lane, c1w[0]);
        emul_weights<2, NS>(reinterpret_cast<const uint4 *>(w.stemx[SCH - 1]), ng, 1, lane, c1w[1]);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bcur[rg] = w.stemb[ng * 16 + q * 4 + rg];
#pragma unroll
        for (int s = 0; s < NS; s++) hw[s] = uint2{0u, 0u};
        emul_weights<18, NS>(reinterpret_cast<const uint4 *>(blkx[0]), ng, 0, lane, wnxt);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bnxt[rg] = w.blkb[0][ng * 16 + q * 4 + rg];
    }
    // the 64-channel split image = 8 NS planes of CS slots; the stem's input image ([position][4 planes], 16-bit) behind it
    uint2 *inP = reinterpret_cast<uint2 *>(lds + XF);
    constexpr int EPT = (G::MR + NTH - 1) / NTH;
    int e_pos[EPT];
    bool e_me[EPT], e_op[EPT], e_last[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int m = tid + e * NTH;
        int pos = G::PW + 1, cell = 0xFFFF;
        if (m < G::MR) {
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
        e_pos[e] = pos;
        e_me[e] = e_op[e] = e_last[e] = false;
        if (cell != 0xFFFF) {
            const int g = cell / G::nn, p = cell - g * G::nn;
            const int b = b0 + g;
            if (b < d.B) {
                const u64 *lf = d.leaf + (size_t)b * 8;
                const int ps = sym_cell(d.leaf_sym, b, p, N);
                e_me[e] = (lf[ps >> 6] >> (ps & 63)) & 1ull;          // games.py:86-129 encode
                e_op[e] = (lf[4 + (ps >> 6)] >> (ps & 63)) & 1ull;
                e_last[e] = d.leaf_last[b] == ps;
            }
        }
    }
    {
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < (XF + 2 * G::CS) / 4; i += NTH) z[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (!any_active) return;
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if (e_me[e] || e_op[e] || e_last[e])
            inP[e_pos[e]] = uint2{(e_me[e] ? E::ONE : 0u) | (e_op[e] ? E::ONE << 16 : 0u), e_last[e] ? E::ONE : 0u};
    __syncthreads();
    f32x4 keep[MTW];
    const uint4 *X = reinterpret_cast<const uint4 *>(lds);
    conv_first_emul<G, 64, SCH>(inP, reinterpret_cast<uint2 *>(lds), c1w, bcur, wpos, cellof, wave, lane, keep);
    __syncthreads();
    // layer l's first K-block and bias were requested a layer ago (wnxt / bnxt); take them over and request layer l + 1's
    auto advance = [&](int l) {
#pragma unroll
        for (int s = 0; s < NS; s++) wcur[s] = wnxt[s];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bcur[rg] = bnxt[rg];
        if (l + 1 < 6) {
            emul_weights<18, NS>(reinterpret_cast<const uint4 *>(blkx[l + 1]), ng, 0, lane, wnxt);
#pragma unroll
            for (int rg = 0; rg < 4; rg++) bnxt[rg] = w.blkb[l + 1][ng * 16 + q * 4 + rg];
        }
    };
#pragma unroll 1
    for (int blk = 0; blk < 3; blk++) {
        advance(2 * blk);           // block conv1: h = relu(bn1(conv1(x)))
        conv_layer_emul<G, 64, 64, EMUL_OUT_IMAGE, NH, SCH>(X, lds, reinterpret_cast<const uint4 *>(blkx[2 * blk]), wcur, bcur, hw, wpos, cellof,
                                                           wave, lane, keep);
        __syncthreads();
        advance(2 * blk + 1);
        if (blk < 2) {              // block conv2: relu(bn2(conv2(h)) + x); the new x stays in registers as well
            conv_layer_emul<G, 64, 64, EMUL_OUT_IMAGE | EMUL_SKIP | EMUL_KEEP, NH, SCH>(X, lds, reinterpret_cast<const uint4 *>(blkx[2 * blk + 1]),
                                                                                     wcur, bcur, hw, wpos, cellof, wave, lane, keep);
        } else {                    // res3.conv2 + skip, then the 1x1 heads straight from the accumulators
            const uint2 *hx = reinterpret_cast<const uint2 *>(w.hdx[SCH - 1]) + (size_t)ng * NS * 64 + lane;
#pragma unroll
            for (int s = 0; s < NS; s++) hw[s] = hx[s * 64];
            conv_layer_emul<G, 64, 64, EMUL_OUT_HEADS | EMUL_SKIP, NH, SCH>(X, lds, reinterpret_cast<const uint4 *>(blkx[5]), wcur, bcur, hw, wpos,
                                                                          cellof, wave, lane, keep);
        }
        __syncthreads();
    }
    for (int o = tid; o < NH * G::MR; o += NTH) {
        const int j = o / G::MR, m = o - j * G::MR;
        const int cell = cellof[m];
        if (cell != 0xFFFF) {
            const int g = cell / G::nn, p = cell - g * G::nn;
            const int b = b0 + g;
            if (b < d.B && d.s_net[b] == net_id) {
                float v = lds[o];
#pragma unroll
                for (int t = 1; t < NG; t++) v = v + lds[t * NH * G::MR + o];
                v = v + w.hdb[j];
                feat[(size_t)b * G::FROW + j * G::nn + p] = v > 0.0f ? v : 0.0f;      // 0-1 policy_conv, 2 value_conv
            }
        }
    }
}
