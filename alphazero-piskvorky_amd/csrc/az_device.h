// az_device.h -- device-side helpers for the gfx950 self-play engine.
//
// Everything here is written for CDNA4 only (64-wide wavefronts, no portability shims).
// Floating-point helpers are the engine's "canonical" forms: fixed sequences of IEEE
// operations (fma chains, correctly rounded divides) so that results do not depend on a
// vendor libm.  Build with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AZ_WAVE 64

// ---------------------------------------------------------------------------------------
// canonical math (same operation sequence as the reference restatement used for testing)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float az_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float t = x * 1.44269504088896341f;
    float kf = __builtin_rintf(t);
    float r = __builtin_fmaf(kf, -0.693145751953125f, x);
    r = __builtin_fmaf(kf, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 5040.0f;
    p = __builtin_fmaf(p, r, 1.0f / 720.0f);
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    int k = (int)kf;
    return p * __uint_as_float((unsigned)(k + 127) << 23);
}

__device__ __forceinline__ float az_tanhf(float x)
{
    float a = __builtin_fabsf(x);
    float t = az_expf(-2.0f * a);
    float r = (1.0f - t) / (1.0f + t);
    return x < 0.0f ? -r : r;
}

__device__ __forceinline__ double az_exp(double x)
{
    if (x < -708.0) return 0.0;
    if (x > 709.0) x = 709.0;
    double kf = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(kf, -6.93147180369123816490e-01, x);
    r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    long long k = (long long)kf;
    return p * __longlong_as_double((long long)((u64)(k + 1023) << 52));
}

__device__ __forceinline__ unsigned az_fmix32(unsigned x)
{
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

// ---------------------------------------------------------------------------------------
// wave-level reductions (64 lanes).  Butterfly order is part of the canonical definition.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_butterfly(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}
// Cross-lane moves on the DPP path (no LDS round trip, unlike __shfl's ds_bpermute).  Only patterns whose meaning leaves no
// room for a direction convention: the two quad permutes that are xor 1 / xor 2, and the two mirrors.  After the four steps
// below every lane of a 16-lane row holds the row's result of an idempotent, commutative reduction (max, min); the four rows
// are combined through v_readlane.  max and min do not depend on the order of their operands, so these give the same bits
// as the butterfly they replace (measured on the tree step of the 5x5 search kernel: one selection level 3.0 k -> see DESIGN).
#define AZ_DPP_XOR1 0xB1         // quad_perm [1, 0, 3, 2]
#define AZ_DPP_XOR2 0x4E         // quad_perm [2, 3, 0, 1]
#define AZ_DPP_HALF_MIRROR 0x141 // lane i <- lane 7 - i within every 8 lanes
#define AZ_DPP_MIRROR 0x140      // lane i <- lane 15 - i within every 16 lanes
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v)
{
    return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_d(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

__device__ __forceinline__ float wave_max_f(float v)
{
    v = fmaxf(v, dpp_f<AZ_DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<AZ_DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<AZ_DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<AZ_DPP_MIRROR>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        double o = __shfl_xor(v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_xor_u(unsigned v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v ^ (unsigned)__shfl_xor((int)v, m, 64);
    return v;
}
__device__ __forceinline__ double max_d(double a, double b) { return b > a ? b : a; }
__device__ __forceinline__ int min_i(int a, int b) { return b < a ? b : a; }
// argmax with first-index tie-break: max score, then min index (mcts.py:71 "first maximal child"); idx < 0 = no candidate in
// this lane.  Two order-independent reductions: the maximum of the candidates' scores, then the minimum index among the
// lanes that hold it.
__device__ __forceinline__ void wave_argmax(double &s, int &idx)
{
    double m = idx >= 0 ? s : -INFINITY;
    m = max_d(m, dpp_d<AZ_DPP_XOR1>(m));
    m = max_d(m, dpp_d<AZ_DPP_XOR2>(m));
    m = max_d(m, dpp_d<AZ_DPP_HALF_MIRROR>(m));
    m = max_d(m, dpp_d<AZ_DPP_MIRROR>(m));
    m = max_d(max_d(lane_d(m, 0), lane_d(m, 16)), max_d(lane_d(m, 32), lane_d(m, 48)));
    int c = (idx >= 0 && s == m) ? idx : 0x7FFFFFFF;
    c = min_i(c, dpp_i<AZ_DPP_XOR1>(c));
    c = min_i(c, dpp_i<AZ_DPP_XOR2>(c));
    c = min_i(c, dpp_i<AZ_DPP_HALF_MIRROR>(c));
    c = min_i(c, dpp_i<AZ_DPP_MIRROR>(c));
    c = min_i(min_i(__builtin_amdgcn_readlane(c, 0), __builtin_amdgcn_readlane(c, 16)),
              min_i(__builtin_amdgcn_readlane(c, 32), __builtin_amdgcn_readlane(c, 48)));
    idx = c == 0x7FFFFFFF ? -1 : c;
    s = m;
}

// value_fc2 (net.py:70): acc = fma(h[i], w2[i], acc) for i = 0 .. 63 in this order, the operands held one per lane.  Every lane
// computes the same chain; the operands come through v_readlane (until round 3: two ds_bpermutes and an LDS round trip per
// step, 5.9 k of the 14 k cycles of a tree step).
__device__ __forceinline__ float value_fc2_chain(float h_l, float w2_l)
{
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 64; i++) acc = __builtin_fmaf(lane_f(h_l, i), lane_f(w2_l, i), acc);
    return acc;
}

// the same chain with both operand vectors in LDS (16-byte aligned): every lane reads them with broadcast ds_read_b128
__device__ __forceinline__ float value_fc2_chain_lds(const float *h, const float *w2)
{
    const float4 *h4 = reinterpret_cast<const float4 *>(h), *w4 = reinterpret_cast<const float4 *>(w2);
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float4 a = h4[i], b = w4[i];
        acc = __builtin_fmaf(a.x, b.x, acc);
        acc = __builtin_fmaf(a.y, b.y, acc);
        acc = __builtin_fmaf(a.z, b.z, acc);
        acc = __builtin_fmaf(a.w, b.w, acc);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------
// bit-planed boards: 4 x u64 per plane (n*n <= 225 bits), cell j -> word j>>6, bit j&63
// ---------------------------------------------------------------------------------------
struct Plane {
    u64 w[4];
};
__device__ __forceinline__ bool pl_get(const Plane &p, int j)
{
    int wi = j >> 6;
    u64 v = wi == 0 ? p.w[0] : wi == 1 ? p.w[1] : wi == 2 ? p.w[2] : p.w[3];
    return (v >> (j & 63)) & 1ull;
}
__device__ __forceinline__ void pl_set(Plane &p, int j)
{
    int wi = j >> 6;
    u64 b = 1ull << (j & 63);
    p.w[0] |= wi == 0 ? b : 0ull;
    p.w[1] |= wi == 1 ? b : 0ull;
    p.w[2] |= wi == 2 ? b : 0ull;
    p.w[3] |= wi == 3 ? b : 0ull;
}
__device__ __forceinline__ int pl_count(const Plane &p)
{
    return __popcll(p.w[0]) + __popcll(p.w[1]) + __popcll(p.w[2]) + __popcll(p.w[3]);
}
// number of set bits strictly below position j
__device__ __forceinline__ int pl_rank(const Plane &p, int j)
{
    int wi = j >> 6;
    u64 mask = (1ull << (j & 63)) - 1ull;
    int c = 0;
    c += wi > 0 ? __popcll(p.w[0]) : (wi == 0 ? __popcll(p.w[0] & mask) : 0);
    c += wi > 1 ? __popcll(p.w[1]) : (wi == 1 ? __popcll(p.w[1] & mask) : 0);
    c += wi > 2 ? __popcll(p.w[2]) : (wi == 2 ? __popcll(p.w[2] & mask) : 0);
    c += wi == 3 ? __popcll(p.w[3] & mask) : 0;
    return c;
}
__device__ __forceinline__ Plane pl_load(const u64 *g)
{
    Plane p;
    p.w[0] = g[0]; p.w[1] = g[1]; p.w[2] = g[2]; p.w[3] = g[3];
    return p;
}
__device__ __forceinline__ void pl_store(u64 *g, const Plane &p)
{
    g[0] = p.w[0]; g[1] = p.w[1]; g[2] = p.w[2]; g[3] = p.w[3];
}

// games.py:133-166 restricted to lines through the stone just placed at cell a of plane `me`
// (any new run of >= k must contain it; overlines count, games.py:212-227).
__device__ __forceinline__ bool wins_through(const Plane &me, int a, int n, int k)
{
    int r = a / n, c = a - r * n;
    const int dr[4] = {0, 1, 1, 1}, dc[4] = {1, 0, 1, -1};
    bool win = false;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        int cnt = 1;
        for (int s = 1; s < k; s++) {
            int rr = r + s * dr[d], cc = c + s * dc[d];
            if (rr < 0 || rr >= n || cc < 0 || cc >= n || !pl_get(me, rr * n + cc)) break;
            cnt++;
        }
        for (int s = 1; s < k; s++) {
            int rr = r - s * dr[d], cc = c - s * dc[d];
            if (rr < 0 || rr >= n || cc < 0 || cc >= n || !pl_get(me, rr * n + cc)) break;
            cnt++;
        }
        win = win || (cnt >= k);
    }
    return win;
}

// ---------------------------------------------------------------------------------------
// dihedral symmetries of the board: k < 4 = np.rot90 k times (counter-clockwise), k >= 4 = rot90(fliplr(x), k - 4)
// (games.py:183-197 rot90 / flip).  sym_src: source cell of output cell (i, j), i.e. sym(x)[i][j] = x[sym_src(k, i, j)].
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int sym_src(int k, int i, int j, int n)
{
    int si, sj;
    switch (k & 3) {
    case 0: si = i; sj = j; break;
    case 1: si = j; sj = n - 1 - i; break;
    case 2: si = n - 1 - i; sj = n - 1 - j; break;
    default: si = n - 1 - j; sj = i; break;
    }
    if (k >= 4) sj = n - 1 - sj;
    return si * n + sj;
}
__device__ __forceinline__ int sym_inverse(int k) { return k < 4 ? ((4 - k) & 3) : k; }    // the reflections are involutions
// Opt-in random-symmetry leaf evaluation (az_set_leaf_symmetry): which of the 8 symmetries evaluation `idx` (0 = the root,
// s + 1 = simulation s) of the search of `game` at `ply` shows to the net.  Same hash in the oracle (orc_leaf_sym).
__device__ __forceinline__ int leaf_sym_of(int game, int ply, int idx)
{
    return (int)(az_fmix32((unsigned)game * 0x9E3779B1u ^ az_fmix32((unsigned)ply * 0x85EBCA6Bu + (unsigned)idx + 1u)) & 7u);
}
// board cell whose stone the net sees at image cell p (p itself without the option)
__device__ __forceinline__ int sym_cell(const int *leaf_sym, int item, int p, int n)
{
    if (!leaf_sym) return p;
    const int i = p / n;
    return sym_src(leaf_sym[item], i, p - i * n, n);
}

// The same test done by the 64 lanes of a wavefront together (every lane must be active and hold the same plane and a):
// lane = 16 * direction + 8 * side + (s - 1) probes the cell s steps from a along that direction and side; one ballot
// gathers the 64 probes, and the run through a is counted with count-trailing-zero on the inverted 8-bit groups.  Replaces
// up to 8 (k - 1) dependent scalar probes on the selection's latency chain.  Falls back to the loop for k > 9.
__device__ __forceinline__ bool wins_through_wave(const Plane &me, int a, int n, int k, int lane)
{
    if (k > 9) return wins_through(me, a, n, k);
    const int r = a / n, c = a - r * n;
    const int dsel = lane >> 4, s = (lane & 7) + 1, sgn = (lane & 8) ? -s : s;
    const int dr = dsel == 0 ? 0 : 1, dc = dsel == 0 ? 1 : (dsel == 1 ? 0 : (dsel == 2 ? 1 : -1));
    const int rr = r + sgn * dr, cc = c + sgn * dc;
    const bool inb = s < k && rr >= 0 && rr < n && cc >= 0 && cc < n;
    const bool stone = inb && pl_get(me, inb ? rr * n + cc : 0);
    const u64 m = __ballot(stone);
    bool win = false;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const unsigned g = (unsigned)(m >> (16 * d)) & 0xFFFFu;
        const int fwd = __builtin_ctz(~(g & 0xFFu) | 0x100u);          // consecutive stones on the + side (0..8)
        const int bwd = __builtin_ctz(~((g >> 8) & 0xFFu) | 0x100u);   // ... and on the - side
        win = win || (1 + fwd + bwd >= k);
    }
    return win;
}
