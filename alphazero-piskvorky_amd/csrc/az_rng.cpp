// az_rng.cpp -- host-side, numpy-compatible legacy RandomState streams.
//
// The reference draws its randomness from numpy's global legacy MT19937 generator:
//   mcts.py:114  np.random.dirichlet([alpha] * len(legal))     (legacy gamma rejection sampler)
//   mcts.py:177  np.random.choice(len(actions), p=probs)        (exactly one random_sample())
// The draw sequence per game does not depend on the trajectory, so the host can
// produce each game's "tape" ahead of the device search.  This file restates the published
// algorithms of numpy's legacy generator (numpy/random/src/mt19937, src/legacy/legacy-distributions.c)
// so that RandomState(seed) streams are reproduced bit for bit (tests/test_host_rng.py checks
// against numpy itself).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/az_engine.h"

namespace azrng {

struct MT {
    uint32_t key[624];
    int pos;
    bool has_gauss = false;
    double gauss = 0.0;

    explicit MT(uint32_t seed)
    {
        for (int i = 0; i < 624; i++) {
            key[i] = seed;
            seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    void gen()
    {
        const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAT = 0x9908b0dfu;
        int i;
        uint32_t y;
        for (i = 0; i < 624 - 397; i++) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
        }
        for (; i < 623; i++) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
        }
        y = (key[623] & UPPER) | (key[0] & LOWER);
        key[623] = key[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MAT);
        pos = 0;
    }
    uint32_t next32()
    {
        if (pos == 624) gen();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double next_double()
    {
        int32_t a = (int32_t)(next32() >> 5), b = (int32_t)(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    double std_exponential() { return -std::log(1.0 - next_double()); }
    double legacy_gauss()
    {
        if (has_gauss) {
            double t = gauss;
            has_gauss = false;
            gauss = 0.0;
            return t;
        }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * next_double() - 1.0;
            x2 = 2.0 * next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = std::sqrt(-2.0 * std::log(r2) / r2);
        gauss = f * x1;
        has_gauss = true;
        return f * x2;
    }
    double std_gamma(double shape)
    {
        if (shape == 1.0) return std_exponential();
        if (shape == 0.0) return 0.0;
        if (shape < 1.0) {
            for (;;) {
                double U = next_double();
                double V = std_exponential();
                if (U <= 1.0 - shape) {
                    double X = std::pow(U, 1. / shape);
                    if (X <= V) return X;
                } else {
                    double Y = -std::log((1 - U) / shape);
                    double X = std::pow(1.0 - shape + shape * Y, 1. / shape);
                    if (X <= (V + Y)) return X;
                }
            }
        }
        double b = shape - 1. / 3.;
        double c = 1. / std::sqrt(9 * b);
        for (;;) {
            double X, V;
            do {
                X = legacy_gauss();
                V = 1.0 + c * X;
            } while (V <= 0.0);
            V = V * V * V;
            double U = next_double();
            if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return b * V;
            if (std::log(U) < 0.5 * X * X + b * (1. - V + std::log(V))) return b * V;
        }
    }
    // RandomState.dirichlet([alpha]*k): gammas, then scale by 1/sum
    void dirichlet(double alpha, int k, double *out)
    {
        double acc = 0.0;
        for (int j = 0; j < k; j++) {
            out[j] = std_gamma(alpha);
            acc = acc + out[j];
        }
        double inv = 1 / acc;
        for (int j = 0; j < k; j++) out[j] = out[j] * inv;
    }
};

// One game's self-play tape: per ply m, dirichlet over nn-m cells then one uniform.
void selfplay_tape(uint64_t seed, int nn, double alpha, int max_plies, double *noise, double *u)
{
    MT mt((uint32_t)(seed & 0xffffffffu));
    size_t off = 0;
    int plies = max_plies > 0 && max_plies < nn ? max_plies : nn;
    for (int m = 0; m < plies; m++) {
        mt.dirichlet(alpha, nn - m, noise + off);
        off += (size_t)(nn - m);
        u[m] = mt.next_double();
    }
}

// Resumable per-game streams: the same tapes, produced a few plies at a time while the device plays
// (the engine's tape producer).  Game i is RandomState(seed0 + i); plies must be requested in order.
struct Streams {
    std::vector<MT> mt;
};

Streams *streams_new(uint64_t seed0, int count)
{
    Streams *s = new Streams();
    s->mt.reserve((size_t)count);
    for (int i = 0; i < count; i++) s->mt.emplace_back((uint32_t)((seed0 + (uint64_t)i) & 0xffffffffu));
    return s;
}

void streams_free(Streams *s) { delete s; }

// plies [m0, m1) of game i: noise_row receives sum_{m} (nn - m) doubles back to back, u_row one double per ply
void streams_plies(Streams *s, int i, int nn, double alpha, int m0, int m1, double *noise_row, double *u_row)
{
    MT &mt = s->mt[(size_t)i];
    size_t off = 0;
    for (int m = m0; m < m1; m++) {
        mt.dirichlet(alpha, nn - m, noise_row + off);
        off += (size_t)(nn - m);
        u_row[m - m0] = mt.next_double();
    }
}

void uniforms(uint64_t seed, int count, double *u)
{
    MT mt((uint32_t)(seed & 0xffffffffu));
    for (int i = 0; i < count; i++) u[i] = mt.next_double();
}

// Tapes for games [g0, g0+count) written with a fixed stride, generated by a small thread pool.
void selfplay_tapes_parallel(uint64_t seed0, int g0, int count, int nn, double alpha, int max_plies, double *noise,
                             int64_t noise_stride, double *u, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > count) threads = count > 0 ? count : 1;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([=]() {
            for (int i = t; i < count; i += threads)
                selfplay_tape(seed0 + (uint64_t)(g0 + i), nn, alpha, max_plies, noise + (int64_t)i * noise_stride,
                              u + (int64_t)i * nn);
        });
    }
    for (auto &th : pool) th.join();
}

} // namespace azrng

extern "C" int az_rng_selfplay_tape(uint64_t seed, int board_size, double alpha, int max_plies, double *noise, double *u)
{
    if (board_size < 1 || board_size > 15 || !noise || !u || alpha <= 0.0) return AZ_ERR_INVALID;
    azrng::selfplay_tape(seed, board_size * board_size, alpha, max_plies, noise, u);
    return AZ_OK;
}

extern "C" int az_rng_uniforms(uint64_t seed, int count, double *u)
{
    if (count < 0 || !u) return AZ_ERR_INVALID;
    azrng::uniforms(seed, count, u);
    return AZ_OK;
}
