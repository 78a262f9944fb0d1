/*
 * az_oracle.c -- CPU ORACLE for the AlphaZero-Piskvorky self-play hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (alphazero-piskvorky_amd/)
 * never links, imports or calls anything in oracle/.
 *
 * It restates, in plain C, the algorithm of the Python reference (paths relative to the
 * reference checkout):
 *   alphazero/games.py      Gomoku rules, encode                      (games.py:21-227)
 *   alphazero/mcts.py       Node / MCTS.run                           (mcts.py:25-183)
 *   alphazero/net.py        GomokuNet.forward                         (net.py:55-72)
 *   alphazero/controller.py make_policy_value_fn                      (controller.py:33-55)
 *   alphazero/self_play.py  _worker loop body, z labels, augmentation (self_play.py:24-108)
 *   alphazero/evaluator.py  ModelEvaluator.evaluate game loop         (evaluator.py:50-104)
 *
 * Pinning: checked in tests/test_oracle_*.py against golden vectors captured by importing
 * the reference in the build container (tests/golden/make_golden.py):
 *   - rules / legal masks / outcomes / encode planes: bit-exact
 *   - tree statistics (visit counts, W, pi, action) with the synthetic evaluator: bit-exact
 *   - net outputs vs torch CPU: |dlogit| <= 2e-5, |dP| <= 1e-6, |dv| <= 1e-6 (summation order differs)
 *   - full real-net games: teacher-forced per ply, same tolerances
 *
 * Floating-point conventions ("canonical order", shared by definition with the HIP engine so
 * that engine == oracle bit for bit):
 *   - every dot product is ONE k-ordered chain of fmaf starting from +0, bias added afterwards
 *     (this is what v_mfma_f32_* computes); conv k = (ky*3+kx)*Cin + ci
 *   - expf / exp / tanhf are the fixed polynomial forms below (libm is not bit-reproducible on GPU)
 *   - softmax denominator = 64-lane butterfly sum of per-lane partials (cell j -> lane j%64)
 *   - tree arithmetic in IEEE double exactly in the reference's operator order (mcts.py:73,78-80)
 * Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXN 15
#define ORC_MAXNN (ORC_MAXN * ORC_MAXN)

enum { RES_NONE = 0, RES_X = 1, RES_O = 2, RES_DRAW = 3 };

/* ------------------------------------------------------------------ deterministic math */
static float orc_expf(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float t = x * 1.44269504088896341f;
    float kf = rintf(t);
    float r = fmaf(kf, -0.693145751953125f, x);
    r = fmaf(kf, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int k = (int)kf;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)(k + 127) << 23;
    return p * s.f;
}

static float orc_tanhf(float x)
{
    float a = fabsf(x);
    float t = orc_expf(-2.0f * a);
    float r = (1.0f - t) / (1.0f + t);
    return x < 0.0f ? -r : r;
}

static double orc_exp(double x)
{
    if (x < -708.0) return 0.0;
    if (x > 709.0) x = 709.0;
    double kf = rint(x * 1.4426950408889634074);
    double r = fma(kf, -6.93147180369123816490e-01, x);
    r = fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;          /* 1/13! */
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int64_t k = (int64_t)kf;
    union { uint64_t u; double f; } s;
    s.u = (uint64_t)(k + 1023) << 52;
    return p * s.f;
}

static uint32_t fmix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

/* 64-lane butterfly sum: lane l ends with the same value in every lane (a+b == b+a). */
static float butterfly64f(float *v)
{
    for (int m = 32; m >= 1; m >>= 1) {
        float t[64];
        for (int l = 0; l < 64; l++) t[l] = v[l] + v[l ^ m];
        memcpy(v, t, sizeof t);
    }
    return v[0];
}

/* ------------------------------------------------------------------ rules (games.py) */
typedef struct {
    int n, k;
    uint8_t cell[ORC_MAXNN]; /* 0 empty, 1 X, 2 O      (games.py:27 board of None/'X'/'O') */
    int player;              /* 1 X, 2 O               (games.py:29) */
    int last;                /* r*n+c or -1            (games.py:32) */
    int winner;              /* RES_* cached           (games.py:31,140-141,206) */
} orc_state;

static void st_init(orc_state *s, int n, int k)
{
    memset(s, 0, sizeof *s);
    s->n = n; s->k = k; s->player = 1; s->last = -1; s->winner = RES_NONE;
}

/* games.py:212-227 _check_line: exactly win_length consecutive stones starting at (r,c) */
static int check_line(const orc_state *s, int r, int c, int dr, int dc, int pl)
{
    int cnt = 0;
    for (int i = 0; i < s->k; i++) {
        int nr = r + i * dr, nc = c + i * dc;
        if (nr >= 0 && nr < s->n && nc >= 0 && nc < s->n && s->cell[nr * s->n + nc] == pl) cnt++;
        else break;
    }
    return cnt == s->k;
}

/* games.py:133-166 is_terminal (mutates the cached winner, like the reference) */
static int st_terminal(orc_state *s)
{
    if (s->winner != RES_NONE) return 1;
    int n = s->n, full = 1;
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) {
            int pl = s->cell[r * n + c];
            if (!pl) { full = 0; continue; }
            if (check_line(s, r, c, 0, 1, pl) || check_line(s, r, c, 1, 0, pl) ||
                check_line(s, r, c, 1, 1, pl) || check_line(s, r, c, 1, -1, pl)) {
                s->winner = pl;
                return 1;
            }
        }
    if (full) { s->winner = RES_DRAW; return 1; }
    return 0;
}

/* games.py:64-82 apply_action; returns -1 for the reference's ValueError("Invalid move") */
static int st_apply(orc_state *s, int a)
{
    if (s->cell[a]) return -1;
    s->cell[a] = (uint8_t)s->player;
    s->player = 3 - s->player;
    s->last = a;
    return 0;
}

/* games.py:86-129 encode: ch0 mover, ch1 opponent, ch2 last move, ch3 zeros */
static void st_encode(const orc_state *s, float *planes)
{
    int nn = s->n * s->n;
    memset(planes, 0, sizeof(float) * 4 * nn);
    for (int i = 0; i < nn; i++) {
        if (s->cell[i] == s->player) planes[i] = 1.0f;
        else if (s->cell[i]) planes[nn + i] = 1.0f;
    }
    if (s->last >= 0) planes[2 * nn + s->last] = 1.0f;
}

/* ---- exported rules entry points ---- */
int orc_replay(int n, int k, const int16_t *actions, int nact, uint8_t *term_before, uint8_t *board_out,
               int *player_out, int *result_out)
{
    orc_state s; st_init(&s, n, k);
    for (int i = 0; i < nact; i++) {
        if (term_before) term_before[i] = (uint8_t)st_terminal(&s);
        if (st_apply(&s, actions[i]) != 0) return -1;
    }
    int t = st_terminal(&s);
    if (board_out) memcpy(board_out, s.cell, (size_t)n * n);
    if (player_out) *player_out = s.player;
    if (result_out) *result_out = t ? s.winner : RES_NONE;
    return 0;
}

void orc_encode(int n, const uint8_t *board, int player, int last, float *planes)
{
    orc_state s; st_init(&s, n, 5);
    memcpy(s.cell, board, (size_t)n * n); s.player = player; s.last = last;
    st_encode(&s, planes);
}

int orc_result(int n, int k, const uint8_t *board)
{
    orc_state s; st_init(&s, n, k);
    memcpy(s.cell, board, (size_t)n * n);
    return st_terminal(&s) ? s.winner : RES_NONE;
}

/* ------------------------------------------------------------------ net (net.py / controller.py) */
typedef struct {
    int n;
    int kind;                       /* 0 = GomokuNet (net.py), 1 = ResidualBlock variant (forward defined by the build) */
    float *rw[7], *rb[7];           /* kind 1: stem + six 64->64 convs, BN folded by the caller, repacked [tap][ci][co] */
    /* repacked for the canonical k order: conv w[tap][ci][co] */
    float *c1w, *c1b, *c2w, *c2b, *c3w, *c3b;
    float *pcw, *pcb, *pfw, *pfb;   /* policy_conv [4][128], policy_fc [nn][4nn] */
    float *vcw, *vcb, *v1w, *v1b, *v2w, *v2b;
} orc_net;

static float *dupf(const float *p, size_t cnt)
{
    float *q = (float *)malloc(cnt * sizeof(float));
    memcpy(q, p, cnt * sizeof(float));
    return q;
}

static float *repack_conv(const float *w, int cout, int cin)
{
    /* torch [co][ci][ky][kx] -> [tap][ci][co] */
    float *q = (float *)malloc((size_t)cout * cin * 9 * sizeof(float));
    for (int co = 0; co < cout; co++)
        for (int ci = 0; ci < cin; ci++)
            for (int t = 0; t < 9; t++)
                q[((size_t)t * cin + ci) * cout + co] = w[((size_t)co * cin + ci) * 9 + t];
    return q;
}

/* tensors in state_dict order (net.py:37-53): conv1.w,b conv2.w,b conv3.w,b policy_conv.w,b
 * policy_fc.w,b value_conv.w,b value_fc1.w,b value_fc2.w,b */
orc_net *orc_net_create(int n, const float *const *t)
{
    int nn = n * n;
    orc_net *N = (orc_net *)calloc(1, sizeof *N);
    N->n = n;
    N->c1w = repack_conv(t[0], 32, 4);   N->c1b = dupf(t[1], 32);
    N->c2w = repack_conv(t[2], 64, 32);  N->c2b = dupf(t[3], 64);
    N->c3w = repack_conv(t[4], 128, 64); N->c3b = dupf(t[5], 128);
    N->pcw = dupf(t[6], 4 * 128);  N->pcb = dupf(t[7], 4);
    N->pfw = dupf(t[8], (size_t)nn * 4 * nn); N->pfb = dupf(t[9], nn);
    N->vcw = dupf(t[10], 2 * 128); N->vcb = dupf(t[11], 2);
    N->v1w = dupf(t[12], (size_t)64 * 2 * nn); N->v1b = dupf(t[13], 64);
    N->v2w = dupf(t[14], 64); N->v2b = dupf(t[15], 1);
    return N;
}

/* ResidualBlock variant, tensors[24] exactly as include/az_engine.h az_load_weights_resnet documents (BN folded). */
orc_net *orc_resnet_create(int n, const float *const *t)
{
    int nn = n * n;
    orc_net *N = (orc_net *)calloc(1, sizeof *N);
    N->n = n; N->kind = 1;
    N->rw[0] = repack_conv(t[0], 64, 4); N->rb[0] = dupf(t[1], 64);
    for (int i = 0; i < 6; i++) { N->rw[1 + i] = repack_conv(t[2 + 2 * i], 64, 64); N->rb[1 + i] = dupf(t[3 + 2 * i], 64); }
    N->pcw = dupf(t[14], 2 * 64); N->pcb = dupf(t[15], 2);
    N->vcw = dupf(t[16], 64);     N->vcb = dupf(t[17], 1);
    N->pfw = dupf(t[18], (size_t)nn * 2 * nn); N->pfb = dupf(t[19], nn);
    N->v1w = dupf(t[20], (size_t)64 * nn);     N->v1b = dupf(t[21], 64);
    N->v2w = dupf(t[22], 64); N->v2b = dupf(t[23], 1);
    return N;
}

void orc_net_free(orc_net *N)
{
    if (!N) return;
    for (int i = 0; i < 7; i++) { free(N->rw[i]); free(N->rb[i]); }
    free(N->c1w); free(N->c1b); free(N->c2w); free(N->c2b); free(N->c3w); free(N->c3b);
    free(N->pcw); free(N->pcb); free(N->pfw); free(N->pfb);
    free(N->vcw); free(N->vcb); free(N->v1w); free(N->v1b); free(N->v2w); free(N->v2b);
    free(N);
}

static void conv3x3_core(int n, int cin, int cout, const float *in, const float *w, const float *b, float *out, int residual);
/* 3x3 same-padding conv + bias + ReLU, canonical chain; in [cin][nn], out [cout][nn] */
static void conv3x3_relu(int n, int cin, int cout, const float *in, const float *w, const float *b, float *out)
{
    conv3x3_core(n, cin, cout, in, w, b, out, 0);
}
/* residual != 0: out = relu((acc + bias) + out), the skip connection of a ResidualBlock updated in place */
static void conv3x3_core(int n, int cin, int cout, const float *in, const float *w, const float *b, float *out, int residual)
{
    int nn = n * n;
    float acc[128];
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) {
            for (int co = 0; co < cout; co++) acc[co] = 0.0f;
            for (int t = 0; t < 9; t++) {
                int rr = r + t / 3 - 1, cc = c + t % 3 - 1;
                if (rr < 0 || rr >= n || cc < 0 || cc >= n) continue; /* fmaf(0,w,acc)==acc: exact skip */
                for (int ci = 0; ci < cin; ci++) {
                    float x = in[ci * nn + rr * n + cc];
                    const float *wr = w + ((size_t)t * cin + ci) * cout;
                    for (int co = 0; co < cout; co++) acc[co] = fmaf(x, wr[co], acc[co]);
                }
            }
            for (int co = 0; co < cout; co++) {
                float v = acc[co] + b[co];
                if (residual) v = v + out[co * nn + r * n + c];
                out[co * nn + r * n + c] = v > 0.0f ? v : 0.0f;
            }
        }
}

/* net.py:55-72 forward on one encoded state; outputs raw logits[nn] and tanh value */
/* Canonical order of an FC output over K inputs (shared by definition with the HIP engine, csrc/az_net.h fc_chain_groups):
 * the inputs are cut into FOUR contiguous blocks of 16 * ceil(ceil(K / 16) / 4) inputs; each block is one k-ordered fmaf chain
 * from +0 and the output is (p0 + p1) + (p2 + p3), bias added by the caller.  (Round 3: was a single chain.  torch's own
 * order is unknowable either way -- net.py:65,69 -- and the result stays within the tolerances the tests grant it.) */
static float fc_dot(const float *x, const float *w, int K)
{
    const int blk = 16 * ((((K + 15) / 16) + 3) / 4);
    float p[4];
    for (int c = 0; c < 4; c++) {
        float acc = 0.0f;
        const int k1 = (c + 1) * blk < K ? (c + 1) * blk : K;
        for (int k = c * blk; k < k1; k++) acc = fmaf(x[k], w[k], acc);
        p[c] = acc;
    }
    return (p[0] + p[1]) + (p[2] + p[3]);
}

static void net_forward(const orc_net *N, const float *planes, float *logits, float *value)
{
    int n = N->n, nn = n * n;
    static __thread float a1[32 * ORC_MAXNN], a2[64 * ORC_MAXNN], a3[128 * ORC_MAXNN];
    static __thread float pf[4 * ORC_MAXNN], vf[2 * ORC_MAXNN];
    if (N->kind == 1) {
        /* stem, then 3 x { h = relu(conv1(x)); x = relu(conv2(h) + x) } with BN folded (legacy/resnet/example.py:9-27) */
        float *x = a2, *h = a3;
        conv3x3_relu(n, 4, 64, planes, N->rw[0], N->rb[0], x);
        for (int blk = 0; blk < 3; blk++) {
            conv3x3_core(n, 64, 64, x, N->rw[1 + 2 * blk], N->rb[1 + 2 * blk], h, 0);
            conv3x3_core(n, 64, 64, h, N->rw[2 + 2 * blk], N->rb[2 + 2 * blk], x, 1);
        }
        for (int pos = 0; pos < nn; pos++) {
            for (int c = 0; c < 2; c++) {
                float acc = 0.0f;
                for (int ci = 0; ci < 64; ci++) acc = fmaf(x[ci * nn + pos], N->pcw[c * 64 + ci], acc);
                float v = acc + N->pcb[c];
                pf[c * nn + pos] = v > 0.0f ? v : 0.0f;
            }
            float acc = 0.0f;
            for (int ci = 0; ci < 64; ci++) acc = fmaf(x[ci * nn + pos], N->vcw[ci], acc);
            float v = acc + N->vcb[0];
            vf[pos] = v > 0.0f ? v : 0.0f;
        }
        for (int j = 0; j < nn; j++) logits[j] = fc_dot(pf, N->pfw + (size_t)j * 2 * nn, 2 * nn) + N->pfb[j];
        float hh[64];
        for (int i = 0; i < 64; i++) {
            float v = fc_dot(vf, N->v1w + (size_t)i * nn, nn) + N->v1b[i];
            hh[i] = v > 0.0f ? v : 0.0f;
        }
        float acc = 0.0f;
        for (int i = 0; i < 64; i++) acc = fmaf(hh[i], N->v2w[i], acc);
        *value = orc_tanhf(acc + N->v2b[0]);
        return;
    }
    conv3x3_relu(n, 4, 32, planes, N->c1w, N->c1b, a1);
    conv3x3_relu(n, 32, 64, a1, N->c2w, N->c2b, a2);
    conv3x3_relu(n, 64, 128, a2, N->c3w, N->c3b, a3);
    for (int pos = 0; pos < nn; pos++) {
        for (int c = 0; c < 4; c++) {
            float acc = 0.0f;
            for (int ci = 0; ci < 128; ci++) acc = fmaf(a3[ci * nn + pos], N->pcw[c * 128 + ci], acc);
            float v = acc + N->pcb[c];
            pf[c * nn + pos] = v > 0.0f ? v : 0.0f;
        }
        for (int c = 0; c < 2; c++) {
            float acc = 0.0f;
            for (int ci = 0; ci < 128; ci++) acc = fmaf(a3[ci * nn + pos], N->vcw[c * 128 + ci], acc);
            float v = acc + N->vcb[c];
            vf[c * nn + pos] = v > 0.0f ? v : 0.0f;
        }
    }
    for (int j = 0; j < nn; j++) logits[j] = fc_dot(pf, N->pfw + (size_t)j * 4 * nn, 4 * nn) + N->pfb[j];
    float h[64];
    for (int i = 0; i < 64; i++) {
        float v = fc_dot(vf, N->v1w + (size_t)i * 2 * nn, 2 * nn) + N->v1b[i];
        h[i] = v > 0.0f ? v : 0.0f;
    }
    float acc = 0.0f;
    for (int i = 0; i < 64; i++) acc = fmaf(h[i], N->v2w[i], acc);
    *value = orc_tanhf(acc + N->v2b[0]);
}

/* controller.py:49 softmax over all n^2 logits (no legality mask), canonical wave order */
static void softmax_canon(int nn, const float *logits, float *P)
{
    float mx[64], part[64];
    for (int l = 0; l < 64; l++) { mx[l] = -INFINITY; part[l] = 0.0f; }
    for (int j = 0; j < nn; j++) if (logits[j] > mx[j & 63]) mx[j & 63] = logits[j];
    float m = -INFINITY;
    for (int l = 0; l < 64; l++) if (mx[l] > m) m = mx[l];
    for (int j = 0; j < nn; j++) { P[j] = orc_expf(logits[j] - m); }
    for (int j = 0; j < nn; j++) part[j & 63] = part[j & 63] + P[j]; /* per-lane, increasing j */
    float s = butterfly64f(part);
    for (int j = 0; j < nn; j++) P[j] = P[j] / s;
}

void orc_net_eval(const orc_net *N, const float *planes, float *logits, float *P, float *value)
{
    net_forward(N, planes, logits, value);
    softmax_canon(N->n * N->n, logits, P);
}

/* synthetic evaluator (build-owned test hook; same integer hash in make_golden.py and the HIP engine) */
static void synth_eval(const orc_state *s, float *P, float *value)
{
    int nn = s->n * s->n;
    uint32_t hs = 0;
    for (int i = 0; i < nn; i++) {
        uint32_t code = s->cell[i] == 0 ? 0u : (s->cell[i] == s->player ? 1u : 2u);
        hs ^= fmix32((uint32_t)i * 3u + code + 0x9E3779B9u);
    }
    hs ^= fmix32(0x51ED270Bu + (uint32_t)(s->last + 1));
    for (int i = 0; i < nn; i++) {
        uint32_t r = fmix32(hs + (uint32_t)(i + 1) * 0x9E3779B1u);
        P[i] = (float)(((r >> 8) & 0xFFFFu) + 1u) * 0x1p-23f;
    }
    int vv = (int)(fmix32(hs ^ 0x7F4A7C15u) & 0x1FFu);
    *value = (float)(vv - 256) / 256.0f;
}

void orc_synth_eval(int n, const uint8_t *board, int player, int last, float *P, float *value)
{
    orc_state s; st_init(&s, n, 5);
    memcpy(s.cell, board, (size_t)n * n); s.player = player; s.last = last;
    synth_eval(&s, P, value);
}

/* ------------------------------------------------------------------ MCTS (mcts.py) */
typedef struct {
    int n, k, S;
    double c_puct, alpha, w;
    int eval_kind;              /* 0 = net, 1 = synthetic */
    const float *log_table;     /* log_table[N] = np.log(float32(N)+1e-8) as float32, N = 0..S; NULL -> logf */
    int reuse;                  /* opt-in subtree reuse between the plies of a self-play game (include/az_engine.h,
                                   az_set_subtree_reuse); 0 = the reference's behaviour (new root every run, mcts.py:106) */
    int vl;                     /* opt-in virtual-loss batching (include/az_engine.h, az_set_virtual_loss): leaves selected per
                                   evaluation batch; 0 = the reference's sequential loop (mcts.py:123-141).  vl = 1 runs the
                                   batched code with batches of one, which must reproduce the reference exactly. */
    int leaf_sym;               /* opt-in random-symmetry leaf evaluation (include/az_engine.h, az_set_leaf_symmetry; SURVEY 8f-2's
                                   optional half, README.md:61,82): every evaluation shows the net one of the 8 dihedral symmetries
                                   of the position, chosen by a fixed hash of (game_id, ply, evaluation index); 0 = the reference */
    int game_id;                /* ... the game's global name: the low 32 bits of its seed, seed0 + id (single searches: 0) */
} orc_cfg;

typedef struct {
    int parent, action;
    double prior;   /* mcts.py:63 float(policy[r,c]) */
    int N;          /* mcts.py:37 */
    double W;       /* mcts.py:38 */
    int first, cnt; /* children block (insertion = row-major legal order, mcts.py:58) */
    int vl;         /* virtual-loss mode only: simulations of the current batch that pass through this node */
} orc_node;

typedef struct {
    orc_node *nodes; int used, cap;
    long expansions, sims, terminal_hits, depth_sum;
    int retained;               /* subtree reuse: nodes[0] is an already expanded root kept from the previous ply */
    long reused_roots;
    long dup_sims;              /* virtual-loss mode: simulations that landed on a leaf already pending in their batch */
} orc_tree;

/* dihedral symmetries: k < 4 = np.rot90 k times (counter-clockwise), k >= 4 = rot90(fliplr(x), k - 4) (games.py:183-197);
 * source cell of output cell (i, j): sym(x)[i][j] = x[orc_sym_src(k, i, j)] */
static int orc_sym_src(int k, int i, int j, int n)
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
/* the symmetry evaluation idx (0 = the root, s + 1 = simulation s) of the search of game `game` at ply `ply` shows the net */
static int orc_leaf_sym(int game, int ply, int idx)
{
    return (int)(fmix32((uint32_t)game * 0x9E3779B1u ^ fmix32((uint32_t)ply * 0x85EBCA6Bu + (uint32_t)idx + 1u)) & 7u);
}

/* The net's raw outputs for a position shown under symmetry t: planes are in board order; the net sees sym_t(planes); the
 * logits come back in BOARD order (logit of board cell j = the net's logit at the image cell j moved to). */
void orc_net_eval_sym(const orc_net *N, int n, const float *planes, int t, float *logits_board, float *value)
{
    const int nn = n * n, ti = t < 4 ? ((4 - t) & 3) : t;
    float img[4 * ORC_MAXNN], limg[ORC_MAXNN];
    for (int ch = 0; ch < 4; ch++)
        for (int c = 0; c < nn; c++) img[ch * nn + c] = planes[ch * nn + orc_sym_src(t, c / n, c % n, n)];
    net_forward(N, img, limg, value);
    for (int j = 0; j < nn; j++) logits_board[j] = limg[orc_sym_src(ti, j / n, j % n, n)];
}
int orc_leaf_sym_of(int game, int ply, int idx) { return orc_leaf_sym(game, ply, idx); }

/* idx: which evaluation of the search this is (0 = the root, s + 1 = simulation s, also within a virtual-loss batch); idx < 0: no symmetry */
static void evaluate(const orc_cfg *cfg, const orc_net *net, const orc_state *s, float *P, float *v, int ply, int idx)
{
    if (cfg->eval_kind == 1) { synth_eval(s, P, v); return; }
    float planes[4 * ORC_MAXNN], logits[ORC_MAXNN];
    st_encode(s, planes);
    if (cfg->leaf_sym && idx >= 0)      /* the usual canonical softmax runs on the logits brought back to board order */
        orc_net_eval_sym(net, s->n, planes, orc_leaf_sym(cfg->game_id, ply, idx), logits, v);
    else
        net_forward(net, planes, logits, v);
    softmax_canon(s->n * s->n, logits, P);
}

/* mcts.py:50-64 expand: one child per legal action, row-major */
static void expand(orc_tree *t, int node, const orc_state *s, const float *P)
{
    int nn = s->n * s->n;
    t->nodes[node].first = t->used;
    int cnt = 0;
    for (int a = 0; a < nn; a++) {
        if (s->cell[a]) continue;
        orc_node *c = &t->nodes[t->used++];
        c->parent = node; c->action = a; c->prior = (double)P[a]; c->N = 0; c->W = 0.0; c->first = -1; c->cnt = 0; c->vl = 0;
        cnt++;
    }
    t->nodes[node].cnt = cnt;
}

/* numpy pairwise sum of doubles (np.add.reduce on a contiguous array), result = 0.0 + pw(a) */
static double pw_sum(const double *a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return pw_sum(a, n2) + pw_sum(a + n2, n - n2);
}

/* mcts.py:144-177: visit counts -> tempered softmax -> pi, sampled action (np.random.choice with one uniform u) */
static int extract_policy(const orc_cfg *cfg, const orc_tree *t, int nn, double T, double u, float *pi_out)
{
    const orc_node *root = &t->nodes[0];
    int A = root->cnt;
    double e[ORC_MAXNN];
    for (int j = 0; j < nn; j++) pi_out[j] = 0.0f;
    if (A == 0) return -1;
    float lg[ORC_MAXNN];
    for (int i = 0; i < A; i++) {
        int N = t->nodes[root->first + i].N;
        lg[i] = cfg->log_table ? cfg->log_table[N] : logf((float)N + 1e-8f);
    }
    if (T <= 1e-7) {
        /* temperature clipped to the Python float 1e-7: the whole expression stays float32;
           differences between distinct counts exceed 1e4, so exp() is exactly 0 or 1 */
        float y[ORC_MAXNN], m = -INFINITY, ssum = 0.0f;
        for (int i = 0; i < A; i++) { y[i] = lg[i] / 1e-7f; if (y[i] > m) m = y[i]; }
        for (int i = 0; i < A; i++) { y[i] = (y[i] - m) == 0.0f ? 1.0f : orc_expf(y[i] - m); }
        /* float32 pairwise sum of 0/1 values is exact */
        for (int i = 0; i < A; i++) ssum += y[i];
        for (int i = 0; i < A; i++) e[i] = (double)(y[i] / ssum);
    } else {
        double m = -INFINITY;
        for (int i = 0; i < A; i++) { e[i] = (double)lg[i] / T; if (e[i] > m) m = e[i]; }
        for (int i = 0; i < A; i++) e[i] = orc_exp(e[i] - m);
        double s = 0.0 + pw_sum(e, A);
        if (s < 1e-8 || s != s) for (int i = 0; i < A; i++) e[i] = 1.0 / (double)A;
        else for (int i = 0; i < A; i++) e[i] = e[i] / s;
    }
    for (int i = 0; i < A; i++) pi_out[t->nodes[root->first + i].action] = (float)e[i];
    /* RandomState.choice: cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(u, 'right') */
    double cdf[ORC_MAXNN], run = 0.0;
    for (int i = 0; i < A; i++) { run = (i == 0) ? e[0] : run + e[i]; cdf[i] = run; }
    double last = cdf[A - 1];
    int idx = 0;
    for (int i = 0; i < A; i++) { cdf[i] = cdf[i] / last; if (cdf[i] <= u) idx = i + 1; }
    if (idx >= A) idx = A - 1;
    return t->nodes[root->first + idx].action;
}

/* mcts.py:101-183 MCTS.run.  noise = Dirichlet sample over the legal cells (row-major) or NULL. */
static int mcts_run(const orc_cfg *cfg, const orc_net *net, const orc_state *root_state, double T,
                    const double *noise, double u, orc_tree *t, float *pi_out, int *maxdepth_out)
{
    int n = cfg->n, nn = n * n;
    float P[ORC_MAXNN], v;
    int first_sim = 0;
    int ply = 0;                                      /* stones on the board = plies played (leaf-symmetry hash) */
    for (int a = 0; a < nn; a++) ply += root_state->cell[a] != 0;
    orc_node *root = &t->nodes[0];
    if (t->retained) {
        /* subtree reuse (not in the reference): the root kept from the previous ply is not evaluated again; its
         * children keep priors and statistics, the fresh Dirichlet sample is mixed in with a new root's arithmetic,
         * and only the simulations that top the children's visits up to S are run.  root->N = visits of its children,
         * the quantity a fresh root's N holds after the same number of simulations. */
        int carried = 0;
        float om = (float)(1.0 - cfg->w);
        for (int i = 0; i < root->cnt; i++) {
            orc_node *c = &t->nodes[root->first + i];
            carried += c->N;
            if (noise) {
                float scaled = om * (float)c->prior;
                c->prior = (double)(float)((double)scaled + cfg->w * noise[i]);
            }
        }
        root->N = carried;
        first_sim = carried;
        t->retained = 0;
        t->reused_roots++;
    } else {
    t->used = 1;
    root->parent = -1; root->action = -1; root->prior = 1.0; root->N = 0; root->W = 0.0; root->first = -1; root->cnt = 0; root->vl = 0;

    evaluate(cfg, net, root_state, P, &v, ply, 0);    /* mcts.py:109 (root value discarded) */
    if (noise) {                                      /* mcts.py:113-116, arithmetic per SURVEY Q8 */
        float om = (float)(1.0 - cfg->w);
        int i = 0;
        for (int a = 0; a < nn; a++) {
            if (root_state->cell[a]) continue;
            float scaled = om * P[a];
            P[a] = (float)((double)scaled + cfg->w * noise[i]);
            i++;
        }
    }
    expand(t, 0, root_state, P);                      /* mcts.py:120 */
    }
    int maxd = 0;
    if (cfg->vl > 0) {
        /* Virtual-loss batching (NOT in the reference; its TODO list names it, mcts.py:17-22).  The S simulations run in
         * batches of L = cfg->vl: the L leaves of a batch are selected one after another from the same tree, each
         * selection seeing the earlier ones of its batch as one visit that lost (N + 1, W - 1 on every edge of their
         * paths); then the batch is evaluated together, and the simulations are finished in selection order: the virtual
         * visit is taken back, the leaf is expanded and its value backed up exactly like mcts.py:136-141.  A selection
         * that ends on a leaf an earlier simulation of the batch is already waiting on ("duplicate") does not evaluate or
         * expand again; it backs up that leaf's value.  Terminal leaves are scored at once as in mcts.py:132-134. */
        enum { VL_MAX = 32 };
        const int L = cfg->vl > VL_MAX ? VL_MAX : cfg->vl;
        int leaf[VL_MAX], kind[VL_MAX], dup_of[VL_MAX], depth_of[VL_MAX];   /* kind: 0 expand, 1 terminal, 2 duplicate */
        double tval[VL_MAX];
        float vals[VL_MAX];
        static __thread float Pb[VL_MAX][ORC_MAXNN];
        static __thread orc_state sb[VL_MAX];
        for (int done = first_sim; done < cfg->S; ) {
            const int nb = cfg->S - done < L ? cfg->S - done : L;
            for (int j = 0; j < nb; j++) {
                int node = 0, depth = 0;
                orc_state s = *root_state;
                s.winner = root_state->winner;
                while (t->nodes[node].cnt > 0 && !st_terminal(&s)) {
                    const orc_node *nd = &t->nodes[node];
                    double sq = sqrt((double)(nd->N + nd->vl) + 1e-8);
                    int best = -1; double bs = 0.0;
                    for (int i = 0; i < nd->cnt; i++) {
                        const orc_node *c = &t->nodes[nd->first + i];
                        int Nv = c->N + c->vl;
                        double Wv = c->W - (double)c->vl;
                        double Q = Nv ? Wv / (double)Nv : 0.0;
                        double sc = Q + ((cfg->c_puct * c->prior) * sq) / (double)(1 + Nv);
                        if (best < 0 || sc > bs) { best = i; bs = sc; }
                    }
                    node = nd->first + best;
                    st_apply(&s, t->nodes[node].action);
                    depth++;
                }
                leaf[j] = node; depth_of[j] = depth; dup_of[j] = -1;
                if (st_terminal(&s)) {
                    kind[j] = 1;
                    tval[j] = s.winner == RES_DRAW ? 0.0 : (s.winner == s.player ? 1.0 : -1.0);
                } else {
                    kind[j] = 0;
                    for (int i = 0; i < j; i++)
                        if (kind[i] == 0 && leaf[i] == node) { kind[j] = 2; dup_of[j] = i; break; }
                    if (kind[j] == 0) sb[j] = s;
                }
                for (int nd = node; nd >= 0; nd = t->nodes[nd].parent) t->nodes[nd].vl += 1;
            }
            for (int j = 0; j < nb; j++)
                if (kind[j] == 0) evaluate(cfg, net, &sb[j], Pb[j], &vals[j], ply, done + j + 1);   /* simulation done + j: evaluation done + j + 1 */
            for (int j = 0; j < nb; j++) {
                const int node = leaf[j];
                for (int nd = node; nd >= 0; nd = t->nodes[nd].parent) t->nodes[nd].vl -= 1;
                double value;
                if (kind[j] == 1) { value = tval[j]; t->terminal_hits++; }
                else if (kind[j] == 0) { expand(t, node, &sb[j], Pb[j]); value = (double)vals[j]; t->expansions++; }
                else { value = (double)vals[dup_of[j]]; t->dup_sims++; }
                if (depth_of[j] > maxd) maxd = depth_of[j];
                t->sims++; t->depth_sum += depth_of[j];
                double val = -value;
                for (int nd = node; nd >= 0; nd = t->nodes[nd].parent) {
                    t->nodes[nd].N += 1;
                    t->nodes[nd].W += val;
                    val = -val;
                }
            }
            done += nb;
        }
        if (maxdepth_out) *maxdepth_out = maxd;
        return extract_policy(cfg, t, nn, T, u, pi_out);
    }
    for (int sim = first_sim; sim < cfg->S; sim++) {  /* mcts.py:123 */
        int node = 0, depth = 0;
        orc_state s = *root_state;                    /* clone */
        s.winner = root_state->winner;
        while (t->nodes[node].cnt > 0 && !st_terminal(&s)) {   /* mcts.py:127 */
            const orc_node *nd = &t->nodes[node];
            double sq = sqrt((double)nd->N + 1e-8);
            int best = -1; double bs = 0.0;
            for (int i = 0; i < nd->cnt; i++) {       /* mcts.py:71-74, first max wins */
                const orc_node *c = &t->nodes[nd->first + i];
                double Q = c->N ? c->W / (double)c->N : 0.0;
                double sc = Q + ((cfg->c_puct * c->prior) * sq) / (double)(1 + c->N);
                if (best < 0 || sc > bs) { best = i; bs = sc; }
            }
            node = nd->first + best;
            st_apply(&s, t->nodes[node].action);
            depth++;
        }
        double value;
        if (st_terminal(&s)) {                        /* mcts.py:132-134 */
            value = s.winner == RES_DRAW ? 0.0 : (s.winner == s.player ? 1.0 : -1.0);
            t->terminal_hits++;
        } else {                                      /* mcts.py:136-138 */
            evaluate(cfg, net, &s, P, &v, ply, sim + 1);
            expand(t, node, &s, P);
            value = (double)v;
            t->expansions++;
        }
        if (depth > maxd) maxd = depth;
        t->sims++; t->depth_sum += depth;
        double val = -value;                          /* mcts.py:141, 76-82 */
        for (int nd = node; nd >= 0; nd = t->nodes[nd].parent) {
            t->nodes[nd].N += 1;
            t->nodes[nd].W += val;
            val = -val;
        }
    }
    if (maxdepth_out) *maxdepth_out = maxd;
    return extract_policy(cfg, t, nn, T, u, pi_out);
}

static orc_tree *tree_new(const orc_cfg *cfg)
{
    orc_tree *t = (orc_tree *)calloc(1, sizeof *t);
    t->cap = 1 + (cfg->S + 1) * cfg->n * cfg->n;
    t->nodes = (orc_node *)malloc(sizeof(orc_node) * (size_t)t->cap);
    return t;
}
static void tree_free(orc_tree *t) { free(t->nodes); free(t); }

/* subtree reuse: the subtree below root child `action` becomes the whole tree (breadth-first copy, children blocks stay
 * contiguous and in row-major order).  Returns 0 and leaves the tree alone when that child was never expanded. */
static int tree_reroot(orc_tree *t, int action)
{
    const orc_node *root = &t->nodes[0];
    int c = -1;
    for (int i = 0; i < root->cnt; i++)
        if (t->nodes[root->first + i].action == action) c = root->first + i;
    if (c < 0 || t->nodes[c].cnt == 0) return 0;
    orc_node *nn_ = (orc_node *)malloc(sizeof(orc_node) * (size_t)t->cap);
    int *old_of = (int *)malloc(sizeof(int) * (size_t)t->cap);
    int used = 1;
    nn_[0] = t->nodes[c]; nn_[0].parent = -1; old_of[0] = c;
    for (int i = 0; i < used; i++) {
        const orc_node *o = &t->nodes[old_of[i]];
        if (o->cnt == 0) continue;
        nn_[i].first = used;
        for (int j = 0; j < o->cnt; j++) {
            nn_[used] = t->nodes[o->first + j];
            nn_[used].parent = i;
            old_of[used] = o->first + j;
            used++;
        }
    }
    free(t->nodes); free(old_of);
    t->nodes = nn_; t->used = used; t->retained = 1;
    return 1;
}

/* single search from an arbitrary position; also returns the root children's statistics */
int orc_search(const orc_cfg *cfg, const orc_net *net, const uint8_t *board, int player, int last, double T,
               const double *noise, double u, float *pi_out, int32_t *N_out, double *W_out, float *P_out,
               int *nexp_out, int *maxd_out)
{
    int nn = cfg->n * cfg->n;
    orc_state s; st_init(&s, cfg->n, cfg->k);
    memcpy(s.cell, board, (size_t)nn); s.player = player; s.last = last;
    orc_tree *t = tree_new(cfg);
    int maxd = 0;
    int a = mcts_run(cfg, net, &s, T, noise, u, t, pi_out, &maxd);
    for (int j = 0; j < nn; j++) { if (N_out) N_out[j] = 0; if (W_out) W_out[j] = 0.0; if (P_out) P_out[j] = 0.0f; }
    const orc_node *root = &t->nodes[0];
    for (int i = 0; i < root->cnt; i++) {
        const orc_node *c = &t->nodes[root->first + i];
        if (N_out) N_out[c->action] = c->N;
        if (W_out) W_out[c->action] = c->W;
        if (P_out) P_out[c->action] = (float)c->prior;
    }
    if (nexp_out) *nexp_out = 1 + (int)t->expansions; /* nodes with children, root included */
    if (maxd_out) *maxd_out = maxd;
    tree_free(t);
    return a;
}

/* ------------------------------------------------------------------ self-play game (self_play.py:48-73) */
/* noise_tape: concatenation over plies m of Dirichlet samples of length nn-m; u_tape[m]; T_table[m].
 * Outputs per ply: board before the move (absolute cells), mover, last action, pi, root visit counts, action.
 * z[m] per self_play.py:71 (99 if the game was cut at maxply before terminal). Returns number of plies. */
int orc_selfplay_game(const orc_cfg *cfg, const orc_net *net, const double *noise_tape, const double *u_tape,
                      const double *T_table, int maxply, uint8_t *boards, uint8_t *movers, int16_t *lasts,
                      float *pis, int32_t *visits, int16_t *actions, int8_t *z, int *result_out, long *counters)
{
    int nn = cfg->n * cfg->n;
    orc_state s; st_init(&s, cfg->n, cfg->k);
    orc_tree *t = tree_new(cfg);
    int m = 0; size_t noff = 0;
    while (!st_terminal(&s) && m < maxply) {                 /* self_play.py:52 */
        if (boards) memcpy(boards + (size_t)m * nn, s.cell, (size_t)nn);
        movers[m] = (uint8_t)s.player;
        if (lasts) lasts[m] = (int16_t)s.last;
        int a = mcts_run(cfg, net, &s, T_table[m], noise_tape ? noise_tape + noff : NULL, u_tape[m], t,
                         pis + (size_t)m * nn, NULL);       /* self_play.py:54-58 */
        if (visits) {
            for (int j = 0; j < nn; j++) visits[(size_t)m * nn + j] = 0;
            for (int i = 0; i < t->nodes[0].cnt; i++) {
                const orc_node *c = &t->nodes[t->nodes[0].first + i];
                visits[(size_t)m * nn + c->action] = c->N;
            }
        }
        actions[m] = (int16_t)a;
        noff += (size_t)(nn - m);
        st_apply(&s, a);                                     /* self_play.py:64 */
        m++;
        if (cfg->reuse && !st_terminal(&s) && m < maxply) tree_reroot(t, a);
    }
    int res = st_terminal(&s) ? s.winner : RES_NONE;
    for (int i = 0; i < m; i++)                              /* self_play.py:71 */
        z[i] = (int8_t)(res == RES_NONE ? 99 : (res == RES_DRAW ? 0 : (movers[i] == res ? 1 : -1)));
    if (result_out) *result_out = res;
    if (counters) { counters[0] = t->expansions; counters[1] = t->sims; counters[2] = t->terminal_hits; counters[3] = t->depth_sum; counters[4] = m - t->reused_roots; counters[5] = t->dup_sims; }
    tree_free(t);
    return m;
}

/* ------------------------------------------------------------------ arena game (evaluator.py:50-104) */
/* candidate = X (1), baseline = O (2); odd game index => O moves first (evaluator.py:64-69).
 * T_table[step] = evaluator.temperature_schedule(step); step bookkeeping per evaluator.py:71-90.
 * u_tape[p]: one uniform per ply (no root noise).  Returns result (RES_*), writes moves. */
int orc_arena_game(const orc_cfg *cfg, const orc_net *cand, const orc_net *base, int game_index,
                   const double *u_tape, const double *T_table, int16_t *actions, double *temps, int *nply_out)
{
    int nn = cfg->n * cfg->n;
    orc_state s; st_init(&s, cfg->n, cfg->k);
    s.player = (game_index % 2 == 0) ? 1 : 2;
    orc_tree *t = tree_new(cfg);
    int cstep = 0, bstep = 0, p = 0;
    float pi[ORC_MAXNN];
    while (!st_terminal(&s)) {
        const orc_net *net = s.player == 1 ? cand : base;
        int step = s.player == 1 ? cstep : bstep;
        double T = T_table[step];
        if (temps) temps[p] = T;
        int a = mcts_run(cfg, net, &s, T, NULL, u_tape[p], t, pi, NULL);
        actions[p] = (int16_t)a;
        st_apply(&s, a);
        if (s.player == 1) cstep++; else bstep++;          /* evaluator.py:87-90: NEW side to move */
        p++;
        if (p >= nn) { st_terminal(&s); break; }
    }
    st_terminal(&s);
    if (nply_out) *nply_out = p;
    int r = s.winner;
    tree_free(t);
    return r;
}

/* ------------------------------------------------------------------ augmentation (self_play.py:94-108) */
/* out_states[k] = rot90(state, k, dims (1,2)) CCW; out_pis[k] = rot90(pi) ONCE for every k (bug-compatible, Q16) */
void orc_augment(int n, const float *state4, const float *pi, float *out_states, float *out_pis)
{
    int nn = n * n;
    for (int k = 0; k < 4; k++) {
        for (int c = 0; c < 4; c++)
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) {
                    /* np.rot90 k times CCW: out[i][j] = in[si][sj] */
                    int si, sj;
                    switch (k) {
                    case 0: si = i; sj = j; break;
                    case 1: si = j; sj = n - 1 - i; break;
                    case 2: si = n - 1 - i; sj = n - 1 - j; break;
                    default: si = n - 1 - j; sj = i; break;
                    }
                    out_states[((size_t)k * 4 + c) * nn + i * n + j] = state4[c * nn + si * n + sj];
                }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) out_pis[(size_t)k * nn + i * n + j] = pi[j * n + (n - 1 - i)];
    }
}

/* z label truth table, self_play.py:71 / alphazero/tests/tests.py:11-13 */
int orc_zlabel(int result, int mover) { return result == RES_DRAW ? 0 : (mover == result ? 1 : -1); }

/* exposed math for tests */
float orc_test_expf(float x) { return orc_expf(x); }
float orc_test_tanhf(float x) { return orc_tanhf(x); }
double orc_test_exp(double x) { return orc_exp(x); }
double orc_test_pwsum(const double *a, int n) { return 0.0 + pw_sum(a, n); }
