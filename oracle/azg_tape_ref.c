/*
 * azg_tape_ref.c -- ORACLE side of the random tape + third-party arithmetic restatements.
 * TEST INFRASTRUCTURE ONLY (see azg_oracle.h header).
 *
 * The reference draws every random number from numpy's global legacy MT19937 stream, interleaved over all
 * games of a worker (alphazero/MCTS.pyx:79,199; alphazero/SelfPlayAgent.pyx:46,81,84,160).  "Identical seeds"
 * is therefore defined through an explicit, counter-based *random tape* (DESIGN.md "Random tape"): the golden
 * generator monkeypatches np.random.{shuffle,choice,dirichlet,random_sample} in the reference process with the
 * functions below, so the reference, this oracle and the HIP engine all consume the same numbers.
 *
 * All floating point here is +,-,*,/ and sqrt on IEEE doubles (correctly rounded on x86-64 and gfx950 alike),
 * compiled with -ffp-contract=off: results are bit-identical between gcc and hipcc.
 */
#include "azg_oracle.h"
#include <math.h>
#include <string.h>

static inline uint64_t sm64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31; return z;
}

uint64_t azo_tape_u64(uint64_t seed, uint64_t stream, uint64_t ctr) {
    uint64_t z = sm64(seed + 0x9E3779B97F4A7C15ULL * (stream + 1));
    z = sm64(z ^ (0xD1B54A32D192ED03ULL * (ctr + 1)));
    return sm64(z + 0x9E3779B97F4A7C15ULL);
}

static inline double u53(uint64_t z) { return (double)(z >> 11) * (1.0 / 9007199254740992.0); }            /* [0,1) */
static inline double u52_open(uint64_t z) { return ((double)(z >> 12) + 0.5) * (1.0 / 4503599627370496.0); } /* (0,1) */

double azo_tape_uniform(uint64_t seed, uint64_t stream, uint64_t ctr) { return u53(azo_tape_u64(seed, stream, ctr)); }

/* ---- replay of RECORDED draws (the MT19937 tier of "identical seeds"): a stream registered here takes its shuffles,
 * Dirichlet vectors and choice uniforms from arrays indexed by the tape counter instead of the counter-based tape -- the draws the
 * reference itself made on numpy's global stream under np.random.seed(s) (tests/golden/c4_mt19937_agent.npz), so that the oracle can be
 * held to the reference move for move on the CPU.  Same layout as the product's azg_set_random_tape.  Test infrastructure. */
typedef struct { uint64_t stream; const int16_t *ranks; const double *u; const int32_t *noise_off; const float *noise_pool; int len; } azo_replay;
static azo_replay g_replay[256];
static int g_nreplay = 0;
void azo_tape_set_replay(uint64_t stream, const int16_t *ranks, const double *u, const int32_t *noise_off, const float *noise_pool, int len) {
    for (int i = 0; i < g_nreplay; i++) if (g_replay[i].stream == stream) { g_replay[i] = (azo_replay){stream, ranks, u, noise_off, noise_pool, len}; return; }
    if (g_nreplay < 256) g_replay[g_nreplay++] = (azo_replay){stream, ranks, u, noise_off, noise_pool, len};
}
void azo_tape_clear_replay(void) { g_nreplay = 0; }
static const azo_replay *replay_of(uint64_t stream) {
    for (int i = 0; i < g_nreplay; i++) if (g_replay[i].stream == stream) return &g_replay[i];
    return 0;
}

/* Replaces np.random.shuffle(list) at MCTS.pyx:79: element i gets key tape(ctr+i); the new order is ascending
 * (key, i).  pos[i] = rank of element i. */
void azo_tape_shuffle_pos(uint64_t seed, uint64_t stream, uint64_t ctr, int k, int32_t *pos) {
    const azo_replay *rp = replay_of(stream);
    if (rp && rp->ranks) { for (int i = 0; i < k; i++) pos[i] = ctr + (uint64_t)i < (uint64_t)rp->len ? rp->ranks[ctr + (uint64_t)i] : i; return; }
    uint64_t key[1024];
    for (int i = 0; i < k; i++) key[i] = azo_tape_u64(seed, stream, ctr + (uint64_t)i);
    for (int i = 0; i < k; i++) {
        int r = 0;
        for (int j = 0; j < k; j++) r += (key[j] < key[i]) || (key[j] == key[i] && j < i);
        pos[i] = r;
    }
}

/* Replaces np.random.choice(n, p=p) at SelfPlayAgent.pyx:160 with numpy's legacy algorithm
 * (numpy/random/mtrand.pyx RandomState.choice: cdf = p.cumsum() in double; cdf /= cdf[-1];
 *  idx = cdf.searchsorted(uniform, side='right')). */
int azo_tape_choice(uint64_t seed, uint64_t stream, uint64_t ctr, const float *p, int n) {
    const azo_replay *rp = replay_of(stream);
    double u = (rp && rp->u && ctr < (uint64_t)rp->len) ? rp->u[ctr] : u53(azo_tape_u64(seed, stream, ctr));
    double total = 0.0;
    for (int i = 0; i < n; i++) total += (double)p[i];
    double acc = 0.0; int idx = 0;
    for (int i = 0; i < n; i++) { acc += (double)p[i]; if (acc / total <= u) idx = i + 1; }
    if (idx >= n) idx = n - 1;
    return idx;
}

/* ---- deterministic log / exp ------------------------------------------------------------------ */
double azo_det_log(double x) {
    /* x > 0, finite. x = m * 2^e, m in [sqrt(1/2), sqrt(2)) */
    uint64_t b; memcpy(&b, &x, 8);
    int e = (int)((b >> 52) & 0x7FF);
    if (e == 0) { x *= 18014398509481984.0; memcpy(&b, &x, 8); e = (int)((b >> 52) & 0x7FF) - 54; } /* subnormal */
    e -= 1023;
    b = (b & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
    double m; memcpy(&m, &b, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    /* log(m) = 2s * (1 + z/3 + z^2/5 + ... + z^11/23) */
    double r = 1.0 / 23.0;
    r = r * z + 1.0 / 21.0; r = r * z + 1.0 / 19.0; r = r * z + 1.0 / 17.0; r = r * z + 1.0 / 15.0;
    r = r * z + 1.0 / 13.0; r = r * z + 1.0 / 11.0; r = r * z + 1.0 / 9.0;  r = r * z + 1.0 / 7.0;
    r = r * z + 1.0 / 5.0;  r = r * z + 1.0 / 3.0;  r = r * z + 1.0;
    double lm = 2.0 * s * r;
    return (double)e * 0.6931471803691238 + ((double)e * 1.9082149292705877e-10 + lm);
}

double azo_det_exp(double x) {
    if (x < -745.0) return 0.0;
    if (x > 709.0) x = 709.0;
    double t = x * 1.4426950408889634;
    long long k = (long long)(t + (t < 0 ? -0.5 : 0.5));
    double r = (x - (double)k * 0.6931471803691238) - (double)k * 1.9082149292705877e-10;
    /* exp(r), |r| <= 0.35, Taylor degree 14 */
    double p = 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0; p = p * r + 1.0 / 479001600.0; p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;    p = p * r + 1.0 / 362880.0;    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;       p = p * r + 1.0 / 720.0;       p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;         p = p * r + 1.0 / 6.0;         p = p * r + 0.5;
    p = p * r + 1.0;                p = p * r + 1.0;
    /* scale by 2^k in two steps so that subnormal results round once at the end only approximately; exactness
       is not needed, determinism is */
    int k1 = (int)(k / 2), k2 = (int)k - k1;
    uint64_t b1 = (uint64_t)(k1 + 1023) << 52, b2 = (uint64_t)(k2 + 1023) << 52;
    double s1, s2; memcpy(&s1, &b1, 8); memcpy(&s2, &b2, 8);
    return p * s1 * s2;
}

/* ---- gamma / dirichlet ------------------------------------------------------------------------ */
typedef struct { uint64_t key, sub, j; } substream;
static inline uint64_t ss_next(substream *s) { return azo_tape_u64(s->key, s->sub, s->j++); }

static double ss_normal(substream *s) {
    for (;;) {
        double a = 2.0 * u52_open(ss_next(s)) - 1.0;
        double b = 2.0 * u52_open(ss_next(s)) - 1.0;
        double r = a * a + b * b;
        if (r < 1.0 && r > 0.0) return a * sqrt(-2.0 * azo_det_log(r) / r);
    }
}

static double ss_gamma(substream *s, double alpha) {
    double boost = 1.0, a = alpha;
    if (a < 1.0) { double u = u52_open(ss_next(s)); boost = azo_det_exp(azo_det_log(u) / a); a = a + 1.0; }
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x, v;
        do { x = ss_normal(s); v = 1.0 + c * x; } while (v <= 0.0);
        v = v * v * v;
        double u = u52_open(ss_next(s));
        double x2 = x * x;
        if (u < 1.0 - 0.0331 * (x2 * x2)) return d * v * boost;
        if (azo_det_log(u) < 0.5 * x2 + d * (1.0 - v + azo_det_log(v))) return d * v * boost;
    }
}

/* Replaces np.random.dirichlet([alpha]*k) at MCTS.pyx:199-201 (normalised gamma variates, numpy legacy
 * normalisation order: acc += val[i] sequentially; val[i] *= 1/acc). */
void azo_tape_dirichlet(uint64_t seed, uint64_t stream, uint64_t ctr, int k, double alpha, double *out) {
    const azo_replay *rp = replay_of(stream);
    if (rp && rp->noise_off && ctr < (uint64_t)rp->len && rp->noise_off[ctr] >= 0) {      /* (float32, as MCTS.pyx:198-200 casts the vector) */
        for (int i = 0; i < k; i++) out[i] = (double)rp->noise_pool[rp->noise_off[ctr] + i];
        return;
    }
    uint64_t key = azo_tape_u64(seed, stream, ctr);
    double acc = 0.0;
    for (int i = 0; i < k; i++) { substream s = { key, (uint64_t)i, 0 }; out[i] = ss_gamma(&s, alpha); }
    for (int i = 0; i < k; i++) acc += out[i];
    double inv = 1.0 / acc;
    for (int i = 0; i < k; i++) out[i] = out[i] * inv;
}

/* ---- numpy float32 arithmetic restatements ---------------------------------------------------- */
/* numpy pairwise summation for contiguous float32 (numpy/_core/src/umath/loops_utils.h.src,
 * @TYPE@_pairwise_sum, PW_BLOCKSIZE = 128), as used by np.sum at MCTS.pyx:245,252,320,321. */
static float pairwise_f32(const float *a, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8]; int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2; n2 -= n2 % 8;
        return pairwise_f32(a, n2) + pairwise_f32(a + n2, n - n2);
    }
}
float azo_np_sum_f32(const float *a, int n) { return pairwise_f32(a, n); }

/* float32_array ** python_float (numpy 2.x): the exponent is a weak scalar -> float32; fast paths for
 * 1 (copy), 2 (square), 0.5 (sqrt); everything else powf.  Bit-exact tier: exponents 1 and 2 only. */
float azo_np_pow_f32(float x, double e) {
    if (e == 1.0) return x;
    if (e == 2.0) return x * x;
    if (e == 0.5) return sqrtf(x);
    return powf(x, (float)e);
}

/* Synthetic evaluator: positive float32 policy / value rows normalised with numpy's float32 sum, derived
 * from the tape so that fixtures only need the seed. */
void azo_fake_eval(uint64_t seed, uint64_t slot, uint64_t sim, int A, int nv, float *p, float *v) {
    uint64_t s2 = seed ^ 0x5EEDFACE0DDBA11ULL, stream = (slot << 24) ^ sim;
    for (int a = 0; a < A; a++) p[a] = (float)((azo_tape_u64(s2, stream, (uint64_t)a) >> 40) + 1) * (1.0f / 16777216.0f);
    float sp = azo_np_sum_f32(p, A);
    for (int a = 0; a < A; a++) p[a] = p[a] / sp;
    for (int j = 0; j < nv; j++) v[j] = (float)((azo_tape_u64(s2, stream, (uint64_t)(A + j)) >> 40) + 1) * (1.0f / 16777216.0f);
    float sv = azo_np_sum_f32(v, nv);
    for (int j = 0; j < nv; j++) v[j] = v[j] / sv;
}
