// azg_device.h -- device-side building blocks of the MI355X self-play engine (gfx950, wave64).
//
// Layout (DESIGN.md "Data layout in HBM"):
//   Node      32-byte record, one per tree node; the k children of a node are k consecutive records in list order
//             (the reference's shuffled Node._children, alphazero/MCTS.pyx:76-79), so one wavefront reads a whole
//             child block with two coalesced dwordx4 loads per lane.
//   TreeHdr   per tree, one 64-byte line: the root node itself, arena cursor, leaf record of the last find_leaf.
// Arithmetic follows the C that Cython generates from MCTS.pyx; this file must be compiled with
// -ffp-contract=off (no FMA contraction) -- bit-exact visit counts depend on it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/azg.h"

#define AZG_DEV __device__ __forceinline__
#define AZG_HD __host__ __device__ __forceinline__

namespace azg {

// ------------------------------------------------------------------------------------------------ node store
struct __attribute__((aligned(32))) Node {
    int32_t  n;            // visits                      Node.n   (MCTS.pyx:55)
    float    q;            // mean value, parent's mover   Node.q   (:53)
    float    p;            // prior                        Node.p   (:56)
    float    v;            // own-viewpoint first value    Node.v   (:54)
    int32_t  first_child;  // index of child block in this tree's arena, -1 = no children   Node._children (:50)
    uint16_t a;            // action                       Node.a   (:51)
    uint16_t nchild;
    uint8_t  player;       // player to move               Node.player (:57)
    uint8_t  e;            // terminal flags bit j = e[j]  Node.e   (:52)
    uint16_t pad0;
    int32_t  pad1;
};
static_assert(sizeof(Node) == 32, "Node must be 32 bytes");

// Per tree, 64 bytes = one cache line that every kernel reads with ONE load: the ROOT node itself (MCTS._root lives here, not in
// the node store: update_root copies the chosen child in), the arena cursor, and the record of the last find_leaf, so that
// process_results needs no dependent load to learn what the leaf was.
struct __attribute__((aligned(64))) TreeHdr {
    Node     root;         // MCTS._root
    int32_t  base;         // offset of the live semi-space inside this tree's node store: 0 or cap (see k_compact)
    int32_t  alloc;        // arena cursor (next free node index in the live space)
    int32_t  depth;        // MCTS.depth
    int32_t  max_depth;    // MCTS.max_depth
    int32_t  leaf;         // MCTS._curnode after find_leaf: node index, LEAF_IS_ROOT = the root
    int32_t  leaf_fc;      // its child block
    uint32_t leaf_info;    // nchild | e << 16 | player << 24
    int32_t  expanded;     // last find_leaf took the n == 0 branch
};
static_assert(sizeof(TreeHdr) == 64, "TreeHdr must be one 64-byte line");
enum { LEAF_IS_ROOT = -1 };

// One level of the last find_leaf path, 16 bytes: the chosen node, who moved into it, and a snapshot of its (n, q) taken when
// best_child had the child block in registers -- nothing else touches a tree between find_leaf and process_results, so the
// backup is pure stores.
struct __attribute__((aligned(16))) PathEnt { uint32_t idx_mover; int32_t n; float q; uint32_t pad; };


// Engine view passed by value to every kernel.
struct View {
    Node     *nodes;       // [trees][2][cap]: two semi-spaces per tree, hdr.base selects the live one
    TreeHdr  *hdr;         // [trees]
    PathEnt  *path;        // [trees][maxd]  X_1..X_depth of the last find_leaf
    azg_state *states;     // [B] root states
    azg_state *leaf_states;// [B]
    uint64_t *tape_ctr;    // [B]
    int32_t  *next_reset;  // [B]
    int32_t  *hist_len;    // [B]
    azg_state *hist_state; // [B][max_turns]
    float    *hist_pi;     // [B][max_turns][A]
    int32_t  *last_action; // [B]
    int32_t  *fin_flag;    // [B] 0 none, 1 finished
    int32_t  *fin_ridx;    // [B] result index
    int32_t  *fin_counted; // [B]
    int32_t  *fin_soff;    // [B] sample offset
    int64_t  *slot_sims, *slot_exp;   // [B]
    int32_t  *gcount;      // [8]: 0 games_played, 1 num_results, 2 num_examples, 3 error, 4 max_nodes, 5 max nodes kept by a compaction
    float    *ex_obs, *ex_pi, *ex_z;  // examples
    uint8_t  *res_ws; int32_t *res_turns, *res_slot;
    const float *temp_table;
    int32_t B, T, cap, maxd, arena, ex_cap, res_cap, temp_len, compact_reserve;
    int32_t add_noise, add_temp, symmetric, reset_thr, games_cap, max_hist;
    float cpuct, fpu_reduction, noise_frac, root_temp, arena_temp;
    uint64_t seed, slot_base;
    // recorded shuffles (azg_set_shuffle_tape; null = the counter-based tape): ranks [B][perm_len] int16, the rank of child i of the
    // expansion that starts at tape counter c is perm_tape[slot][c + i] -- the replay of np.random.shuffle permutations RECORDED from
    // the reference running on its own MT19937 stream (np.random.seed(s)), MCTS.pyx:79
    const int16_t *perm_tape; int32_t perm_len;
    // ... and, with them, the other two draws of a self-play game (azg_set_random_tape), indexed by the same per-slot tape counter:
    // u_tape [B][perm_len] double -- the uniform np.random.choice drew for the move made at counter c (SelfPlayAgent.pyx:160);
    // noise_off [B][perm_len] int32 -- offset into noise_pool (float32) of the np.random.dirichlet vector mixed into the root priors at
    // counter c (MCTS.pyx:197-206), one value per child in list order
    const double *u_tape; const int32_t *noise_off; const float *noise_pool; int32_t noise_len;
    unsigned long long *dbg;   // AZG_TREE_TIMING builds only: s_memtime stamps [B][16] of the last simulation of every slot
};

// phase stamps of the tree kernels (measurement builds: hipcc -DAZG_TREE_TIMING; tools/time_tree.py reads them)
#ifdef AZG_TREE_TIMING
#define AZG_TSTAMP(ev, slot, lane, i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    if ((ev).dbg && (lane) == 0) (ev).dbg[(size_t)(slot) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AZG_TSTAMP(ev, slot, lane, i) do { } while (0)
#endif

enum { GC_GAMES = 0, GC_RESULTS = 1, GC_EXAMPLES = 2, GC_ERROR = 3, GC_MAXNODES = 4, GC_MAXLIVE = 5, GC_BOUNDS_SITE = 6 };

// ------------------------------------------------------------------------------------------------ random tape
// Definition in DESIGN.md "Random tape"; independent re-implementation of the spec (the oracle has its own).
AZG_HD uint64_t sm64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31; return z;
}
AZG_HD uint64_t tape_u64(uint64_t seed, uint64_t stream, uint64_t ctr) {
    uint64_t z = sm64(seed + 0x9E3779B97F4A7C15ULL * (stream + 1));
    z = sm64(z ^ (0xD1B54A32D192ED03ULL * (ctr + 1)));
    return sm64(z + 0x9E3779B97F4A7C15ULL);
}
AZG_HD double u53(uint64_t z) { return (double)(z >> 11) * (1.0 / 9007199254740992.0); }
AZG_HD double u52_open(uint64_t z) { return ((double)(z >> 12) + 0.5) * (1.0 / 4503599627370496.0); }

AZG_HD double det_log(double x) {
    union { double d; uint64_t u; } c; c.d = x;
    int e = (int)((c.u >> 52) & 0x7FF);
    if (e == 0) { c.d = x * 18014398509481984.0; e = (int)((c.u >> 52) & 0x7FF) - 54; }
    e -= 1023;
    c.u = (c.u & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
    double m = c.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double f = m - 1.0, s = f / (2.0 + f), z = s * s;
    double r = 1.0 / 23.0;
    r = r * z + 1.0 / 21.0; r = r * z + 1.0 / 19.0; r = r * z + 1.0 / 17.0; r = r * z + 1.0 / 15.0;
    r = r * z + 1.0 / 13.0; r = r * z + 1.0 / 11.0; r = r * z + 1.0 / 9.0;  r = r * z + 1.0 / 7.0;
    r = r * z + 1.0 / 5.0;  r = r * z + 1.0 / 3.0;  r = r * z + 1.0;
    double lm = 2.0 * s * r;
    return (double)e * 0.6931471803691238 + ((double)e * 1.9082149292705877e-10 + lm);
}
AZG_HD double det_exp(double x) {
    if (x < -745.0) return 0.0;
    if (x > 709.0) x = 709.0;
    double t = x * 1.4426950408889634;
    long long k = (long long)(t + (t < 0 ? -0.5 : 0.5));
    double r = (x - (double)k * 0.6931471803691238) - (double)k * 1.9082149292705877e-10;
    double p = 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0; p = p * r + 1.0 / 479001600.0; p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;    p = p * r + 1.0 / 362880.0;    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;       p = p * r + 1.0 / 720.0;       p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;         p = p * r + 1.0 / 6.0;         p = p * r + 0.5;
    p = p * r + 1.0;                p = p * r + 1.0;
    int k1 = (int)(k / 2), k2 = (int)k - k1;
    union { double d; uint64_t u; } s1, s2;
    s1.u = (uint64_t)(k1 + 1023) << 52; s2.u = (uint64_t)(k2 + 1023) << 52;
    return p * s1.d * s2.d;
}

struct SubStream { uint64_t key, sub, j; };
AZG_HD uint64_t ss_next(SubStream &s) { return tape_u64(s.key, s.sub, s.j++); }
AZG_HD double ss_normal(SubStream &s) {
    for (;;) {
        double a = 2.0 * u52_open(ss_next(s)) - 1.0;
        double b = 2.0 * u52_open(ss_next(s)) - 1.0;
        double r = a * a + b * b;
        if (r < 1.0 && r > 0.0) return a * sqrt(-2.0 * det_log(r) / r);
    }
}
AZG_HD double ss_gamma(SubStream &s, double alpha) {
    double boost = 1.0, a = alpha;
    if (a < 1.0) { double u = u52_open(ss_next(s)); boost = det_exp(det_log(u) / a); a = a + 1.0; }
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x, v;
        do { x = ss_normal(s); v = 1.0 + c * x; } while (v <= 0.0);
        v = v * v * v;
        double u = u52_open(ss_next(s));
        double x2 = x * x;
        if (u < 1.0 - 0.0331 * (x2 * x2)) return d * v * boost;
        if (det_log(u) < 0.5 * x2 + d * (1.0 - v + det_log(v))) return d * v * boost;
    }
}

// ------------------------------------------------------------------------------------------------ wave helpers
AZG_DEV int   rl(int x, int lane)   { return __builtin_amdgcn_readlane(x, lane); }
AZG_DEV float rl(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
AZG_DEV unsigned rl(unsigned x, int lane) { return (unsigned)__builtin_amdgcn_readlane((int)x, lane); }
AZG_DEV double rl(double x, int lane) {
    union { double d; int i[2]; } c; c.d = x;
    c.i[0] = __builtin_amdgcn_readlane(c.i[0], lane); c.i[1] = __builtin_amdgcn_readlane(c.i[1], lane);
    return c.d;
}
AZG_DEV uint64_t rl(uint64_t x, int lane) {
    union { uint64_t u; int i[2]; } c; c.u = x;
    c.i[0] = __builtin_amdgcn_readlane(c.i[0], lane); c.i[1] = __builtin_amdgcn_readlane(c.i[1], lane);
    return c.u;
}
// DPP lane permutations inside a row of 16 (no LDS crossbar, one VALU slot each): butterfly steps xor 1, xor 2, then the
// half-row and row mirrors -- after the four every lane holds the reduction of its row of 16
template <int CTRL> AZG_DEV int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, false); }
template <int CTRL> AZG_DEV float dpp_f(float x) { return __int_as_float(dpp_i<CTRL>(__float_as_int(x))); }
enum { DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140, DPP_WAVE_ROR1 = 0x13C };
AZG_DEV float wave_max(float m) {                              // every lane returns the maximum over the 64 lanes
    m = fmaxf(m, dpp_f<DPP_QUAD_XOR1>(m)); m = fmaxf(m, dpp_f<DPP_QUAD_XOR2>(m));
    m = fmaxf(m, dpp_f<DPP_ROW_HALF_MIRROR>(m)); m = fmaxf(m, dpp_f<DPP_ROW_MIRROR>(m));
    return fmaxf(fmaxf(rl(m, 0), rl(m, 16)), fmaxf(rl(m, 32), rl(m, 48)));
}
// max over lanes [0, N) only (N a power of two <= 64; the other lanes must hold -inf or be ignored by the caller)
template <int N> AZG_DEV float wave_max_n(float m) {
    m = fmaxf(m, dpp_f<DPP_QUAD_XOR1>(m)); m = fmaxf(m, dpp_f<DPP_QUAD_XOR2>(m));
    if (N > 4) m = fmaxf(m, dpp_f<DPP_ROW_HALF_MIRROR>(m));
    if (N > 8) m = fmaxf(m, dpp_f<DPP_ROW_MIRROR>(m));
    float r = rl(m, 0);
    if (N > 16) r = fmaxf(r, rl(m, 16));
    if (N > 32) r = fmaxf(fmaxf(r, rl(m, 32)), rl(m, 48));
    return r;
}
AZG_DEV float wave_sum_f(float m) {                            // every lane returns the SAME float sum over the 64 lanes (fixed tree)
    m += dpp_f<DPP_QUAD_XOR1>(m); m += dpp_f<DPP_QUAD_XOR2>(m); m += dpp_f<DPP_ROW_HALF_MIRROR>(m); m += dpp_f<DPP_ROW_MIRROR>(m);
    return (rl(m, 0) + rl(m, 16)) + (rl(m, 32) + rl(m, 48));
}
AZG_DEV int wave_max_i(int m) {                                // every lane returns the maximum over the 64 lanes
    m = max(m, dpp_i<DPP_QUAD_XOR1>(m)); m = max(m, dpp_i<DPP_QUAD_XOR2>(m)); m = max(m, dpp_i<DPP_ROW_HALF_MIRROR>(m)); m = max(m, dpp_i<DPP_ROW_MIRROR>(m));
    return max(max(rl(m, 0), rl(m, 16)), max(rl(m, 32), rl(m, 48)));
}
template <int CTRL> AZG_DEV double dpp_d(double x) {
    union { double d; int i[2]; } c; c.d = x;
    c.i[0] = dpp_i<CTRL>(c.i[0]); c.i[1] = dpp_i<CTRL>(c.i[1]);
    return c.d;
}
// sum of 64 doubles by a fixed tree -- callers use it only where every partial sum is exact, so that the order does not matter
AZG_DEV double wave_sum_d(double m) {
    m += dpp_d<DPP_QUAD_XOR1>(m); m += dpp_d<DPP_QUAD_XOR2>(m); m += dpp_d<DPP_ROW_HALF_MIRROR>(m); m += dpp_d<DPP_ROW_MIRROR>(m);
    return (rl(m, 0) + rl(m, 16)) + (rl(m, 32) + rl(m, 48));
}
AZG_DEV int wave_sum_i(int m) {                                // every lane returns the sum over the 64 lanes
    m += dpp_i<DPP_QUAD_XOR1>(m); m += dpp_i<DPP_QUAD_XOR2>(m); m += dpp_i<DPP_ROW_HALF_MIRROR>(m); m += dpp_i<DPP_ROW_MIRROR>(m);
    return (rl(m, 0) + rl(m, 16)) + (rl(m, 32) + rl(m, 48));
}
// number of set bits of a wave ballot below this lane (v_mbcnt: two VALU slots)
AZG_DEV int lanes_below(uint64_t ballot) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ballot, 0u)); }
// Ordering point between the lanes of ONE wavefront (every tree function is one wave working on wave-private LDS and on its own
// tree in HBM): earlier LDS / global accesses of the wave have completed before later ones start.  No s_barrier: the functions
// can be called by several waves of a workgroup independently (the two-wave tree launch, the persistent search kernel).
AZG_DEV void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// exclusive prefix sum over lanes (int)
AZG_DEV int wave_excl_scan(int x, int lane) {
    int s = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(s, o); if (lane >= o) s += t; }
    return s - x;
}

// numpy float32 `array ** python_float` (weak-scalar exponent -> float32; fast paths 1, 2, 0.5; else powf).
// powf is evaluated as a double pow rounded to float: correctly rounded except in ~2^-29 of cases (1-ulp tier).
AZG_DEV float np_pow_f32(float x, double e) {
    if (e == 1.0) return x;
    if (e == 2.0) return x * x;
    if (e == 0.5) return sqrtf(x);
    return (float)pow((double)x, (double)(float)e);
}

// numpy float32 pairwise np.sum over m[0..N) held in LDS, evaluated cooperatively by one wave with exactly numpy's association
// order (loops_utils.h.src @TYPE@_pairwise_sum, PW_BLOCKSIZE 128): 8 strided accumulators per <=128-element block, combined
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), tail sequential, blocks combined by the recursion tree.  Every lane returns the sum.
// `scr` = 64 floats of LDS scratch.  The plan (leaf offsets / lengths, combine program) is worked out at compile time from the
// length -- every caller sums a policy-sized vector, N = A -- so the offsets are immediates, the up-to-16 strided reads of a
// leaf are issued together and the combine program runs on registers (a plan read from HBM cost a chain of ~15 dependent scalar
// loads: ~6 k cycles for N = 588).
template <int N> struct NpPlan {
    int nleaves = 0, nprog = 0, off[64] = {}, len[64] = {};
    unsigned char prog[128] = {};
    constexpr void rec(int o, int n) {
        if (n <= 128) { off[nleaves] = o; len[nleaves] = n; nleaves++; prog[nprog++] = 0; return; }
        int n2 = n / 2; n2 -= n2 % 8;
        rec(o, n2); rec(o + n2, n - n2);
        prog[nprog++] = 1;
    }
    constexpr NpPlan() { rec(0, N); }
};
template <int N>
AZG_DEV float np_sum_static(const float *m, float *scr, int lane) {
    static_assert(N >= 8, "n < 8 is a sequential sum (callers handle it)");
    constexpr NpPlan<N> PL{};
    constexpr int NL = PL.nleaves;
    static_assert(NL <= 64, "one scratch slot per leaf");
    const int g = lane >> 3, j = lane & 7;
#pragma unroll
    for (int l0 = 0; l0 < NL; l0 += 8) {
        int off = 0, len = 8;
#pragma unroll
        for (int q = 0; q < 8; q++) if (l0 + q < NL && g == q) { off = PL.off[l0 + q]; len = PL.len[l0 + q]; }
        const int lim = len - (len & 7);
        float v[16];                                                         // (a leaf has at most 128 elements: 16 per lane)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = m[off + min(8 * i, lim - 8) + j];
        float r = v[0];
#pragma unroll
        for (int i = 1; i < 16; i++) if (8 * i < lim) r += v[i];
        r = r + dpp_f<DPP_QUAD_XOR1>(r);
        r = r + dpp_f<DPP_QUAD_XOR2>(r);
        r = r + dpp_f<DPP_ROW_HALF_MIRROR>(r);                               // lane j <-> 7 - j of its group of 8: the xor-4 partner's quad
        float res = r;
        for (int i = lim; i < len; i++) res += m[off + i];
        if (l0 + g < NL && j == 0) scr[l0 + g] = res;
    }
    wave_sync();
    float lf[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) lf[l] = scr[l];
    float st[8];
    int sp = 0, li = 0;
#pragma unroll
    for (int i = 0; i < PL.nprog; i++) {
        if (PL.prog[i] == 0) st[sp++] = lf[li++];
        else { st[sp - 2] = st[sp - 2] + st[sp - 1]; sp--; }
    }
    wave_sync();
    return st[0];
}

// softmax over the A policy logits of one board by one wavefront (NNetArchitecture.py:112-118, exp(log_softmax)): lane owns logits
// lane, lane + 64, ... (A <= 1024).  pol may be global or LDS.
// AC > 0: the row length is known at compile time, so only the ceil(AC / 64) register slots that can hold a logit are touched; the
// others would hold -inf, whose exp is an exact +0 in the lane's sum -- the result is bit-identical to the run-time form.
template <int AC = 0>
AZG_DEV void policy_softmax_row(const float *lg, int lane, int A, float *pol) {
    constexpr int NJ = AC > 0 ? (AC + 63) / 64 : 16;
    float x[NJ], m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int o = lane + 64 * j;
        x[j] = lg[min(o, A - 1)];
        if (o >= A) x[j] = -INFINITY;
        m = fmaxf(m, x[j]);
    }
    m = wave_max(m);                                           // (DPP row reductions + four readlanes: ~60 cycles; six ds_bpermute round
    float sum = 0.f;                                           //  trips -- __shfl_xor -- cost ~600 on a wavefront that has its SIMD alone)
#pragma unroll
    for (int j = 0; j < NJ; j++) { x[j] = __expf(x[j] - m); sum += x[j]; }    // exp(-inf) = 0 for the padding
    sum = wave_sum_f(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < NJ; j++) { const int o = lane + 64 * j; if (o < A) pol[o] = x[j] * inv; }
}
// the same for the NV <= 64 value logits lg[0 .. NV): lane j < NV returns probability j (other lanes 0)
AZG_DEV float value_softmax(const float *lg, int lane, int NV) {
    const float v = lg[min(lane, NV - 1)];
    const float vm = wave_max(lane < NV ? v : -INFINITY);
    const float ev = lane < NV ? __expf(v - vm) : 0.f;
    return ev / wave_sum_f(ev);
}
AZG_DEV void heads_softmax_row(const float *lg, int lane, int A, int NV, float *pol, float *val) {
    policy_softmax_row(lg, lane, A, pol);
    const float pv = value_softmax(lg + A, lane, NV);
    if (lane < NV) val[lane] = pv;
}

}  // namespace azg
