// azg_kernels.h -- the tree kernels.  One wavefront (= one 64-thread workgroup) owns one game slot: block b always
// works on slot b, so a slot's tree stays in the L2 of XCD (b mod 8) across select / backup / advance launches.
//
//   k_select   MCTS.find_leaf   (alphazero/MCTS.pyx:208-228) + Node.best_child/uct (:86-104) + Node.add_children
//              (:76-79) + leaf GameState.observation -> dense NN input row        [SelfPlayAgent.generateBatch]
//   k_backup   MCTS.process_results (:230-289) + Node.update_policy (:81-84) + _add_root_noise (:197-206)
//                                                                                 [SelfPlayAgent.processBatch]
//   k_play / k_finalize / k_emit   SelfPlayAgent.playMoves (alphazero/SelfPlayAgent.pyx:153-202)
//   k_root_*   MCTS.counts / probs / value (:297-344)
#pragma once
#include "azg_games.h"

namespace azg {

struct NodeR {                       // a node held in registers
    int n; float q, p, v; int fc; int a, nchild, player, e;
};
AZG_DEV void unpack(const uint4 &lo, const uint4 &hi, NodeR &r) {
    r.n = (int)lo.x; r.q = __uint_as_float(lo.y); r.p = __uint_as_float(lo.z); r.v = __uint_as_float(lo.w);
    r.fc = (int)hi.x; r.a = hi.y & 0xFFFF; r.nchild = hi.y >> 16; r.player = hi.z & 0xFF; r.e = (hi.z >> 8) & 0xFF;
}
AZG_DEV uint4 pack_hi(int fc, int a, int nchild, int player, int e) {
    return make_uint4((unsigned)fc, (unsigned)(a & 0xFFFF) | ((unsigned)nchild << 16), (unsigned)player | ((unsigned)e << 8), 0u);
}
AZG_DEV void load_node(const Node *p, uint4 &lo, uint4 &hi) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    lo = q[0]; hi = q[1];
}
AZG_DEV void raise_error(const View &ev, int code) { atomicCAS(&ev.gcount[GC_ERROR], 0, code); }

AZG_DEV void init_tree(const View &ev, int tree, int lane) {           // MCTS.__init__/reset (:133-160)
    if (lane == 0) {
        uint4 *q = reinterpret_cast<uint4 *>(ev.nodes + (size_t)tree * ev.cap);
        q[0] = make_uint4(0, 0, 0, 0);
        q[1] = pack_hi(-1, 0xFFFF, 0, 0, 0);
        TreeHdr *h = ev.hdr + tree;
        h->root = 0; h->alloc = 1; h->cur = 0; h->depth = 0; h->max_depth = 0; h->expanded = 0;
    }
}

// Node.add_children (:76-79) for node `idx` of a tree: k stubs, list order = ascending (tape key, index).
// my_a[c] = action of child index c*64+lane (ascending action order).  Returns first_child (or -1 on overflow).
template <class G>
AZG_DEV int add_children(const View &ev, int slot, Node *nodes, TreeHdr *h, int k, const int (&my_a)[(G::MAXK + 63) / 64], int lane) {
    constexpr int NCH = (G::MAXK + 63) / 64;
    int fc = __builtin_amdgcn_readfirstlane(h->alloc);
    if (fc + k > ev.cap) { if (lane == 0) raise_error(ev, AZG_E_TREE_FULL); return -1; }
    uint64_t ctr = ev.tape_ctr[slot];
    uint64_t key[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) key[c] = tape_u64(ev.seed, ev.slot_base + (uint64_t)slot, ctr + (uint64_t)(c * 64 + lane));
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        int i = c * 64 + lane, pos = 0;
#pragma unroll
        for (int c2 = 0; c2 < NCH; c2++) {
            int lim = min(64, k - c2 * 64);
            for (int j = 0; j < lim; j++) {
                uint64_t kj = rl(key[c2], j);
                int jj = c2 * 64 + j;
                pos += (kj < key[c] || (kj == key[c] && jj < i)) ? 1 : 0;
            }
        }
        if (i < k) {
            uint4 *q = reinterpret_cast<uint4 *>(nodes + fc + pos);
            q[0] = make_uint4(0, 0, 0, 0);
            q[1] = pack_hi(-1, my_a[c], 0, 0, 0);
        }
    }
    if (lane == 0) {
        h->alloc = fc + k;
        ev.tape_ctr[slot] = ctr + (uint64_t)k;          // (no global high-water atomic here: 2048 waves on one word cost ~20 us)
    }
    return fc;
}

// ================================================================================================ select
// One wavefront runs find_leaf for one slot; `sink(st, lane)` receives the leaf state (it writes the observation wherever the
// network reads it: the dense batch in HBM for k_select, straight into the tower's LDS image for the fused search kernel).
template <class G, class Sink>
AZG_DEV void select_slot(const View &ev, int slot, int lane, int *act_lds, Sink &&sink) {
    constexpr int NCH = (G::MAXK + 63) / 64;
    typename G::S st = G::load(&ev.states[slot], lane);
    const int tree = ev.arena ? slot * ev.T + st.player : slot;
    Node *nodes = ev.nodes + (size_t)tree * ev.cap;
    TreeHdr *h = ev.hdr + tree;
    uint32_t *path = ev.path + (size_t)tree * ev.maxd;
    int cur = __builtin_amdgcn_readfirstlane(h->root);
    NodeR cn;
    { uint4 lo, hi; load_node(nodes + cur, lo, hi); unpack(lo, hi, cn); }
    int depth = 0;
    const float cpuct = ev.cpuct;
    while (cn.n > 0 && cn.e == 0 && depth < ev.maxd) {                       // MCTS.pyx:213
        const int k = cn.nchild, fc = cn.fc;
        if (k == 0 || fc < 0) { if (lane == 0) raise_error(ev, AZG_E_TREE_FULL); break; }
        uint4 lo[NCH], hi[NCH];
        double seen = 0.0;                                                   // :91 python sum() in double, list order
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            int i = c * 64 + lane;
            if (i < k) load_node(nodes + fc + i, lo[c], hi[c]);
            else { lo[c] = make_uint4(0, 0, 0, 0); hi[c] = make_uint4(0, 0, 0, 0); }
            uint64_t vis = __ballot(i < k && (int)lo[c].x > 0);
            while (vis) { int b = __ffsll((unsigned long long)vis) - 1; seen += (double)rl(__uint_as_float(lo[c].z), b); vis &= vis - 1; }
        }
        const float seen_f = (float)seen;
        const float fpu = (float)((double)cn.v - ((double)ev.fpu_reduction * sqrt((double)seen_f)));   // :92
        // :94 (float)sqrt((double)n): a correctly rounded f32 sqrt of the (exactly representable) count gives the same
        // float -- rounding a 53-bit sqrt to 24 bits is innocuous double rounding (53 >= 2*24 + 2)
        const float sqn = cn.n < (1 << 24) ? sqrtf((float)cn.n) : (float)sqrt((double)cn.n);
        float best = -INFINITY; int bi = 0; NodeR sel = cn;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            int i = c * 64 + lane;
            int ni = (int)lo[c].x;
            float t = ni == 0 ? fpu : __uint_as_float(lo[c].y);
            float u = t + (((cpuct * __uint_as_float(lo[c].z)) * sqn) / ((float)(1 + ni)));               // :87
            if (i >= k) u = -INFINITY;
            constexpr int RW = G::MAXK <= 8 ? 8 : G::MAXK <= 16 ? 16 : G::MAXK <= 32 ? 32 : 64;
            float m = RW == 64 ? wave_max(u) : wave_max_n<RW>(u);
            if (m > best) {                                                  // strict '>' : first max wins (:100)
                uint64_t bal = __ballot(i < k && u == m);
                int b = __ffsll((unsigned long long)bal) - 1;
                best = m; bi = c * 64 + b;
                uint4 slo, shi;
                slo.x = rl(lo[c].x, b); slo.y = rl(lo[c].y, b); slo.z = rl(lo[c].z, b); slo.w = rl(lo[c].w, b);
                shi.x = rl(hi[c].x, b); shi.y = rl(hi[c].y, b); shi.z = rl(hi[c].z, b); shi.w = 0;
                unpack(slo, shi, sel);
            }
        }
        cur = fc + bi;
        if (lane == 0) path[depth] = (uint32_t)cur | ((uint32_t)cn.player << 28);
        cn = sel;
        G::play(st, cn.a);                                                   // :216
        depth++;
    }
    int expanded = 0;
    if (cn.n == 0) {                                                         // :223-226 expand
        const int e = G::win_bits(st);
        int my_a[NCH];
        const int k = G::valid_list(st, lane, act_lds, my_a);
        int fc = add_children<G>(ev, slot, nodes, h, k, my_a, lane);
        if (lane == 0) {
            uint4 *q = reinterpret_cast<uint4 *>(nodes + cur);
            q[1] = pack_hi(fc, cn.a, fc < 0 ? 0 : k, st.player, e);
        }
        expanded = 1;
    }
    if (lane == 0) {
        h->cur = cur; h->depth = depth; h->expanded = expanded;
        if (depth > h->max_depth) h->max_depth = depth;                      // :219-221
        ev.slot_exp[slot] += expanded;
    }
    G::store(st, &ev.leaf_states[slot], lane);
    sink(st, lane);
}

template <class G, typename OT, bool NHWC8 = false>
__global__ __launch_bounds__(64) void k_select(View ev, OT *obs, const int32_t *row_of_slot) {
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;   // sticky device error: stop touching the trees
    __shared__ int act_lds[((G::MAXK + 63) / 64) * 64];
    const int slot = blockIdx.x;
    select_slot<G>(ev, slot, threadIdx.x, act_lds, [&](const typename G::S &st, int lane) {
        if (obs) {
            const int row = row_of_slot ? row_of_slot[slot] : slot;
            if constexpr (NHWC8) G::write_obs_nhwc8(st, (_Float16 *)obs + (size_t)row * G::CELLS * 8, lane);
            else G::template write_obs<OT>(st, obs + (size_t)row * G::OBS, lane);     // SelfPlayAgent.pyx:116-123
        }
    });
}

// ================================================================================================ backup
// One wavefront runs process_results for one slot with its policy row pi[A] and value row vrow[P+1] (HBM or LDS).
// m_lds [max(A, 8)] and scr [64] are wave-private scratch (only used when A >= 8).
template <class G>
AZG_DEV void backup_slot(const View &ev, int slot, int lane, const float *pi, const float *vrow, float *m_lds, float *scr) {
    constexpr int NCH = (G::MAXK + 63) / 64;
    constexpr int A = G::A, P = G::P, NV = P + 1, NE = P + G::HAS_DRAW;
    const int mover0 = __builtin_amdgcn_readfirstlane(ev.states[slot].player);
    const int tree = ev.arena ? slot * ev.T + mover0 : slot;
    Node *nodes = ev.nodes + (size_t)tree * ev.cap;
    TreeHdr *h = ev.hdr + tree;
    const uint32_t *path = ev.path + (size_t)tree * ev.maxd;
    const int cur = __builtin_amdgcn_readfirstlane(h->cur), depth = __builtin_amdgcn_readfirstlane(h->depth);
    const int root = __builtin_amdgcn_readfirstlane(h->root);
    NodeR cn;
    { uint4 lo, hi; load_node(nodes + cur, lo, hi); unpack(lo, hi, cn); }
    float val[NV > NE ? NV : NE];
    int vsize;
    if (cn.e) {                                                              // :234-235 terminal: value = float32(e)
#pragma unroll
        for (int j = 0; j < NE; j++) val[j] = (float)((cn.e >> j) & 1);
        vsize = NE;
    } else {
#pragma unroll
        for (int j = 0; j < NV; j++) val[j] = vrow[j];
        vsize = NV;
        // ---- mask + renormalise the policy over the node's children (:239-245) ----
        const int k = cn.nchild, fc = cn.fc;
        int ca[NCH]; float cp[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            int i = c * 64 + lane;
            ca[c] = i < k ? (int)nodes[fc + i].a : -1;
            cp[c] = i < k ? pi[ca[c]] : 0.f;                                 // pi * valids: valid entries keep pi[a]
        }
        float s;
        const bool at_root = cur == root;
        const bool use_temp = at_root && ev.add_temp;
        if constexpr (A < 8) {
            s = 0.f;                                                         // np.sum, n < 8: sequential in action order
            for (int a = 0; a < A; a++) {
                uint64_t bal = __ballot(ca[0] == a);
                if (bal) s += rl(cp[0], __ffsll((unsigned long long)bal) - 1);
            }
        } else {
            for (int a = lane; a < A; a += 64) m_lds[a] = 0.f;
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NCH; c++) if (ca[c] >= 0) m_lds[ca[c]] = cp[c];
            __syncthreads();
            s = np_sum_wave(m_lds, ev.plan, scr, lane);
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) cp[c] = cp[c] / s;                     // :245
        if (use_temp) {                                                      // :249-252 root temperature
            const double ex = 1.0 / (double)ev.root_temp;
#pragma unroll
            for (int c = 0; c < NCH; c++) cp[c] = np_pow_f32(cp[c], ex);
            float s2;
            if constexpr (A < 8) {
                s2 = 0.f;
                for (int a = 0; a < A; a++) {
                    uint64_t bal = __ballot(ca[0] == a);
                    if (bal) s2 += rl(cp[0], __ffsll((unsigned long long)bal) - 1);
                }
            } else {
                __syncthreads();
#pragma unroll
                for (int c = 0; c < NCH; c++) if (ca[c] >= 0) m_lds[ca[c]] = cp[c];
                __syncthreads();
                s2 = np_sum_wave(m_lds, ev.plan, scr, lane);
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) cp[c] = cp[c] / s2;
        }
        if (at_root && ev.add_noise) {                                       // :197-206 Dirichlet noise (tape)
            const uint64_t ctr = ev.tape_ctr[slot];
            const uint64_t key = tape_u64(ev.seed, ev.slot_base + (uint64_t)slot, ctr);
            const double alpha = 10.83 / (double)k;
            double g[NCH]; double acc = 0.0;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                int i = c * 64 + lane;
                g[c] = 0.0;
                if (i < k) { SubStream ss = { key, (uint64_t)i, 0 }; g[c] = ss_gamma(ss, alpha); }
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) { int lim = min(64, k - c * 64); for (int j = 0; j < lim; j++) acc += rl(g[c], j); }
            const double inv = 1.0 / acc;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                float nz = (float)(g[c] * inv);
                cp[c] = (float)(((double)cp[c] * (1.0 - (double)ev.noise_frac)) + (double)(ev.noise_frac * nz));
            }
            if (lane == 0) ev.tape_ctr[slot] = ctr + 1;
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) { int i = c * 64 + lane; if (i < k) nodes[fc + i].p = cp[c]; }      // update_policy :81-84
    }
    // ---- backup along the path (:265-287): X_j = path[j-1] node, mover = player of X_{j-1}; all levels independent
    const float draw_share = vsize > P ? (val[P] / ((float)P)) : 0.f;
    for (int j0 = 0; j0 < depth; j0 += 64) {
        int j = j0 + lane;
        if (j < depth) {
            uint32_t ent = path[j];
            int idx = (int)(ent & 0x0FFFFFFFu), mover = (int)(ent >> 28);
            float vm = val[0];
#pragma unroll
            for (int pp = 1; pp < P; pp++) if (mover == pp) vm = val[pp];
            float v = vsize > P ? vm + draw_share : vm;                      // _get_value :291-295
            Node *x = nodes + idx;
            int n = x->n; float q = x->q;
            x->q = (((q * (float)n) + (v * 1.0f)) / ((float)(n + 1)));       // :282 (discount == 1)
            if (n == 0) {                                                    // :283-284 (only the leaf can have n == 0)
                float vo = val[0];
#pragma unroll
                for (int pp = 1; pp < P; pp++) if (cn.player == pp) vo = val[pp];
                x->v = vsize > P ? vo + draw_share : vo;
            }
            x->n = n + 1;
        }
    }
    if (lane == 0) { nodes[root].n += 1; ev.slot_sims[slot] += 1; }          // :289
}

template <class G>
__global__ __launch_bounds__(64) void k_backup(View ev, const float *policy, const float *value, const int32_t *row_of_slot) {
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;   // sticky device error: stop touching the trees
    __shared__ float m_lds[G::A < 8 ? 8 : G::A];
    __shared__ float scr[64];
    const int slot = blockIdx.x;
    const int row = row_of_slot ? row_of_slot[slot] : slot;
    backup_slot<G>(ev, slot, threadIdx.x, policy + (size_t)row * G::A, value + (size_t)row * (G::P + 1), m_lds, scr);
}

// backup of simulation k and find_leaf of simulation k + 1 of the same slot in one launch (they are consecutive in the lock-step
// loop, SelfPlayAgent.pyx:87-92, and touch the same tree): one kernel boundary and one pass over the path's cache lines less.
template <class G, typename OT, bool NHWC8 = false>
__global__ __launch_bounds__(64) void k_backup_select(View ev, const float *policy, const float *value, OT *obs, const int32_t *row_of_slot) {
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;
    __shared__ float m_lds[G::A < 8 ? 8 : G::A];
    __shared__ float scr[64];
    __shared__ int act_lds[((G::MAXK + 63) / 64) * 64];
    const int slot = blockIdx.x;
    const int row = row_of_slot ? row_of_slot[slot] : slot;
    backup_slot<G>(ev, slot, threadIdx.x, policy + (size_t)row * G::A, value + (size_t)row * (G::P + 1), m_lds, scr);
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;
    select_slot<G>(ev, slot, threadIdx.x, act_lds, [&](const typename G::S &st, int lane) {
        if (obs) {
            if constexpr (NHWC8) G::write_obs_nhwc8(st, (_Float16 *)obs + (size_t)row * G::CELLS * 8, lane);
            else G::template write_obs<OT>(st, obs + (size_t)row * G::OBS, lane);
        }
    });
}

// The same with the softmaxes of the wide heads folded in: the network hands over LOGITS rows (stride `ld`: A policy logits, then
// P + 1 value logits, as azg_policy_value_heads_f16 leaves them in its workspace) and this wavefront turns its own row into
// probabilities in LDS -- one launch and one HBM round trip of the probabilities less per simulation.  do_select = 0: backup only.
template <class G, typename OT, bool NHWC8 = false>
__global__ __launch_bounds__(64) void k_backup_select_logits(View ev, const float *logits, int ld, OT *obs, const int32_t *row_of_slot,
                                                             int do_select) {
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;
    __shared__ float m_lds[G::A < 8 ? 8 : G::A];
    __shared__ float scr[64];
    __shared__ int act_lds[((G::MAXK + 63) / 64) * 64];
    __shared__ float pi_lds[G::A], v_lds[G::P + 1];
    const int slot = blockIdx.x;
    const int row = row_of_slot ? row_of_slot[slot] : slot;
    heads_softmax_row(logits + (size_t)row * ld, threadIdx.x, G::A, G::P + 1, pi_lds, v_lds);
    __syncthreads();
    backup_slot<G>(ev, slot, threadIdx.x, pi_lds, v_lds, m_lds, scr);
    if (!do_select) return;
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;
    select_slot<G>(ev, slot, threadIdx.x, act_lds, [&](const typename G::S &st, int lane) {
        if (obs) {
            if constexpr (NHWC8) G::write_obs_nhwc8(st, (_Float16 *)obs + (size_t)row * G::CELLS * 8, lane);
            else G::template write_obs<OT>(st, obs + (size_t)row * G::OBS, lane);
        }
    });
}

// ================================================================================================ root stats
// MCTS.probs (:308-329) of a tree's root into LDS pr[A]; every lane returns.  cnt = LDS float counts.
template <class G>
AZG_DEV void root_probs(const View &ev, const Node *nodes, int root_fc, int root_k, float temp, float *cnt, float *pr, float *scr, int lane) {
    constexpr int A = G::A;
    for (int a = lane; a < A; a += 64) cnt[a] = 0.f;
    __syncthreads();
    for (int i = lane; i < root_k; i += 64) cnt[nodes[root_fc + i].a] = (float)nodes[root_fc + i].n;    // counts :297-303
    __syncthreads();
    if (temp == 0.f) {                                                       // :313-317 one-hot first argmax
        float best = cnt[0]; int b = 0;
        for (int a = 1; a < A; a++) if (cnt[a] > best) { best = cnt[a]; b = a; }
        for (int a = lane; a < A; a += 64) pr[a] = a == b ? 1.f : 0.f;
        __syncthreads();
        return;
    }
    float s;
    if constexpr (A < 8) { s = 0.f; for (int a = 0; a < A; a++) s += cnt[a]; }
    else s = np_sum_wave(cnt, ev.plan, scr, lane);
    const double ex = 1.0 / (double)temp;
    for (int a = lane; a < A; a += 64) pr[a] = np_pow_f32(cnt[a] / s, ex);    // :320
    __syncthreads();
    float s2;
    if constexpr (A < 8) { s2 = 0.f; for (int a = 0; a < A; a++) s2 += pr[a]; }
    else s2 = np_sum_wave(pr, ev.plan, scr, lane);
    for (int a = lane; a < A; a += 64) pr[a] = pr[a] / s2;                    // :321
    __syncthreads();
}

template <class G>
__global__ __launch_bounds__(64) void k_root_stats(View ev, int what, float temp, int average, int32_t *counts, float *probs, float *values) {
    constexpr int A = G::A;
    __shared__ float cnt[A < 8 ? 8 : A], pr[A < 8 ? 8 : A], scr[64];
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int mover0 = __builtin_amdgcn_readfirstlane(ev.states[slot].player);
    const int tree = ev.arena ? slot * ev.T + mover0 : slot;
    const Node *nodes = ev.nodes + (size_t)tree * ev.cap;
    const int root = ev.hdr[tree].root;
    const int fc = nodes[root].first_child, k = nodes[root].nchild;
    if (what == 0) {
        for (int a = lane; a < A; a += 64) counts[(size_t)slot * A + a] = 0;
        __syncthreads();
        for (int i = lane; i < k; i += 64) counts[(size_t)slot * A + nodes[fc + i].a] = nodes[fc + i].n;
    } else if (what == 1) {
        root_probs<G>(ev, nodes, fc, k, temp, cnt, pr, scr, lane);
        for (int a = lane; a < A; a += 64) probs[(size_t)slot * A + a] = pr[a];
    } else if (lane == 0) {                                                  // MCTS.value :331-344
        float value = 0.f;
        if (average) {
            double s = 0.0;
            for (int i = 0; i < k; i++) if (nodes[fc + i].n > 0) s += (double)nodes[fc + i].q;
            value = (float)(s / (double)k);
        } else {
            for (int i = 0; i < k; i++) if (nodes[fc + i].q > value && nodes[fc + i].n > 0) value = nodes[fc + i].q;
        }
        values[slot] = value;
    }
}

// MCTS.update_root (:185-195) for one tree; returns false if the action is not a child.
template <class G>
AZG_DEV bool update_root(const View &ev, int slot, int tree, const typename G::S &st, int action, int *act_lds, int lane) {
    constexpr int NCH = (G::MAXK + 63) / 64;
    Node *nodes = ev.nodes + (size_t)tree * ev.cap;
    TreeHdr *h = ev.hdr + tree;
    const int root = __builtin_amdgcn_readfirstlane(h->root);
    int fc = __builtin_amdgcn_readfirstlane(nodes[root].first_child);
    int k = __builtin_amdgcn_readfirstlane((int)nodes[root].nchild);
    if (k == 0) {                                                            // :186-187 unexpanded root: add + shuffle
        int my_a[NCH];
        k = G::valid_list(st, lane, act_lds, my_a);
        fc = add_children<G>(ev, slot, nodes, h, k, my_a, lane);
        if (fc < 0) return false;
        if (lane == 0) { nodes[root].first_child = fc; nodes[root].nchild = (uint16_t)k; }
        __syncthreads();
    }
    int found = -1;
    for (int i0 = 0; i0 < k; i0 += 64) {
        int i = i0 + lane;
        uint64_t bal = __ballot(i < k && (int)nodes[fc + i].a == action);
        if (bal) { found = i0 + __ffsll((unsigned long long)bal) - 1; break; }
    }
    if (found < 0) return false;
    if (lane == 0) h->root = fc + found;
    return true;
}

template <class G>
__global__ __launch_bounds__(64) void k_update_root(View ev, int slot, int action, int32_t *ok) {
    __shared__ int act_lds[((G::MAXK + 63) / 64) * 64];
    const int lane = threadIdx.x;
    typename G::S st = G::load(&ev.states[slot], lane);
    bool good = true;
    for (int t = 0; t < ev.T; t++) good = update_root<G>(ev, slot, slot * ev.T + t, st, action, act_lds, lane) && good;
    if (lane == 0) *ok = good ? 1 : 0;
}

// ================================================================================================ advance
// Phase 1 (per slot): SelfPlayAgent.playMoves :156-176
template <class G>
__global__ __launch_bounds__(64) void k_play(View ev, int record_history) {
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;   // sticky device error: stop touching the trees
    constexpr int A = G::A;
    __shared__ float cnt[A < 8 ? 8 : A], pr[A < 8 ? 8 : A], scr[64];
    __shared__ int act_lds[((G::MAXK + 63) / 64) * 64];
    const int slot = blockIdx.x, lane = threadIdx.x;
    typename G::S st = G::load(&ev.states[slot], lane);
    if (G::win_bits(st) != 0) {                                              // a finished game that was not counted (:179-183 else
        if (lane == 0) ev.fin_flag[slot] = 0;                                // branch) keeps its final state and idles: the reference's
        return;                                                              // agent loop has ended by then (SelfPlayAgent.pyx:79-80)
    }
    const int tree = ev.arena ? slot * ev.T + st.player : slot;
    const Node *nodes = ev.nodes + (size_t)tree * ev.cap;
    const int root = __builtin_amdgcn_readfirstlane(ev.hdr[tree].root);
    const int fc = __builtin_amdgcn_readfirstlane(nodes[root].first_child);
    const int k = __builtin_amdgcn_readfirstlane((int)nodes[root].nchild);
    float temp;
    if (ev.arena) temp = ev.arena_temp;                                      // :158
    else { int t = st.turns < ev.temp_len ? st.turns : ev.temp_len - 1; temp = ev.temp_table[t]; }   // :156-157
    root_probs<G>(ev, nodes, fc, k, temp, cnt, pr, scr, lane);               // :159
    // np.random.choice(A, p=policy) via the tape (:160): cdf in double, first index whose cdf/total > u
    uint64_t ctr = ev.tape_ctr[slot];
    const double u = u53(tape_u64(ev.seed, ev.slot_base + (uint64_t)slot, ctr));
    double total = 0.0;
    for (int a = 0; a < A; a++) { float x = pr[a]; if (x != 0.f) total += (double)x; }
    double acc = 0.0; int action = 0; bool le = true;                        // cdf_i <= u  (searchsorted side='right')
    for (int a = 0; a < A; a++) {
        float x = pr[a];
        if (x != 0.f) { acc += (double)x; le = (acc / total) <= u; }
        if (le) action = a + 1;
    }
    if (action >= A) action = A - 1;
    __syncthreads();
    if (lane == 0) { ev.tape_ctr[slot] = ctr + 1; ev.last_action[slot] = action; }
    if (record_history && !ev.arena && ev.max_hist > 0) {                                       // :161-165 history.append((clone, probs(T=1)))
        const int hl = __builtin_amdgcn_readfirstlane(ev.hist_len[slot]);
        if (hl < ev.max_hist) {
            if (temp != 1.0f) root_probs<G>(ev, nodes, fc, k, 1.0f, cnt, pr, scr, lane);
            float *hp = ev.hist_pi + ((size_t)slot * ev.max_hist + hl) * A;
            for (int a = lane; a < A; a += 64) hp[a] = pr[a];
            G::store(st, &ev.hist_state[(size_t)slot * ev.max_hist + hl], lane);
            if (lane == 0) ev.hist_len[slot] = hl + 1;
        } else if (lane == 0) raise_error(ev, AZG_E_EXAMPLES_FULL);
    }
    __syncthreads();
    bool ok = true;                                                          // :167-170 update_root (every tree in arena)
    if (ev.arena) { for (int t = 0; t < ev.T; t++) ok = update_root<G>(ev, slot, slot * ev.T + t, st, action, act_lds, lane) && ok; }
    else ok = update_root<G>(ev, slot, tree, st, action, act_lds, lane);
    if (!ok && lane == 0) raise_error(ev, AZG_E_INVALID_ACTION);
    G::play(st, action);                                                     // :171
    if (ev.reset_thr) {                                                      // :172-174
        const int nr = __builtin_amdgcn_readfirstlane(ev.next_reset[slot]);
        if (st.turns >= nr) {
            for (int t = 0; t < ev.T; t++) init_tree(ev, slot * ev.T + t, lane);
            if (lane == 0) ev.next_reset[slot] = st.turns + ev.reset_thr;
        }
    }
    const int ws = G::win_bits(st);                                          // :176
    G::store(st, &ev.states[slot], lane);
    if (lane == 0) ev.fin_flag[slot] = ws;
}

// Phase 2 (one wave, slot order): result_queue order, the games_played cap (:179-183) and sample offsets.
template <class G>
__global__ __launch_bounds__(64) void k_finalize(View ev, const int32_t *counted_in) {
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;   // sticky device error: stop touching the trees
    const int lane = threadIdx.x;
    const int nsym = ev.symmetric ? G::NSYM : 1;
    int gp = ev.gcount[GC_GAMES], nr = ev.gcount[GC_RESULTS], ne = ev.gcount[GC_EXAMPLES];
    for (int s0 = 0; s0 < ev.B; s0 += 64) {
        const int s = s0 + lane;
        const int fin = s < ev.B ? (ev.fin_flag[s] != 0) : 0;
        const int rank = wave_excl_scan(fin, lane);
        const int counted = counted_in ? (fin && counted_in[s] != 0) : (fin && (gp + rank < ev.games_cap));
        const int ns = (counted && !ev.arena) ? ev.hist_len[s] * nsym : 0;
        const int soff = wave_excl_scan(ns, lane);
        if (fin) { ev.fin_ridx[s] = nr + rank; ev.fin_counted[s] = counted; ev.fin_soff[s] = ne + soff; }
        const int nfin = wave_sum_i(fin), ncnt = wave_sum_i(counted), nsum = wave_sum_i(ns);
        gp += ncnt; nr += nfin; ne += nsum;
    }
    if (lane == 0) {
        if (nr > ev.res_cap || (ev.ex_cap > 0 && ne > ev.ex_cap)) raise_error(ev, AZG_E_EXAMPLES_FULL);
        ev.gcount[GC_GAMES] = gp; ev.gcount[GC_RESULTS] = nr; ev.gcount[GC_EXAMPLES] = ne;
    }
}

// Phase 3a (per finished, counted slot and symmetry): the samples of that symmetry (:184-196).  One wavefront per (slot,
// symmetry) -- a finished brandubh game is 100 positions x 8 symmetries x 3.3 KB, too much for the slot's single wave.
template <class G>
__global__ __launch_bounds__(64) void k_emit_samples(View ev) {
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;
    constexpr int A = G::A, NV = G::P + 1;
    const int slot = blockIdx.x, kk = blockIdx.y, lane = threadIdx.x;
    const int ws = __builtin_amdgcn_readfirstlane(ev.fin_flag[slot]);
    if (ws == 0 || !ev.fin_counted[slot] || ev.arena || ev.ex_cap <= 0) return;
    const int nsym = ev.symmetric ? G::NSYM : 1;
    if (kk >= nsym) return;
    const int hl = ev.hist_len[slot], soff = ev.fin_soff[slot];
    for (int hI = 0; hI < hl; hI++) {
        const int si = soff + hI * nsym + kk;
        if (si >= ev.ex_cap) continue;
        typename G::S hs = G::load(&ev.hist_state[(size_t)slot * ev.max_hist + hI], lane);
        const float *hp = ev.hist_pi + ((size_t)slot * ev.max_hist + hI) * A;
        typename G::S ss = G::symmetry(hs, kk);
        G::template write_obs<float>(ss, ev.ex_obs + (size_t)si * G::OBS, lane);
        for (int a = lane; a < A; a += 64) ev.ex_pi[(size_t)si * A + G::sym_action(a, kk)] = hp[a];
        if (lane < NV) ev.ex_z[(size_t)si * NV + lane] = (float)((ws >> lane) & 1);
    }
}

// Phase 3b (per finished slot): result record, reset game + trees (:197-200).
template <class G>
__global__ __launch_bounds__(64) void k_emit(View ev) {
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;   // sticky device error: stop touching the trees
    constexpr int A = G::A, NV = G::P + 1;
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int ws = __builtin_amdgcn_readfirstlane(ev.fin_flag[slot]);
    if (ws == 0) return;
    const int ridx = ev.fin_ridx[slot], counted = ev.fin_counted[slot];
    if (ridx < ev.res_cap && lane == 0) {
        for (int j = 0; j < NV; j++) ev.res_ws[(size_t)ridx * NV + j] = (uint8_t)((ws >> j) & 1);
        ev.res_turns[ridx] = ev.states[slot].turns; ev.res_slot[ridx] = slot;
    }
    if (!counted) return;                                                    // :197-200 sit inside the counted branch
    typename G::S st; G::init(st);
    G::store(st, &ev.states[slot], lane);
    if (lane == 0) ev.hist_len[slot] = 0;
    for (int t = 0; t < ev.T; t++) init_tree(ev, slot * ev.T + t, lane);
}

template <class G>
__global__ __launch_bounds__(64) void k_reset(View ev, int first, int count, int reset_state) {
    const int slot = first + blockIdx.x, lane = threadIdx.x;
    if (slot >= first + count) return;
    if (reset_state) { typename G::S st; G::init(st); G::store(st, &ev.states[slot], lane); }
    if (lane == 0) { ev.hist_len[slot] = 0; ev.next_reset[slot] = 0; ev.fin_flag[slot] = 0; }
    for (int t = 0; t < ev.T; t++) init_tree(ev, slot * ev.T + t, lane);
}

// high-water mark of the tree arenas, computed when the counters are read
__global__ __launch_bounds__(256) void k_max_nodes(View ev) {
    int m = 0;
    for (int t = threadIdx.x; t < ev.B * ev.T; t += 256) m = max(m, ev.hdr[t].alloc);
    atomicMax(&ev.gcount[GC_MAXNODES], m);
}

__global__ __launch_bounds__(64) void k_reset_max_depth(View ev) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t < ev.B * ev.T) ev.hdr[t].max_depth = 0;
}

// arena rows (SelfPlayAgent.pyx:117-132): rows grouped by model = player_to_index[mover], slot order inside a group
struct SeatMap { int32_t v[8]; };                            // player_to_index by value: no host copy, graph-capturable

// seat_of_slot (optional): per-slot seat permutation, 4 bits per player (model of player p = (word >> 4p) & 15) -- every
// concurrent game can have its own seating instead of the one permutation per agent of SelfPlayAgent.pyx:44-47.
__global__ __launch_bounds__(64) void k_arena_rows(View ev, SeatMap seat, const uint32_t *seat_of_slot, int32_t *row_of_slot,
                                                   int32_t *rows_per_model) {
    const int lane = threadIdx.x;
    const int32_t *p2i = seat.v;
    int base = 0;
    for (int mi = 0; mi < ev.T; mi++) {
        int cntm = 0;
        for (int s0 = 0; s0 < ev.B; s0 += 64) {
            const int s = s0 + lane;
            int is = 0;
            if (s < ev.B) {
                const int mover = ev.states[s].player;
                is = (seat_of_slot ? (int)((seat_of_slot[s] >> (4 * mover)) & 15u) : p2i[mover]) == mi;
            }
            const int r = wave_excl_scan(is, lane);
            if (is) row_of_slot[s] = base + cntm + r;
            cntm += wave_sum_i(is);
        }
        if (lane == 0) rows_per_model[mi] = cntm;
        base += cntm;
    }
}

}  // namespace azg
