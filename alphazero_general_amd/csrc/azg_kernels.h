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
// The bounds-checked build (hipcc -DAZG_DEBUG_BOUNDS: alphazero_general_amd/build.py --variant debug; the reference turned its own checks
// off, MCTS.pyx:2-6): every node / child-block / path index the tree kernels form is checked against the store's capacity, the tree's
// live allocation and the path length BEFORE it is used.  A violation raises the sticky AZG_E_INTERNAL, records the first site in
// gcount[GC_BOUNDS_SITE] (azg_debug_bounds_site) and the access is skipped.  The product build compiles the checks away.
#ifdef AZG_DEBUG_BOUNDS
AZG_DEV bool bounds_fail(const View &ev, int site) { raise_error(ev, AZG_E_INTERNAL); atomicCAS(&ev.gcount[GC_BOUNDS_SITE], 0, site); return false; }
#define AZG_BOUNDS_OK(ev, site, cond) ((cond) ? true : bounds_fail(ev, site))
#else
#define AZG_BOUNDS_OK(ev, site, cond) true
#endif
// a child block [fc, fc + k) of a tree whose live space holds `alloc` nodes
#define AZG_BLOCK_OK(ev, site, fc, k, alloc) AZG_BOUNDS_OK(ev, site, (fc) >= 0 && (k) > 0 && (k) <= G::MAXK && (fc) + (k) <= (alloc) && (alloc) <= (ev).cap)

// a tree header in registers: one 64-byte line, every lane loads it (same address: one request, broadcast)
struct HdrR {
    NodeR root; int base, alloc, depth, max_depth, leaf, leaf_fc, leaf_k, leaf_e, leaf_player, expanded;
};
AZG_DEV void load_hdr(const TreeHdr *h, HdrR &r) {
    const uint4 *q = reinterpret_cast<const uint4 *>(h);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    unpack(a, b, r.root);
    r.base = __builtin_amdgcn_readfirstlane((int)c.x); r.alloc = __builtin_amdgcn_readfirstlane((int)c.y);
    r.depth = __builtin_amdgcn_readfirstlane((int)c.z); r.max_depth = __builtin_amdgcn_readfirstlane((int)c.w);
    r.leaf = __builtin_amdgcn_readfirstlane((int)d.x); r.leaf_fc = __builtin_amdgcn_readfirstlane((int)d.y);
    const unsigned li = (unsigned)__builtin_amdgcn_readfirstlane((int)d.z);
    r.leaf_k = li & 0xFFFF; r.leaf_e = (li >> 16) & 0xFF; r.leaf_player = li >> 24;
    r.expanded = __builtin_amdgcn_readfirstlane((int)d.w);
}
// the live semi-space of a tree's node store
AZG_DEV Node *tree_nodes(const View &ev, int tree, int base) { return ev.nodes + (size_t)tree * 2 * ev.cap + base; }

AZG_DEV void init_tree(const View &ev, int tree, int lane) {           // MCTS.__init__/reset (:133-160)
    if (lane == 0) {
        uint4 *q = reinterpret_cast<uint4 *>(ev.hdr + tree);
        q[0] = make_uint4(0, 0, 0, 0);                                   // root: n = 0, q = p = v = 0
        q[1] = pack_hi(-1, 0xFFFF, 0, 0, 0);
        q[2] = make_uint4(0, 0, 0, 0);                                   // base, alloc, depth, max_depth
        q[3] = make_uint4((unsigned)LEAF_IS_ROOT, (unsigned)-1, 0, 0);
    }
}

// Node.add_children (:76-79): k stubs appended to the tree's live space, list order = ascending (tape key, index).
// my_a[c] = action of child index c*64+lane (ascending action order), NC = chunks of 64 children actually in use.  `alloc` is the
// arena cursor held by the caller; returns first_child (or -1 on overflow) and advances `alloc`; the caller stores it.
// rank of child i among the k shuffled children = number of (key, index) pairs below its own
struct NoRanks { AZG_DEV bool operator()(int, int, int &) const { return false; } };
template <class G, int NC, class Ranks>
AZG_DEV int add_children(const View &ev, int slot, Node *nodes, int &alloc, int k, const int (&my_a)[(G::MAXK + 63) / 64], uint64_t &ctr, int lane,
                         Ranks &&ranks) {
    const int fc = alloc;
    if (fc + k > ev.cap) { if (lane == 0) raise_error(ev, AZG_E_TREE_FULL); return -1; }
    int pos[NC];
    if (ev.perm_tape) {                                                      // recorded shuffles replayed (see View::perm_tape)
        if (ctr + (uint64_t)k > (uint64_t)ev.perm_len) { if (lane == 0) raise_error(ev, AZG_E_INVALID_ARG); return -1; }
#pragma unroll
        for (int c = 0; c < NC; c++) pos[c] = c * 64 + lane < k ? (int)ev.perm_tape[(size_t)slot * ev.perm_len + ctr + (uint64_t)(c * 64 + lane)] : 0;
    } else if (!(NC == 1 && ranks(k, lane, pos[0]))) {                       // (the two-wave launch has them ready)
        uint64_t key[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) key[c] = tape_u64(ev.seed, ev.slot_base + (uint64_t)slot, ctr + (uint64_t)(c * 64 + lane));
#pragma unroll
        for (int c = 0; c < NC; c++) {
            int i = c * 64 + lane, pc = 0;
#pragma unroll
            for (int c2 = 0; c2 < NC; c2++) {
                int lim = min(64, k - c2 * 64);
                for (int j = 0; j < lim; j++) {
                    uint64_t kj = rl(key[c2], j);
                    int jj = c2 * 64 + j;
                    pc += (kj < key[c] || (kj == key[c] && jj < i)) ? 1 : 0;
                }
            }
            pos[c] = pc;
        }
    }
    if (ev.perm_tape) {                                                      // a replayed rank outside [0, k) would write outside the expansion's k nodes
        bool bad = false;
#pragma unroll
        for (int c = 0; c < NC; c++) bad |= c * 64 + lane < k && (pos[c] < 0 || pos[c] >= k);
        if (__ballot(bad)) { if (lane == 0) raise_error(ev, AZG_E_INVALID_ARG); return -1; }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int i = c * 64 + lane;
        if (i < k && AZG_BOUNDS_OK(ev, 1, pos[c] >= 0 && pos[c] < k && k <= G::MAXK)) {
            uint4 *q = reinterpret_cast<uint4 *>(nodes + fc + pos[c]);
            q[0] = make_uint4(0, 0, 0, 0);
            q[1] = pack_hi(-1, my_a[c], 0, 0, 0);
        }
    }
    alloc = fc + k;
    ctr += (uint64_t)k;                                     // (no global high-water atomic here: 2048 waves on one word cost ~20 us)
    return fc;
}
template <class G, class Ranks>
AZG_DEV int add_children_any(const View &ev, int slot, Node *nodes, int &alloc, int k, const int (&my_a)[(G::MAXK + 63) / 64], uint64_t &ctr, int lane,
                             Ranks &&ranks) {
    constexpr int NCH = (G::MAXK + 63) / 64;
    if constexpr (NCH > 1) { if (k > 64) return add_children<G, NCH>(ev, slot, nodes, alloc, k, my_a, ctr, lane, NoRanks{}); }
    return add_children<G, 1>(ev, slot, nodes, alloc, k, my_a, ctr, lane, ranks);
}

// For lane i: the set of j in [0, 64) whose (tape key, index) pair sorts below (key_i, i), keys of counters ctr .. ctr + 63.
// The rank of child i among k <= 64 children is then popcount(mask & ((1 << k) - 1)): the all-pairs part of the shuffle does
// not depend on k, so it can be prepared before the leaf is known.
template <int NJ = 64>                                                       // only bits j < k <= NJ are ever looked at (NJ = min(64, the game's MAXK))
AZG_DEV uint64_t shuffle_less_mask(const View &ev, int slot, uint64_t ctr, int lane) {
    const uint64_t key = tape_u64(ev.seed, ev.slot_base + (uint64_t)slot, ctr + (uint64_t)lane);
    uint64_t less = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const uint64_t kj = rl(key, j);
        less |= (uint64_t)((kj < key || (kj == key && j < lane)) ? 1 : 0) << j;
    }
    return less;
}

// ================================================================================================ select
// Node.best_child (:86-104) over the k children at nodes[fc ..]: lane i holds child i (+ 64 per further chunk).  Returns the
// list index of the first maximal child and its record in `sel`.
template <class G, int NC>
AZG_DEV int best_child(const View &ev, const Node *nodes, int fc, int k, const NodeR &cn, int lane, NodeR &sel) {
    uint4 lo[NC], hi[NC];
    double seen = 0.0;                                                       // :91 python sum() in double, list order
#pragma unroll
    for (int c = 0; c < NC; c++) {
        int i = c * 64 + lane;
        if (i < k) load_node(nodes + fc + i, lo[c], hi[c]);
        else { lo[c] = make_uint4(0, 0, 0, 0); hi[c] = make_uint4(0, 0, 0, 0); }
    }
    AZG_TSTAMP(ev, blockIdx.x, lane, 10);
    // Python's sum() adds the visited children's priors one by one in double.  When the priors' exponents lie within 22 binades
    // of each other every partial sum of up to 128 24-bit mantissas is exactly representable in a double (22 + 24 + 7 = 53
    // bits), so NO addition rounds and the order is immaterial: the sum is then taken by a DPP tree (~40 instructions) instead
    // of a serial loop of ~12 instructions per visited child (40 visited children at a root late in a move: 2 k cycles).
    bool serial = NC > 1;
    if constexpr (NC == 1) {
        const bool v = lane < k && (int)lo[0].x > 0 && __uint_as_float(lo[0].z) != 0.f;
        const uint64_t vis = __ballot(v);
        if (__popcll(vis) <= 6) serial = true;                                   // (few terms: the loop is cheaper)
        else {
            const int ex = (int)((lo[0].z >> 23) & 0xFFu);                       // biased exponent (0: a denormal prior -> serial path)
            const int emax = wave_max_i(v ? ex : 0), emin = -wave_max_i(v ? -ex : -255);
            if (emin == 0 || emax - emin > 22) serial = true;
            else seen = wave_sum_d(v ? (double)__uint_as_float(lo[0].z) : 0.0);
        }
    }
    if (serial) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            int i = c * 64 + lane;
            uint64_t vis = __ballot(i < k && (int)lo[c].x > 0);
            while (vis) { int b = __ffsll((unsigned long long)vis) - 1; seen += (double)rl(__uint_as_float(lo[c].z), b); vis &= vis - 1; }
        }
    }
    const float seen_f = (float)seen;
    const float fpu = (float)((double)cn.v - ((double)ev.fpu_reduction * sqrt((double)seen_f)));   // :92
    // :94 (float)sqrt((double)n): a correctly rounded f32 sqrt of the (exactly representable) count gives the same
    // float -- rounding a 53-bit sqrt to 24 bits is innocuous double rounding (53 >= 2*24 + 2)
    const float sqn = cn.n < (1 << 24) ? sqrtf((float)cn.n) : (float)sqrt((double)cn.n);
    const float cpuct = ev.cpuct;
    float best = -INFINITY; int bi = 0;
    sel = cn;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        int i = c * 64 + lane;
        int ni = (int)lo[c].x;
        float t = ni == 0 ? fpu : __uint_as_float(lo[c].y);
        float u = t + (((cpuct * __uint_as_float(lo[c].z)) * sqn) / ((float)(1 + ni)));               // :87
        if (i >= k) u = -INFINITY;
        constexpr int RW = G::MAXK <= 8 ? 8 : G::MAXK <= 16 ? 16 : G::MAXK <= 32 ? 32 : 64;
        float m = RW == 64 ? wave_max(u) : wave_max_n<RW>(u);
        if (m > best) {                                                      // strict '>' : first max wins (:100)
            uint64_t bal = __ballot(i < k && u == m);
            int b = __ffsll((unsigned long long)bal) - 1;
            best = m; bi = c * 64 + b;
            uint4 slo, shi;
            slo.x = rl(lo[c].x, b); slo.y = rl(lo[c].y, b); slo.z = rl(lo[c].z, b); slo.w = rl(lo[c].w, b);
            shi.x = rl(hi[c].x, b); shi.y = rl(hi[c].y, b); shi.z = rl(hi[c].z, b); shi.w = 0;
            unpack(slo, shi, sel);
        }
    }
    return bi;
}

// One wavefront runs find_leaf (:208-228) for one tree whose header is already in registers; `sink(st, lane)` receives the leaf
// state (it writes the observation wherever the network reads it: the dense batch in HBM for k_select, straight into the tower's
// LDS image for the fused search kernel).  `gate(node)` is called before the child block of `node` is read (the two-wave launch
// waits there for the priors the other wave is still writing); it returns true if the slot's tape counter must be re-read.
// Dependent-load chain: header (the root is in it) -> one child block per level.
struct NoGate { AZG_DEV bool operator()(int) const { return false; } };

// find_leaf by TWO wavefronts (launches that have one to spare): the walk needs only the node records -- which child to enter is
// decided by PUCT on the children's statistics -- while the game rules (play_action at every level, then win_state / valid_moves
// / observation of the leaf) need only the sequence of chosen actions.  The walker publishes every action it takes into a mailbox in
// LDS and goes on to the next level at once; the rules wavefront follows one level behind, and hands the leaf's valid-move list
// (left in act_lds), win state and player to move back for add_children.  Words grow monotonically over the simulations of a launch
// (gen = simulation number), so nothing is ever reset: cnt = gen * 1024 + actions published, fin = gen * 1024 + 512 * expand + depth,
// res = gen once k / e / player are valid.  Same arithmetic, same results as the one-wave form.
struct WalkMail { int cnt, fin, res, k, e, player, pad0, pad1; int act[128]; };
AZG_DEV int mail_load(const int *w) { return __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
AZG_DEV void mail_store(int *w, int v, int lane) { if (lane == 0) __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <class G, class Sink>
AZG_DEV void follow_tree(const View &ev, int slot, typename G::S st, WalkMail *mb, int gen, int lane, int *act_lds, Sink &&sink) {
    constexpr int NCH = (G::MAXK + 63) / 64;
    const int base = gen * 1024;
    int d = 0, fin = -1;
    for (int spin = 0; spin < (1 << 23); spin++) {
        const int f = mail_load(&mb->fin);                                   // (fin before cnt: a final depth implies its actions are published)
        const int c = mail_load(&mb->cnt);
        const int avail = c >= base ? c - base : 0;
        while (d < avail) { G::play(st, mb->act[d]); d++; }                  // MCTS.pyx:216
        if (f >= base) { fin = f - base; if (d == (fin & 511)) break; }
        if (d == avail) __builtin_amdgcn_s_sleep(1);
    }
    if (fin < 0) { if (lane == 0) raise_error(ev, AZG_E_INTERNAL); return; }
    if (fin & 512) {                                                         // :223-226 the leaf is new: its win state and moves
        const int e = G::win_bits(st);
        int my_a[NCH];
        const int k = G::valid_list(st, lane, act_lds, my_a);
#pragma unroll
        for (int c = 0; c < NCH; c++) if (c * 64 + lane < k) act_lds[c * 64 + lane] = my_a[c];   // (the hand-over: child i's action at act_lds[i])
        wave_sync();
        if (lane == 0) { mb->k = k; mb->e = e; mb->player = st.player; }
        mail_store(&mb->res, gen, lane);
    }
    G::store(st, &ev.leaf_states[slot], lane);
    sink(st, lane);
}
template <class G, class Sink, class Gate, class Ranks>
AZG_DEV void select_tree(const View &ev, int slot, int tree, const HdrR &hr, typename G::S st, uint64_t ctr, int lane, int *act_lds,
                         Sink &&sink, Gate &&gate, Ranks &&ranks, WalkMail *mb = nullptr, int gen = 0) {
    constexpr int NCH = (G::MAXK + 63) / 64;
    const bool split = mb != nullptr;                                        // (follow_tree runs the rules: see WalkMail)
    TreeHdr *h = ev.hdr + tree;
    Node *nodes = tree_nodes(ev, tree, hr.base);
    PathEnt *path = ev.path + (size_t)tree * ev.maxd;
    int cur = LEAF_IS_ROOT;
    NodeR cn = hr.root;
    int depth = 0;
    AZG_TSTAMP(ev, slot, lane, 4);
#ifdef AZG_TREE_TIMING
    unsigned long long lv_lat = 0, lv_cmp = 0, lv_pub = 0, lv_t0 = 0;
#endif
    while (cn.n > 0 && cn.e == 0 && depth < ev.maxd) {                       // MCTS.pyx:213
        const int k = cn.nchild, fc = cn.fc;
#ifdef AZG_TREE_TIMING
        lv_t0 = __builtin_amdgcn_s_memtime();
#endif
        if (k == 0 || fc < 0) { if (lane == 0) raise_error(ev, AZG_E_TREE_FULL); break; }
        if (!AZG_BLOCK_OK(ev, 2, fc, k, hr.alloc) || !AZG_BOUNDS_OK(ev, 3, (hr.base == 0 || hr.base == ev.cap) && depth < ev.maxd)) break;
        if (gate(cur)) ctr = ev.tape_ctr[slot];
        NodeR sel; int bi;
        if constexpr (NCH > 1) { bi = k > 64 ? best_child<G, NCH>(ev, nodes, fc, k, cn, lane, sel) : best_child<G, 1>(ev, nodes, fc, k, cn, lane, sel); }
        else bi = best_child<G, 1>(ev, nodes, fc, k, cn, lane, sel);
        cur = __builtin_amdgcn_readfirstlane(fc + bi);
        if (lane == 0) {                                                     // the path entry carries the child's (n, q) for the backup
            uint4 ent = make_uint4((uint32_t)cur | ((uint32_t)cn.player << 28), (uint32_t)sel.n, __float_as_uint(sel.q), 0u);
            *reinterpret_cast<uint4 *>(path + depth) = ent;
        }
        cn = sel;
        AZG_TSTAMP(ev, slot, lane, 11);
        if (split) {
            if (lane == 0) mb->act[depth] = cn.a;
            mail_store(&mb->cnt, gen * 1024 + depth + 1, lane);
        } else G::play(st, cn.a);                                            // :216
        AZG_TSTAMP(ev, slot, lane, 12);
#ifdef AZG_TREE_TIMING
        if (ev.dbg && lane == 0) {
            volatile unsigned long long *d = ev.dbg + (size_t)slot * 16;
            const unsigned long long t10 = d[10], t11 = d[11], t12 = d[12];
            lv_lat += t10 - lv_t0; lv_cmp += t11 - t10; lv_pub += t12 - t11;
        }
#endif
        depth++;
    }
#ifdef AZG_TREE_TIMING
    if (ev.dbg && lane == 0) { ev.dbg[(size_t)slot * 16 + 10] = lv_lat; ev.dbg[(size_t)slot * 16 + 11] = lv_cmp; ev.dbg[(size_t)slot * 16 + 12] = lv_pub | ((unsigned long long)depth << 48); }
#endif
    if (split) mail_store(&mb->fin, gen * 1024 + (cn.n == 0 ? 512 : 0) + depth, lane);
    int expanded = 0;
    int alloc = hr.alloc;
    AZG_TSTAMP(ev, slot, lane, 5);
#ifdef AZG_TREE_TIMING
    if (ev.dbg && lane == 0) ev.dbg[(size_t)slot * 16 + 15] = (unsigned long long)depth;
#endif
    if (cn.n == 0) {                                                         // :223-226 expand
        int e, k, player = st.player;
        int my_a[NCH];
        if (split) {                                                         // (the rules wavefront left the list in act_lds)
            for (int spin = 0; mail_load(&mb->res) < gen; spin++) {
                if (spin > (1 << 23)) { if (lane == 0) raise_error(ev, AZG_E_INTERNAL); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            k = mb->k; e = mb->e; player = mb->player;
#pragma unroll
            for (int c = 0; c < NCH; c++) my_a[c] = c * 64 + lane < k ? act_lds[c * 64 + lane] : -1;
        } else {
            e = G::win_bits(st);
            k = G::valid_list(st, lane, act_lds, my_a);
        }
        AZG_TSTAMP(ev, slot, lane, 13);
        const int fc = add_children_any<G>(ev, slot, nodes, alloc, k, my_a, ctr, lane, ranks);
        AZG_TSTAMP(ev, slot, lane, 14);
        cn.fc = fc; cn.nchild = fc < 0 ? 0 : k; cn.player = player; cn.e = e;
        if (lane == 0) {
            ev.tape_ctr[slot] = ctr;
            const uint4 hi = pack_hi(cn.fc, cn.a, cn.nchild, cn.player, cn.e);
            if (cur == LEAF_IS_ROOT) reinterpret_cast<uint4 *>(h)[1] = hi;
            else reinterpret_cast<uint4 *>(nodes + cur)[1] = hi;
        }
        expanded = 1;
    }
    if (lane == 0) {
        const int md = depth > hr.max_depth ? depth : hr.max_depth;         // :219-221
        uint4 *q = reinterpret_cast<uint4 *>(h);
        q[2] = make_uint4((unsigned)hr.base, (unsigned)alloc, (unsigned)depth, (unsigned)md);
        q[3] = make_uint4((unsigned)cur, (unsigned)cn.fc, (unsigned)cn.nchild | ((unsigned)cn.e << 16) | ((unsigned)cn.player << 24), (unsigned)expanded);
        ev.slot_exp[slot] += expanded;
    }
    AZG_TSTAMP(ev, slot, lane, 6);
    if (!split) {
        G::store(st, &ev.leaf_states[slot], lane);
        sink(st, lane);
    }
    AZG_TSTAMP(ev, slot, lane, 7);
}

AZG_DEV int tree_of_slot(const View &ev, int slot) {                         // (self-play: the header load does not wait for the state)
    int tree = slot;
    if (ev.arena) tree = slot * ev.T + __builtin_amdgcn_readfirstlane(ev.states[slot].player);
    return tree;
}

template <class G, class Sink>
AZG_DEV void select_slot(const View &ev, int slot, int lane, int *act_lds, Sink &&sink) {
    const int tree = tree_of_slot(ev, slot);
    HdrR hr; load_hdr(ev.hdr + tree, hr);
    const uint64_t ctr = ev.tape_ctr[slot];
    select_tree<G>(ev, slot, tree, hr, G::load(&ev.states[slot], lane), ctr, lane, act_lds, sink, NoGate{}, NoRanks{});
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
// np.sum of the masked policy (:245,252) in numpy's order: cp[c] = value of child c*64+lane, ca[c] its action.
template <class G, int NC>
AZG_DEV float masked_sum(const View &ev, const int (&ca)[NC], const float (&cp)[NC], float *m_lds, float *scr, int lane, bool first) {
    constexpr int A = G::A;
    if constexpr (A < 8) {
        float s = 0.f;                                                       // n < 8: sequential in action order
        for (int a = 0; a < A; a++) {
            uint64_t bal = __ballot(ca[0] == a);
            if (bal) s += rl(cp[0], __ffsll((unsigned long long)bal) - 1);
        }
        return s;
    } else {
        if (first) { for (int a = lane; a < A; a += 64) m_lds[a] = 0.f; }
        wave_sync();
#pragma unroll
        for (int c = 0; c < NC; c++) if (ca[c] >= 0) m_lds[ca[c]] = cp[c];
        wave_sync();
        return np_sum_static<A>(m_lds, scr, lane);
    }
}

// update_policy (:81-84) of the freshly expanded leaf from the network's policy row: mask + renormalise (:239-245), root
// temperature (:249-252) and Dirichlet noise (:197-206) at the root.
template <class G, int NC>
AZG_DEV void leaf_policy(const View &ev, int slot, Node *nodes, int fc, int k, bool at_root, const float *pi, float *m_lds, float *scr, int lane) {
    int ca[NC]; float cp[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        int i = c * 64 + lane;
        ca[c] = i < k ? (int)nodes[fc + i].a : -1;
        cp[c] = i < k ? pi[ca[c]] : 0.f;                                     // pi * valids: valid entries keep pi[a]
    }
    const float s = masked_sum<G, NC>(ev, ca, cp, m_lds, scr, lane, true);
    if (!(s > 0.f) && lane == 0) raise_error(ev, AZG_E_FLOATING_POINT);      // x / 0: the reference raises (np.seterr(all='raise'), :23)
#pragma unroll
    for (int c = 0; c < NC; c++) cp[c] = cp[c] / s;                          // :245
    if (at_root && ev.add_temp) {                                            // :249-252 root temperature
        const double ex = 1.0 / (double)ev.root_temp;
#pragma unroll
        for (int c = 0; c < NC; c++) cp[c] = np_pow_f32(cp[c], ex);
        const float s2 = masked_sum<G, NC>(ev, ca, cp, m_lds, scr, lane, false);
#pragma unroll
        for (int c = 0; c < NC; c++) cp[c] = cp[c] / s2;
    }
    if (at_root && ev.add_noise) {                                           // :197-206 Dirichlet noise (tape)
        const uint64_t ctr = ev.tape_ctr[slot];
        float nzv[NC];
        if (ev.noise_off) {                                                  // a RECORDED np.random.dirichlet vector (azg_set_random_tape), float32 as :198-200 casts it
            const int off = ctr < (uint64_t)ev.perm_len ? ev.noise_off[(size_t)slot * ev.perm_len + ctr] : -1;
            if (off < 0 || off + k > ev.noise_len) { if (lane == 0) raise_error(ev, AZG_E_INVALID_ARG); return; }
#pragma unroll
            for (int c = 0; c < NC; c++) { const int i = c * 64 + lane; nzv[c] = i < k ? ev.noise_pool[off + i] : 0.f; }
        } else {
            const uint64_t key = tape_u64(ev.seed, ev.slot_base + (uint64_t)slot, ctr);
            const double alpha = 10.83 / (double)k;
            double g[NC]; double acc = 0.0;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                int i = c * 64 + lane;
                g[c] = 0.0;
                if (i < k) { SubStream ss = { key, (uint64_t)i, 0 }; g[c] = ss_gamma(ss, alpha); }
            }
#pragma unroll
            for (int c = 0; c < NC; c++) { int lim = min(64, k - c * 64); for (int j = 0; j < lim; j++) acc += rl(g[c], j); }
            const double inv = 1.0 / acc;
#pragma unroll
            for (int c = 0; c < NC; c++) nzv[c] = (float)(g[c] * inv);
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float nz = nzv[c];
            cp[c] = (float)(((double)cp[c] * (1.0 - (double)ev.noise_frac)) + (double)(ev.noise_frac * nz));
        }
        if (lane == 0) ev.tape_ctr[slot] = ctr + 1;
    }
#pragma unroll
    for (int c = 0; c < NC; c++) { int i = c * 64 + lane; if (i < k) nodes[fc + i].p = cp[c]; }      // update_policy :81-84
}

// process_results (:230-289) in two independent parts (they touch disjoint memory): the POLICY part writes the priors of the
// freshly expanded leaf's children, the PATH part the running means along the path and the root count.  The header's leaf
// record and the (n, q) snapshots in the path make each one level of loads.
template <class G>
AZG_DEV void backup_policy(const View &ev, int slot, const HdrR &hr, Node *nodes, const float *pi, float *m_lds, float *scr, int lane) {
    constexpr int NCH = (G::MAXK + 63) / 64;
    if (hr.leaf_e || hr.leaf_fc < 0) return;                                 // :234-235 terminal: no policy update
    const int k = hr.leaf_k, fc = hr.leaf_fc;
    if (!AZG_BLOCK_OK(ev, 4, fc, k, hr.alloc)) return;
    const bool at_root = hr.leaf == LEAF_IS_ROOT;
    if constexpr (NCH > 1) {
        if (k > 64) leaf_policy<G, NCH>(ev, slot, nodes, fc, k, at_root, pi, m_lds, scr, lane);
        else leaf_policy<G, 1>(ev, slot, nodes, fc, k, at_root, pi, m_lds, scr, lane);
    } else leaf_policy<G, 1>(ev, slot, nodes, fc, k, at_root, pi, m_lds, scr, lane);
}

// vrow: the value row (P + 1 probabilities); returns the header as it stands after the backup (root.n + 1)
template <class G>
AZG_DEV void backup_path(const View &ev, int slot, int tree, HdrR &hr, Node *nodes, const float (&vrow)[G::P + 1], int lane) {
    constexpr int P = G::P, NV = P + 1, NE = P + G::HAS_DRAW;
    TreeHdr *h = ev.hdr + tree;
    const PathEnt *path = ev.path + (size_t)tree * ev.maxd;
    const int depth = hr.depth;
    if (!AZG_BOUNDS_OK(ev, 5, depth >= 0 && depth <= ev.maxd && depth <= G::MAX_TURNS + 2)) return;
    uint4 ent[(G::MAX_TURNS + 2 + 63) / 64];                                 // the path, one level per lane
#pragma unroll
    for (int j0 = 0, c = 0; j0 < G::MAX_TURNS + 2; j0 += 64, c++)
        ent[c] = j0 + lane < depth ? *reinterpret_cast<const uint4 *>(path + j0 + lane) : make_uint4(0, 0, 0, 0);
    float val[NV > NE ? NV : NE];
    int vsize;
    if (hr.leaf_e) {                                                         // :234-235 terminal: value = float32(e)
#pragma unroll
        for (int j = 0; j < NE; j++) val[j] = (float)((hr.leaf_e >> j) & 1);
        vsize = NE;
    } else {
#pragma unroll
        for (int j = 0; j < NV; j++) val[j] = vrow[j];
        vsize = NV;
    }
    // ---- backup along the path (:265-287): X_j = path[j-1] node, mover = player of X_{j-1}; all levels independent
    const float draw_share = vsize > P ? (val[P] / ((float)P)) : 0.f;
#pragma unroll
    for (int j0 = 0, c = 0; j0 < G::MAX_TURNS + 2; j0 += 64, c++) {
        const int j = j0 + lane;
        if (j < depth) {
            const int idx = (int)(ent[c].x & 0x0FFFFFFFu), mover = (int)(ent[c].x >> 28);
            float vm = val[0];
#pragma unroll
            for (int pp = 1; pp < P; pp++) if (mover == pp) vm = val[pp];
            const float v = vsize > P ? vm + draw_share : vm;                // _get_value :291-295
            const int n = (int)ent[c].y; const float q = __uint_as_float(ent[c].z);
            if (!AZG_BOUNDS_OK(ev, 6, idx >= 0 && idx < hr.alloc && hr.alloc <= ev.cap && n >= 0)) continue;
            Node *x = nodes + idx;
            const float qn = (((q * (float)n) + (v * 1.0f)) / ((float)(n + 1)));      // :282 (discount == 1)
            *reinterpret_cast<uint2 *>(x) = make_uint2((unsigned)(n + 1), __float_as_uint(qn));
            if (n == 0) {                                                    // :283-284 (only the leaf can have n == 0)
                float vo = val[0];
#pragma unroll
                for (int pp = 1; pp < P; pp++) if (hr.leaf_player == pp) vo = val[pp];
                x->v = vsize > P ? vo + draw_share : vo;
            }
        }
    }
    hr.root.n += 1;
    if (lane == 0) { h->root.n = hr.root.n; ev.slot_sims[slot] += 1; }       // :289
}

// One wavefront runs all of process_results for one slot with its policy row pi[A] and value row vrow[P+1] (HBM or LDS).
// m_lds [max(A, 8)] and scr [64] are wave-private scratch (only used when A >= 8).
template <class G>
AZG_DEV void backup_slot(const View &ev, int slot, int lane, const float *pi, const float *vrow, float *m_lds, float *scr) {
    const int tree = tree_of_slot(ev, slot);
    HdrR hr; load_hdr(ev.hdr + tree, hr);
    Node *nodes = tree_nodes(ev, tree, hr.base);
    float val[G::P + 1];
#pragma unroll
    for (int j = 0; j < G::P + 1; j++) val[j] = hr.leaf_e ? 0.f : vrow[j];
    AZG_TSTAMP(ev, slot, lane, 1);
    backup_policy<G>(ev, slot, hr, nodes, pi, m_lds, scr, lane);
    AZG_TSTAMP(ev, slot, lane, 2);
    backup_path<G>(ev, slot, tree, hr, nodes, val, lane);
    AZG_TSTAMP(ev, slot, lane, 3);
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

// hand-off flags between the two wavefronts of a slot (LDS, workgroup-scope release / acquire).
// Generation flags: the producer stores `gen`, the consumer waits for a value >= gen.  Every wait is BOUNDED (~10^7 polls, seconds):
// a hand-off that never comes -- it cannot, both wavefronts run the same uniform control flow -- must not hang the GPU; it raises
// the sticky AZG_E_INTERNAL instead.
AZG_DEV void flag_set_gen(int *f, int gen, int lane) { if (lane == 0) __hip_atomic_store(f, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
AZG_DEV void flag_wait_gen(const View &ev, int *f, int gen) {
    for (int spin = 0; spin < (1 << 23); spin++) {
        if (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= gen) return;
        __builtin_amdgcn_s_sleep(2);
    }
    raise_error(ev, AZG_E_INTERNAL);
}

// ================================================================================================ sparse heads
// A factorised head (NNetArchitecture.py:88-102: 1x1 conv -> BN -> flatten -> Linear chain, all linear) ends in one dot product per
// output over the board's head features.  process_results only ever uses the logits of the leaf's VALID actions -- the policy is
// masked and renormalised (MCTS.pyx:239-245) -- so the tree launch computes exactly those, k <= MAXK rows of the collapsed matrix
// instead of all A (brandubh: ~40 of 588, 64 KB of weights per leaf instead of 0.93 MB), and leaves -inf everywhere else: the
// softmax over the row then IS the softmax over the valid actions.  Against softmax-over-all-A followed by mask + renormalise
// this differs only in rounding (the full normaliser cancels): priors agree to ~1e-7 relative, far inside the fp16 network's
// own error; if every valid action underflowed in the full softmax the reference would leave zeros where this path still
// normalises -- logit gaps > 87, never seen.
//   rows: fp16 [A + P + 1][fk] row-major (o < A policy output o over the policy features, then the value outputs over the value
//   features), bias f32 [A + P + 1]; a board's features: fp16 [2][fk] = policy half, value half, index pixel * 16 + channel.
struct HeadRows { const _Float16 *rows; const float *bias; int fk; };
template <class G> constexpr int head_fk() { return (G::CELLS * 16 + 31) / 32 * 32; }
typedef _Float16 hrow8 __attribute__((ext_vector_type(8)));
typedef _Float16 hrow2 __attribute__((ext_vector_type(2)));

// One output per DPP row of 16 lanes: lane j16 owns the 16-byte chunks j16, j16 + 16, ... of the row.  fp32 accumulation with
// v_dot2_f32_f16 in chunk order, then a butterfly over the row: a fixed association, the same in every kernel that calls it.
template <int FK>
struct HeadDot {
    static constexpr int NCH = FK / 8, IT = (NCH + 15) / 16;
    static AZG_DEV void load(hrow8 (&w)[IT], const _Float16 *row, int j16, bool on) {
        const hrow8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < IT; t++) {                                      // (unconditional loads of clamped addresses + a select: a
            const int c = j16 + 16 * t;                                     //  load under a lane predicate becomes a branch, and the
            const hrow8 v = *reinterpret_cast<const hrow8 *>(row + min(c, NCH - 1) * 8);   // compiler then cannot count the loads in flight)
            w[t] = (on && c < NCH) ? v : zero;
        }
    }
    static AZG_DEV float dot(const hrow8 (&w)[IT], const _Float16 *feat, int j16) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < IT; t++) {
            const hrow8 f = *reinterpret_cast<const hrow8 *>(feat + min(j16 + 16 * t, NCH - 1) * 8);      // (past the end: w is zero)
#pragma unroll
            for (int h = 0; h < 4; h++)
                acc = __builtin_amdgcn_fdot2((hrow2){w[t][2 * h], w[t][2 * h + 1]}, (hrow2){f[2 * h], f[2 * h + 1]}, acc, false);
        }
        acc += dpp_f<DPP_QUAD_XOR1>(acc); acc += dpp_f<DPP_QUAD_XOR2>(acc);
        acc += dpp_f<DPP_ROW_HALF_MIRROR>(acc); acc += dpp_f<DPP_ROW_MIRROR>(acc);
        return acc;
    }
};

// policy logits of the last leaf's k children (one wavefront): lg[a] for their actions, -inf for every other a < A.
// The k dot products run on the MFMA pipe, which is idle while the trees are walked: 16 children per v_mfma_f32_16x16x32_f16 --
// A operand = 16 weight rows x 32 features (lane g * 16 + i gathers 16 bytes of child i's row), B operand = the board's features
// from LDS in all 16 columns, one accumulation chain over the FK / 32 k-steps per 16 children.  The rows of the NEXT 16 children are
// fetched while the current chain runs (one 16-byte load per lane and k-step, FK / 32 of them in flight: the job is L2 latency, 64 KB
// of rows per leaf), the biases travel with the rows.  Same association in every kernel that calls it.
typedef float hfloatx4 __attribute__((ext_vector_type(4)));
// TWO_BUFFERS = false (launches that must fit two waves per SIMD): one row buffer, the gather of a subtile is not hidden under the
// previous chain.  Same arithmetic either way.
template <class G, bool TWO_BUFFERS = true>
AZG_DEV void leaf_policy_logits(const HeadRows &hd, const Node *nodes, int fc, int k, const _Float16 *feat, float *lg, int lane) {
    constexpr int A = G::A, FK = head_fk<G>(), KSTEPS = FK / 32, NCH = (G::MAXK + 63) / 64;
    for (int a = lane; a < A; a += 64) lg[a] = -INFINITY;
    int acts[NCH];                                                           // child i's action in lane i & 63 of acts[i >> 6]
#pragma unroll
    for (int c = 0; c < NCH; c++) acts[c] = c * 64 + lane < k ? (int)nodes[fc + c * 64 + lane].a : 0;
    wave_sync();
    const int g = lane >> 4, i16 = lane & 15, nsub = (k + 15) >> 4;
    auto action_of = [&](int child) {                                        // (child < k; any lane may ask for any child)
        int a = 0;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int v = __builtin_amdgcn_ds_bpermute((child & 63) << 2, acts[c]);
            if ((child >> 6) == c) a = v;
        }
        return a;
    };
    // two row buffers: while the chain of 16 children runs out of one, the rows of the next 16 land in the other (the loads are
    // issued BEFORE the chain: the compiler keeps program order between loads and the MFMAs that read their registers)
    hrow8 wa[KSTEPS];
    [[maybe_unused]] hrow8 wb[TWO_BUFFERS ? KSTEPS : 1];
    float bias_a;
    [[maybe_unused]] float bias_b;
    auto fetch = [&](int sub, hrow8 (&w)[KSTEPS], float &bias) {
        const int an = action_of(min(16 * sub + i16, k - 1));
        const _Float16 *rn = hd.rows + (size_t)an * hd.fk + g * 8;
        bias = hd.bias[an];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) w[ks] = *reinterpret_cast<const hrow8 *>(rn + ks * 32);
    };
    const _Float16 *fb = feat + g * 8;
    auto chain = [&](int sub, const hrow8 (&w)[KSTEPS], float bias) {
        hfloatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};       // even / odd k-steps: two independent chains (a
#pragma unroll                                                               //  dependent MFMA waits ~2x its issue interval), summed at the end
        for (int ks = 0; ks < KSTEPS; ks++) {
            const hrow8 b = *reinterpret_cast<const hrow8 *>(fb + ks * 32);
            if (ks & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks], b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks], b, acc0, 0, 0, 0);
        }
        const hfloatx4 acc = acc0 + acc1;
        // D[row 4 g + r][column n] sits in lane g * 16 + n, element r; every column is the same dot product, so lane (g, n < 4)
        // delivers child 16 sub + 4 g + n
        const int child = 16 * sub + 4 * g + (i16 & 3);
        const float x = (i16 & 3) == 0 ? acc[0] : (i16 & 3) == 1 ? acc[1] : (i16 & 3) == 2 ? acc[2] : acc[3];
        const int aw = action_of(min(child, k - 1));
        const float bw = __int_as_float(__builtin_amdgcn_ds_bpermute((4 * g + (i16 & 3)) << 2, __float_as_int(bias)));
        if (i16 < 4 && child < k) lg[aw] = x + bw;
    };
    if constexpr (!TWO_BUFFERS) {
        for (int s = 0; s < nsub; s++) {
            fetch(s, wa, bias_a);
            __builtin_amdgcn_sched_barrier(0);
            chain(s, wa, bias_a);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
    fetch(0, wa, bias_a);
    for (int s = 0; s < nsub; s += 2) {                                      // (sched_barrier: left alone the scheduler sinks the loads
        if (s + 1 < nsub) fetch(s + 1, wb, bias_b);                          //  of the next subtile below the chain that should hide them;
        __builtin_amdgcn_sched_barrier(0);                                   //  the branches are wave-uniform)
        chain(s, wa, bias_a);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 >= nsub) break;
        if (s + 2 < nsub) fetch(s + 2, wa, bias_a);
        __builtin_amdgcn_sched_barrier(0);
        chain(s + 1, wb, bias_b);
        __builtin_amdgcn_sched_barrier(0);
    }
    }
}
// the P + 1 value logits of a board into lg[0 .. NV) (one wavefront, one pass).  feat: the board's value features.
template <class G>
AZG_DEV void leaf_value_logits(const HeadRows &hd, const _Float16 *feat, float *lg, int lane) {
    constexpr int A = G::A, NV = G::P + 1, FK = head_fk<G>();
    static_assert(NV <= 4, "one DPP row per value output");
    using HD = HeadDot<FK>;
    const int r = lane >> 4, j16 = lane & 15;
    hrow8 w[HD::IT];
    HD::load(w, hd.rows + (size_t)(A + min(r, NV - 1)) * hd.fk, j16, r < NV);
    const float x = HD::dot(w, feat, j16);
    if (j16 == 0 && r < NV) lg[r] = x + hd.bias[A + r];
}

// backup of simulation k and find_leaf of simulation k + 1 of the same slot in one launch (they are consecutive in the lock-step
// loop, SelfPlayAgent.pyx:87-92, and touch the same tree), by TWO wavefronts per slot.  Wave 0 walks the tree: path update of
// simulation k, then the descent and expansion of simulation k + 1.  Wave 1 prepares what the walk will need: (1) the priors of
// the previous leaf's children from the network's row (softmax when handed logits, mask, renormalise, root temperature / noise),
// (2) the shuffle of the next expansion -- the all-pairs comparison of the next 64 tape keys, which does not depend on which
// leaf gets expanded.  The walk waits for (1) only if the descent enters the previous leaf, and for (2) when it expands.
// Same arithmetic as the one-wave functions, same results.
//   IN_LOGITS: `policy` holds rows of stride `ld` with A policy logits then P + 1 value logits (as azg_policy_value_heads_f16
//   leaves them) instead of probabilities -- one launch and one HBM round trip of the probabilities less per simulation.
//   IN_FEATURES: `policy` holds the boards' head features (fp16 [2][hd.fk] per row, as azg_resnet_tower_features_f16 leaves them)
//   and the launch computes the logits it needs itself (sparse heads above) -- no heads launch at all.
enum { IN_PROBS = 0, IN_LOGITS = 1, IN_FEATURES = 2 };
template <class G, typename OT, bool NHWC8, int MODE>
__global__ __launch_bounds__(128) void k_backup_select2(View ev, const float *policy, const float *value, int ld, OT *obs,
                                                        const int32_t *row_of_slot, int do_select, HeadRows hd) {
    constexpr int A = G::A, NV = G::P + 1;
    constexpr bool LOGITS = MODE != IN_PROBS;
    __shared__ float m_lds[A < 8 ? 8 : A];
    __shared__ float scr[64];
    __shared__ int act_lds[((G::MAXK + 63) / 64) * 64];
    __shared__ float pi_lds[LOGITS ? A : 1];
    __shared__ float lg_lds[MODE == IN_FEATURES ? A + 4 : 1];                // (features: the logits this launch computes itself)
    __shared__ __attribute__((aligned(16))) _Float16 feat_lds[MODE == IN_FEATURES ? head_fk<G>() : 8];   // the board's policy features
    __shared__ unsigned long long less_lds[64];
    __shared__ int flags[3];                                                 // 0: priors written, 1: shuffle masks ready, 2: sticky error seen
    const int slot = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    AZG_TSTAMP(ev, slot, threadIdx.x, 8);
    if (threadIdx.x < 2) flags[threadIdx.x] = 0;
    if (threadIdx.x == 2) flags[2] = ev.gcount[GC_ERROR];                   // ONE read decides for both wavefronts (a wave that left
                                                                             //  alone would leave its partner waiting for a flag)
    const int row = row_of_slot ? row_of_slot[slot] : slot;
    const int tree = tree_of_slot(ev, slot);
    HdrR hr; load_hdr(ev.hdr + tree, hr);
    const uint64_t ctr0 = ev.tape_ctr[slot];
    Node *nodes = tree_nodes(ev, tree, hr.base);
    const bool has_policy = !hr.leaf_e && hr.leaf_fc >= 0;
    const bool root_noise = has_policy && hr.leaf == LEAF_IS_ROOT && ev.add_noise;     // the backup draws one tape number
    __syncthreads();
    if (flags[2] != 0) return;                                               // sticky device error: stop touching the trees
    if (wave == 1) {                                                         // ---- what the walk will need
        // (computing the logits takes longer than a typical descent: in that mode the shuffle masks, which every expansion waits
        //  for, go first, and the priors, which only a descent through the previous leaf waits for, second)
        constexpr bool MASKS_FIRST = MODE == IN_FEATURES;
        if (MASKS_FIRST && do_select) {
            less_lds[lane] = shuffle_less_mask<(G::MAXK < 64 ? G::MAXK : 64)>(ev, slot, ctr0 + (root_noise ? 1 : 0), lane);
            flag_set_gen(&flags[1], 1, lane);
        }
        if (has_policy) {
            const float *pi = policy + (size_t)row * ld;
            if constexpr (MODE == IN_FEATURES) {
                const uint4 *feat = reinterpret_cast<const uint4 *>(reinterpret_cast<const _Float16 *>(policy) + (size_t)row * 2 * hd.fk);
                for (int c = lane; c < head_fk<G>() / 8; c += 64) reinterpret_cast<uint4 *>(feat_lds)[c] = feat[c];
                wave_sync();
                leaf_policy_logits<G>(hd, nodes, hr.leaf_fc, hr.leaf_k, feat_lds, lg_lds, lane);
                wave_sync();
                pi = lg_lds;
            }
            if constexpr (LOGITS) { policy_softmax_row<A>(pi, lane, A, pi_lds); wave_sync(); pi = pi_lds; }
            backup_policy<G>(ev, slot, hr, nodes, pi, m_lds, scr, lane);
        }
        flag_set_gen(&flags[0], 1, lane);
        AZG_TSTAMP(ev, slot, lane, 9);
        if (!MASKS_FIRST && do_select) {
            less_lds[lane] = shuffle_less_mask<(G::MAXK < 64 ? G::MAXK : 64)>(ev, slot, ctr0 + (root_noise ? 1 : 0), lane);
            flag_set_gen(&flags[1], 1, lane);
        }
        AZG_TSTAMP(ev, slot, lane, 0);
        return;
    }
    typename G::S st = G::load(&ev.states[slot], lane);                      // ---- the walk
    float val[NV];
    if constexpr (MODE == IN_FEATURES) {
        float pv = 0.f;
        if (!hr.leaf_e) {                                                    // (a terminal leaf backs its win state up, not the network)
            leaf_value_logits<G>(hd, reinterpret_cast<const _Float16 *>(policy) + (size_t)row * 2 * hd.fk + hd.fk, lg_lds + A, lane);
            wave_sync();
            pv = value_softmax(lg_lds + A, lane, NV);
        }
#pragma unroll
        for (int j = 0; j < NV; j++) val[j] = rl(pv, j);
    } else if constexpr (LOGITS) {
        const float pv = value_softmax(policy + (size_t)row * ld + A, lane, NV);
#pragma unroll
        for (int j = 0; j < NV; j++) val[j] = rl(pv, j);
    } else {
#pragma unroll
        for (int j = 0; j < NV; j++) val[j] = value[(size_t)row * NV + j];
    }
    const int prev_leaf = hr.leaf;
    AZG_TSTAMP(ev, slot, lane, 1);
    backup_path<G>(ev, slot, tree, hr, nodes, val, lane);
    AZG_TSTAMP(ev, slot, lane, 3);
    if (!do_select) return;
    wave_sync();                                                             // the path stores land before the descent re-reads those nodes
    bool waited = false;
    select_tree<G>(ev, slot, tree, hr, st, ctr0, lane, act_lds, [&](const typename G::S &ls, int ln) {
        if (obs) {
            if constexpr (NHWC8) G::write_obs_nhwc8(ls, (_Float16 *)obs + (size_t)row * G::CELLS * 8, ln);
            else G::template write_obs<OT>(ls, obs + (size_t)row * G::OBS, ln);
        }
    }, [&](int node) {                                                       // before the child block of `node` is read
        if (waited || node != prev_leaf) return false;
        flag_wait_gen(ev, &flags[0], 1); waited = true;                      // the previous leaf's priors are complete from here on
        return root_noise;                                                   // (the tape counter moved)
    }, [&](int k, int ln, int &pos) {                                        // ranks of the k new children
        if (k > 64 || ev.perm_tape) return false;
        flag_wait_gen(ev, &flags[1], 1);
        pos = __popcll(less_lds[ln] & (k == 64 ? ~0ULL : ((1ULL << k) - 1ULL)));
        return true;
    });
}

// The sparse heads on their own (one wavefront per slot): the row the IN_FEATURES launch computes for a slot's last leaf -- policy
// logits of the leaf's children (-inf for every other action), then the P + 1 value logits -- written to logits[row][ld].  The
// same leaf_policy_logits / leaf_value_logits calls, so softmax + azg_backup_select_logits on this row is bit-identical to
// azg_backup_select_features; a leaf that takes no evaluation (terminal) gets a row of zeros.
template <class G>
__global__ __launch_bounds__(64) void k_leaf_heads_sparse(View ev, const _Float16 *feat, HeadRows hd, const int32_t *row_of_slot, float *logits, int ld) {
    constexpr int A = G::A, NV = G::P + 1;
    __shared__ float lg_lds[A + 4];
    __shared__ __attribute__((aligned(16))) _Float16 feat_lds[head_fk<G>()];
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int row = row_of_slot ? row_of_slot[slot] : slot;
    const int tree = tree_of_slot(ev, slot);
    HdrR hr; load_hdr(ev.hdr + tree, hr);
    float *out = logits + (size_t)row * ld;
    if (hr.leaf_e || hr.leaf_fc < 0) { for (int a = lane; a < A + NV; a += 64) out[a] = 0.f; return; }
    const _Float16 *frow = feat + (size_t)row * 2 * hd.fk;
    for (int c = lane; c < head_fk<G>() / 8; c += 64) reinterpret_cast<uint4 *>(feat_lds)[c] = reinterpret_cast<const uint4 *>(frow)[c];
    wave_sync();
    leaf_policy_logits<G>(hd, tree_nodes(ev, tree, hr.base), hr.leaf_fc, hr.leaf_k, feat_lds, lg_lds, lane);
    leaf_value_logits<G>(hd, frow + hd.fk, lg_lds + A, lane);
    wave_sync();
    for (int a = lane; a < A + NV; a += 64) out[a] = lg_lds[a];
}

// ================================================================================================ root stats
// MCTS.probs (:308-329) of a tree's root into LDS pr[A]; every lane returns.  cnt = LDS float counts.
template <class G>
AZG_DEV void root_probs(const View &ev, const Node *nodes, int root_fc, int root_k, float temp, float *cnt, float *pr, float *scr, int lane) {
    constexpr int A = G::A;
    for (int a = lane; a < A; a += 64) cnt[a] = 0.f;
    wave_sync();
    for (int i = lane; i < root_k; i += 64) cnt[nodes[root_fc + i].a] = (float)nodes[root_fc + i].n;    // counts :297-303
    wave_sync();
    if (temp == 0.f) {                                                       // :313-317 one-hot first argmax
        float best = cnt[0]; int b = 0;
        for (int a = 1; a < A; a++) if (cnt[a] > best) { best = cnt[a]; b = a; }
        for (int a = lane; a < A; a += 64) pr[a] = a == b ? 1.f : 0.f;
        wave_sync();
        return;
    }
    float s;
    if constexpr (A < 8) { s = 0.f; for (int a = 0; a < A; a++) s += cnt[a]; }
    else s = np_sum_static<A>(cnt, scr, lane);
    const double ex = 1.0 / (double)temp;
    for (int a = lane; a < A; a += 64) pr[a] = np_pow_f32(cnt[a] / s, ex);    // :320
    wave_sync();
    float s2;
    if constexpr (A < 8) { s2 = 0.f; for (int a = 0; a < A; a++) s2 += pr[a]; }
    else s2 = np_sum_static<A>(pr, scr, lane);
    for (int a = lane; a < A; a += 64) pr[a] = pr[a] / s2;                    // :321
    wave_sync();
}

template <class G>
__global__ __launch_bounds__(64) void k_root_stats(View ev, int what, float temp, int average, int32_t *counts, float *probs, float *values) {
    constexpr int A = G::A;
    __shared__ float cnt[A < 8 ? 8 : A], pr[A < 8 ? 8 : A], scr[64];
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int mover0 = __builtin_amdgcn_readfirstlane(ev.states[slot].player);
    const int tree = ev.arena ? slot * ev.T + mover0 : slot;
    HdrR hr; load_hdr(ev.hdr + tree, hr);
    const Node *nodes = tree_nodes(ev, tree, hr.base);
    const int fc = hr.root.fc, k = hr.root.nchild;
    if (k > 0 && !AZG_BLOCK_OK(ev, 11, fc, k, hr.alloc)) return;
    if (what == 0) {
        for (int a = lane; a < A; a += 64) counts[(size_t)slot * A + a] = 0;
        wave_sync();
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

// MCTS.update_root (:185-195) for one tree: the chosen child's record is copied into the header (it becomes the root; its own
// slot in the parent's child block is garbage from then on); returns false if the action is not a child.
template <class G>
AZG_DEV bool update_root(const View &ev, int slot, int tree, const typename G::S &st, int action, int *act_lds, int lane) {
    constexpr int NCH = (G::MAXK + 63) / 64;
    TreeHdr *h = ev.hdr + tree;
    HdrR hr; load_hdr(h, hr);
    Node *nodes = tree_nodes(ev, tree, hr.base);
    int fc = hr.root.fc, k = hr.root.nchild;
    if (k == 0) {                                                            // :186-187 unexpanded root: add + shuffle
        int my_a[NCH];
        k = G::valid_list(st, lane, act_lds, my_a);
        uint64_t ctr = ev.tape_ctr[slot];
        int alloc = hr.alloc;
        fc = add_children_any<G>(ev, slot, nodes, alloc, k, my_a, ctr, lane, NoRanks{});
        if (fc < 0) return false;
        if (lane == 0) { ev.tape_ctr[slot] = ctr; h->alloc = alloc; }
        wave_sync();
    }
    int found = -1;
    if (!AZG_BLOCK_OK(ev, 7, fc, k, __builtin_amdgcn_readfirstlane(h->alloc))) return false;
    for (int i0 = 0; i0 < k; i0 += 64) {
        int i = i0 + lane;
        uint64_t bal = __ballot(i < k && (int)nodes[fc + i].a == action);
        if (bal) { found = i0 + __ffsll((unsigned long long)bal) - 1; break; }
    }
    if (found < 0) return false;
    uint4 lo, hi; load_node(nodes + fc + found, lo, hi);
    if (lane == 0) { uint4 *q = reinterpret_cast<uint4 *>(h); q[0] = lo; q[1] = hi; }
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

// Node reclamation.  The reference drops the siblings of the played move with Python's GC (MCTS.update_root :185-195 rebinds
// _root).  Here a tree's node store is two semi-spaces of `cap` nodes; when, after a move, fewer than `compact_reserve` free
// nodes are left in the live space, the subtree under the root (which sits in the header) is copied breadth-first into the
// other space -- child blocks stay contiguous and keep their list order, so no result changes -- and the spaces swap roles.
// One wavefront per tree; 64 nodes of the copy frontier per step (their child blocks are sized with a wave scan).
template <class G>
__global__ __launch_bounds__(64) void k_compact(View ev, int force, int first_tree) {
    if (__builtin_amdgcn_readfirstlane(ev.gcount[GC_ERROR]) != 0) return;
    const int tree = first_tree + blockIdx.x, lane = threadIdx.x;      // (grid = the trees to look at: all of them, or one slot's)
    TreeHdr *h = ev.hdr + tree;
    HdrR hr; load_hdr(h, hr);
    if (!force && hr.alloc + ev.compact_reserve <= ev.cap) return;
    const Node *from = tree_nodes(ev, tree, hr.base);
    const int nbase = hr.base ? 0 : ev.cap;
    Node *to = tree_nodes(ev, tree, nbase);
    int nalloc = 0;
    if (hr.root.fc >= 0 && hr.root.nchild > 0) {                             // the root's child block -> to[0 .. k)
        const int k = hr.root.nchild;
        if (!AZG_BLOCK_OK(ev, 8, hr.root.fc, k, hr.alloc)) return;
        for (int i = lane; i < k; i += 64) { uint4 lo, hi; load_node(from + hr.root.fc + i, lo, hi); uint4 *q = reinterpret_cast<uint4 *>(to + i); q[0] = lo; q[1] = hi; }
        nalloc = k;
    }
    wave_sync();
    for (int scan = 0; scan < nalloc;) {                                     // nodes to[scan .. nalloc) still carry from-space child pointers
        const int end = min(scan + 64, nalloc), i = scan + lane;
        scan = end;
        int ofc = -1, kk = 0;
        if (i < end) { const uint4 hi = reinterpret_cast<const uint4 *>(to + i)[1]; ofc = (int)hi.x; kk = ofc >= 0 ? (int)(hi.y >> 16) : 0; }
        const int off = wave_excl_scan(kk, lane), total = wave_sum_i(kk);
        const int nfc = nalloc + off;
        if (kk > 0) reinterpret_cast<int32_t *>(to + i)[4] = nfc;            // first_child now points into to-space
        uint64_t todo = __ballot(kk > 0);
        while (todo) {                                                       // copy the frontier's child blocks, one parent at a time
            const int b = __ffsll((unsigned long long)todo) - 1; todo &= todo - 1;
            const int sfc = rl(ofc, b), sk = rl(kk, b), dfc = rl(nfc, b);
            if (!AZG_BLOCK_OK(ev, 9, sfc, sk, hr.alloc) || !AZG_BOUNDS_OK(ev, 10, dfc >= 0 && dfc + sk <= ev.cap)) continue;
            for (int j = lane; j < sk; j += 64) { uint4 lo, hi; load_node(from + sfc + j, lo, hi); uint4 *q = reinterpret_cast<uint4 *>(to + dfc + j); q[0] = lo; q[1] = hi; }
        }
        nalloc += total;
        wave_sync();
    }
    if (lane == 0) {
        if (hr.root.fc >= 0 && hr.root.nchild > 0) h->root.first_child = 0;
        h->base = nbase; h->alloc = nalloc;
        atomicMax(&ev.gcount[GC_MAXLIVE], nalloc);                           // (rare: a few trees per move; how full the kept subtrees get)
        h->leaf = LEAF_IS_ROOT; h->leaf_fc = -1; h->depth = 0;               // the last find_leaf's indices are void now
    }
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
    HdrR hr; load_hdr(ev.hdr + tree, hr);
    const Node *nodes = tree_nodes(ev, tree, hr.base);
    const int fc = hr.root.fc, k = hr.root.nchild;
    if (k > 0 && !AZG_BLOCK_OK(ev, 11, fc, k, hr.alloc)) return;
    float temp;
    if (ev.arena) temp = ev.arena_temp;                                      // :158
    else { int t = st.turns < ev.temp_len ? st.turns : ev.temp_len - 1; temp = ev.temp_table[t]; }   // :156-157
    root_probs<G>(ev, nodes, fc, k, temp, cnt, pr, scr, lane);               // :159
    {                                                                        // no visited child: counts / 0 (:320) -- the reference
        bool nan = false;                                                    // raises (np.seterr(all='raise'), :23); so does this
        for (int a = lane; a < A; a += 64) nan |= pr[a] != pr[a];
        if (__ballot(nan)) { if (lane == 0) { raise_error(ev, AZG_E_FLOATING_POINT); ev.fin_flag[slot] = 0; } return; }
    }
    // np.random.choice(A, p=policy) via the tape (:160): cdf in double, first index whose cdf/total > u
    uint64_t ctr = ev.tape_ctr[slot];
    double u;
    if (ev.u_tape) {                                                         // the uniform np.random.choice DREW, recorded (azg_set_random_tape)
        if (ctr >= (uint64_t)ev.perm_len) { if (lane == 0) { raise_error(ev, AZG_E_INVALID_ARG); ev.fin_flag[slot] = 0; } return; }
        u = ev.u_tape[(size_t)slot * ev.perm_len + ctr];
    } else u = u53(tape_u64(ev.seed, ev.slot_base + (uint64_t)slot, ctr));
    double total = 0.0;
    for (int a = 0; a < A; a++) { float x = pr[a]; if (x != 0.f) total += (double)x; }
    double acc = 0.0; int action = 0; bool le = true;                        // cdf_i <= u  (searchsorted side='right')
    for (int a = 0; a < A; a++) {
        float x = pr[a];
        if (x != 0.f) { acc += (double)x; le = (acc / total) <= u; }
        if (le) action = a + 1;
    }
    if (action >= A) action = A - 1;
    wave_sync();
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
    wave_sync();
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

// high-water mark of the tree arenas (live semi-space), computed when the counters are read
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
