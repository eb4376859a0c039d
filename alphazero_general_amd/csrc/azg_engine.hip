// azg_engine.hip -- host side of libazg_hip.so: the C ABI of include/azg.h over the kernels of azg_kernels.h.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC  (alphazero_general_amd/build.py)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "azg_kernels.h"
#include "azg_conv.h"

using namespace azg;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(AZG_E_HIP, std::string(#x) + ": " + hipGetErrorString(_e)); } while (0)

struct EvPair { hipEvent_t a, b; bool ext; };

// Timing of ONE kernel: while a profile hook is armed (g_kev), the next AZG_LAUNCH goes through hipExtLaunchKernelGGL, which stamps
// the pair's events with the dispatch's own begin / end timestamps -- the kernel's execution time as rocprofv3 reports it, without
// the ~2 us of marker packets that hipEventRecord brackets add to a 10 us kernel.
static thread_local EvPair *g_kev = nullptr;
#define AZG_LAUNCH(kernel, grid, block, lds, stream, ...) do { \
    if (g_kev) { hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, g_kev->a, g_kev->b, 0, __VA_ARGS__); g_kev = nullptr; } \
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); } while (0)

struct azg_engine {
    azg_config cfg;
    azg_game_info gi;
    View v;
    std::vector<void *> allocs;
    int32_t *d_p2i = nullptr, *d_ok = nullptr;
    int16_t *d_perm = nullptr;                                 // azg_set_shuffle_tape / azg_set_random_tape
    double *d_utape = nullptr; int32_t *d_noff = nullptr; float *d_npool = nullptr;
    bool profile = false;
    std::vector<EvPair> ev[3];
    double ms[3] = {0, 0, 0};
    int64_t launches[3] = {0, 0, 0};
    std::vector<EvPair> pool;
};

static const azg_game_info k_info[] = {
    // action_size, c,h,w, players, has_draw, max_turns, nsym, cells, max_children
    {C4::A, C4::OBS_C, C4::H, C4::W, C4::P, C4::HAS_DRAW, C4::MAX_TURNS, C4::NSYM, C4::CELLS, C4::MAXK},
    {BR::A, BR::OBS_C, BR::H, BR::W, BR::P, BR::HAS_DRAW, BR::MAX_TURNS, BR::NSYM, BR::CELLS, BR::MAXK},
    {TM::A, TM::OBS_C, TM::H, TM::W, TM::P, TM::HAS_DRAW, TM::MAX_TURNS, TM::NSYM, TM::CELLS, TM::MAXK},
};
static const int k_num_games = 3;

extern "C" int azg_abi_version(void) { return AZG_ABI_VERSION; }
#ifndef AZG_SRC_SHA
#define AZG_SRC_SHA "unstamped"
#endif
extern "C" const char *azg_source_sha(void) { return AZG_SRC_SHA; }
extern "C" const char *azg_last_error(void) { return g_err.c_str(); }
extern "C" int azg_game_info_get(int game, azg_game_info *out) {
    if (game < 0 || game >= k_num_games || !out) return fail(AZG_E_INVALID_ARG, "unknown game id");
    *out = k_info[game];
    return AZG_OK;
}
extern "C" int azg_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

extern "C" uint64_t azg_tape_u64(uint64_t seed, uint64_t stream, uint64_t ctr) { return tape_u64(seed, stream, ctr); }
extern "C" double azg_tape_uniform(uint64_t seed, uint64_t stream, uint64_t ctr) { return u53(tape_u64(seed, stream, ctr)); }
extern "C" void azg_tape_shuffle_pos(uint64_t seed, uint64_t stream, uint64_t ctr, int k, int32_t *pos) {
    std::vector<uint64_t> key((size_t)k);
    for (int i = 0; i < k; i++) key[i] = tape_u64(seed, stream, ctr + (uint64_t)i);
    for (int i = 0; i < k; i++) {
        int r = 0;
        for (int j = 0; j < k; j++) r += (key[j] < key[i]) || (key[j] == key[i] && j < i);
        pos[i] = r;
    }
}

template <typename T> static int dalloc(azg_engine *e, T **p, size_t count) {
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    HIPCHK(hipMalloc(&q, bytes));
    HIPCHK(hipMemset(q, 0, bytes));
    e->allocs.push_back(q);
    *p = (T *)q;
    return AZG_OK;
}
#define DALLOC(ptr, count) do { int _r = dalloc(e, &(ptr), (size_t)(count)); if (_r != AZG_OK) return _r; } while (0)

// dispatch a templated kernel launch on the game id
#define GAME_SWITCH(e, CALL) \
    switch ((e)->cfg.game) { \
    case AZG_GAME_CONNECT4: { using G = C4; CALL; } break; \
    case AZG_GAME_BRANDUBH: { using G = BR; CALL; } break; \
    case AZG_GAME_TRIMOK: { using G = TM; CALL; } break; \
    default: return fail(AZG_E_UNSUPPORTED, "game has no device rules"); }

static void prof_begin(azg_engine *e, hipStream_t s, int fam, EvPair &p) {
    if (!e->profile) return;
    if (!e->pool.empty()) { p = e->pool.back(); e->pool.pop_back(); }
    else { (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b); }
    p.ext = fam != 2;                                          // select / backup: one kernel each; advance: a sequence of launches
    if (p.ext) g_kev = &p; else (void)hipEventRecord(p.a, s);
}
static void prof_end(azg_engine *e, hipStream_t s, int fam, EvPair &p) {
    if (!e->profile) return;
    if (!p.ext) (void)hipEventRecord(p.b, s);
    g_kev = nullptr;
    e->ev[fam].push_back(p);
    e->launches[fam]++;
}
static void prof_drain(azg_engine *e) {
    for (int f = 0; f < 3; f++) {
        for (auto &p : e->ev[f]) {
            (void)hipEventSynchronize(p.b);
            float t = 0; (void)hipEventElapsedTime(&t, p.a, p.b);
            e->ms[f] += t;
            e->pool.push_back(p);
        }
        e->ev[f].clear();
    }
}

extern "C" int azg_engine_create(const azg_config *cfg, azg_engine **out) {
    if (!cfg || !out) return fail(AZG_E_INVALID_ARG, "null argument");
    if (cfg->abi_version != AZG_ABI_VERSION) return fail(AZG_E_INVALID_ARG, "ABI version mismatch");
    if (cfg->game < 0 || cfg->game >= k_num_games) return fail(AZG_E_UNSUPPORTED, "game has no device rules");
    if (cfg->num_slots <= 0) return fail(AZG_E_INVALID_ARG, "num_slots must be > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(AZG_E_HIP, "no HIP device: libazg_hip has no CPU fallback");
    HIPCHK(hipSetDevice(cfg->device));
    azg_engine *e = new azg_engine();
    e->cfg = *cfg; e->gi = k_info[cfg->game];
    const azg_game_info &gi = e->gi;
    View &v = e->v;
    memset(&v, 0, sizeof(v));
    v.B = cfg->num_slots; v.arena = cfg->arena ? 1 : 0; v.T = v.arena ? gi.num_players : 1;
    // node store: two semi-spaces of `cap` nodes per tree (k_compact).  A move adds at most sims_per_move * max_children nodes to
    // the live space, and the space is compacted after a move once fewer than one move's worth of free nodes is left.  A search
    // that carries a fraction f of its visits into the played move keeps about f / (1 - f) moves' worth of nodes across moves:
    // the default capacity holds min(max_turns, 16) moves' worth (f up to 0.94 sustained; random-init nets keep 0.2-0.5 of ONE
    // move's worth), the worst case -- a game that never drops a sibling -- needs max_turns, which nodes_per_tree can ask for.
    // Overflow is loud: the sticky AZG_E_TREE_FULL
    const int sims = cfg->sims_per_move > 0 ? cfg->sims_per_move : 100;
    const long long per_move = (long long)sims * gi.max_children;
    const long long cap = cfg->nodes_per_tree > 0 ? cfg->nodes_per_tree : (long long)(gi.max_turns < 16 ? gi.max_turns : 16) * per_move + 64;
    if (cap >= (1 << 28)) { delete e; return fail(AZG_E_INVALID_ARG, "nodes_per_tree must be < 2^28"); }
    v.cap = (int)cap;
    v.compact_reserve = (int)(per_move < cap / 2 ? per_move : cap / 2);
    v.maxd = gi.max_turns + 2; v.max_hist = gi.max_turns + 1;
    v.ex_cap = cfg->example_capacity; v.res_cap = cfg->result_capacity > 0 ? cfg->result_capacity : 4 * v.B + 1024;
    v.add_noise = (cfg->add_root_noise && !v.arena) ? 1 : 0; v.add_temp = (cfg->add_root_temp && !v.arena) ? 1 : 0;
    v.symmetric = cfg->symmetric_samples; v.reset_thr = cfg->mcts_reset_threshold; v.games_cap = cfg->games_per_iteration;
    v.cpuct = cfg->cpuct; v.fpu_reduction = cfg->fpu_reduction; v.noise_frac = cfg->root_noise_frac;
    v.root_temp = cfg->root_policy_temp; v.arena_temp = cfg->arena_temp;
    v.seed = cfg->tape_seed; v.slot_base = cfg->slot_base;
    const size_t trees = (size_t)v.B * v.T;
    const int A = gi.action_size, NV = gi.num_players + 1, O = gi.obs_c * gi.obs_h * gi.obs_w;
    DALLOC(v.nodes, trees * 2 * (size_t)v.cap);
    DALLOC(v.hdr, trees);
    DALLOC(v.path, trees * (size_t)v.maxd);
    DALLOC(v.states, v.B); DALLOC(v.leaf_states, v.B);
    DALLOC(v.tape_ctr, v.B); DALLOC(v.next_reset, v.B); DALLOC(v.hist_len, v.B);
    const bool hist = !v.arena && v.ex_cap > 0;
    DALLOC(v.hist_state, hist ? (size_t)v.B * v.max_hist : 1);
    DALLOC(v.hist_pi, hist ? (size_t)v.B * v.max_hist * A : 1);
    if (!hist) v.max_hist = 0;
    DALLOC(v.last_action, v.B); DALLOC(v.fin_flag, v.B); DALLOC(v.fin_ridx, v.B); DALLOC(v.fin_counted, v.B); DALLOC(v.fin_soff, v.B);
    DALLOC(v.slot_sims, v.B); DALLOC(v.slot_exp, v.B);
    DALLOC(v.gcount, 8);
    DALLOC(v.ex_obs, (size_t)v.ex_cap * O); DALLOC(v.ex_pi, (size_t)v.ex_cap * A); DALLOC(v.ex_z, (size_t)v.ex_cap * NV);
    DALLOC(v.res_ws, (size_t)v.res_cap * NV); DALLOC(v.res_turns, v.res_cap); DALLOC(v.res_slot, v.res_cap);
    // temperature table (args.temp_scaling_fn iterated on the host)
    std::vector<float> tt;
    if (cfg->temp_table && cfg->temp_table_len > 0) tt.assign(cfg->temp_table, cfg->temp_table + cfg->temp_table_len);
    else tt.assign(1, cfg->start_temp);
    float *d_tt; DALLOC(d_tt, tt.size());
    HIPCHK(hipMemcpy(d_tt, tt.data(), tt.size() * sizeof(float), hipMemcpyHostToDevice));
    v.temp_table = d_tt; v.temp_len = (int)tt.size();
    DALLOC(e->d_p2i, 8); DALLOC(e->d_ok, 4);
#ifdef AZG_TREE_TIMING
    DALLOC(v.dbg, (size_t)v.B * 16);
#endif
    *out = e;
    int r = azg_engine_reset(e, nullptr);
    if (r != AZG_OK) { azg_engine_destroy(e); *out = nullptr; return r; }
    HIPCHK(hipDeviceSynchronize());
    return AZG_OK;
}

extern "C" int azg_engine_destroy(azg_engine *e) {
    if (!e) return AZG_OK;
    (void)hipDeviceSynchronize();
    prof_drain(e);
    for (auto &p : e->pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (void *p : e->allocs) (void)hipFree(p);
    if (e->d_perm) (void)hipFree(e->d_perm);
    if (e->d_utape) (void)hipFree(e->d_utape);
    if (e->d_noff) (void)hipFree(e->d_noff);
    if (e->d_npool) (void)hipFree(e->d_npool);
    delete e;
    return AZG_OK;
}

extern "C" int azg_engine_reset(azg_engine *e, void *stream) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    hipStream_t s = (hipStream_t)stream;
    GAME_SWITCH(e, hipLaunchKernelGGL((k_reset<G>), dim3(e->v.B), dim3(64), 0, s, e->v, 0, e->v.B, 1));
    HIPCHK(hipMemsetAsync(e->v.tape_ctr, 0, sizeof(uint64_t) * e->v.B, s));
    HIPCHK(hipMemsetAsync(e->v.gcount, 0, sizeof(int32_t) * 8, s));
    HIPCHK(hipMemsetAsync(e->v.slot_sims, 0, sizeof(int64_t) * e->v.B, s));
    HIPCHK(hipMemsetAsync(e->v.slot_exp, 0, sizeof(int64_t) * e->v.B, s));
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

static int check_range(azg_engine *e, int first, int count) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    if (first < 0 || count < 0 || first + count > e->v.B) return fail(AZG_E_INVALID_ARG, "slot range out of bounds");
    return AZG_OK;
}

extern "C" int azg_set_states(azg_engine *e, void *stream, int first, int count, const azg_state *host, int reset_trees) {
    int r = check_range(e, first, count); if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(e->v.states + first, host, sizeof(azg_state) * count, hipMemcpyHostToDevice, s));
    if (reset_trees) { GAME_SWITCH(e, hipLaunchKernelGGL((k_reset<G>), dim3(count), dim3(64), 0, s, e->v, first, count, 0)); }
    HIPCHK(hipStreamSynchronize(s));
    return AZG_OK;
}
extern "C" int azg_get_states(azg_engine *e, void *stream, int first, int count, azg_state *host) {
    int r = check_range(e, first, count); if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(host, e->v.states + first, sizeof(azg_state) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return AZG_OK;
}
extern "C" int azg_get_leaf_states(azg_engine *e, void *stream, int first, int count, azg_state *host) {
    int r = check_range(e, first, count); if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(host, e->v.leaf_states + first, sizeof(azg_state) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return AZG_OK;
}
extern "C" int azg_set_tape_counters(azg_engine *e, void *stream, int first, int count, const uint64_t *host) {
    int r = check_range(e, first, count); if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(e->v.tape_ctr + first, host, sizeof(uint64_t) * count, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return AZG_OK;
}
extern "C" int azg_get_tape_counters(azg_engine *e, void *stream, int first, int count, uint64_t *host) {
    int r = check_range(e, first, count); if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(host, e->v.tape_ctr + first, sizeof(uint64_t) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return AZG_OK;
}

extern "C" int azg_select(azg_engine *e, void *stream, void *obs, int obs_dtype, const int32_t *row_of_slot) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    if (obs_dtype < 0 || obs_dtype > 2) return fail(AZG_E_INVALID_ARG, "obs_dtype must be 0 (f32), 1 (f16) or 2 (f16 NHWC8)");
    hipStream_t s = (hipStream_t)stream;
    EvPair p; prof_begin(e, s, 0, p);
    if (obs_dtype == 0) { GAME_SWITCH(e, AZG_LAUNCH((k_select<G, float>), dim3(e->v.B), dim3(64), 0, s, e->v, (float *)obs, row_of_slot)); }
    else if (obs_dtype == 1) { GAME_SWITCH(e, AZG_LAUNCH((k_select<G, _Float16>), dim3(e->v.B), dim3(64), 0, s, e->v, (_Float16 *)obs, row_of_slot)); }
    else { GAME_SWITCH(e, AZG_LAUNCH((k_select<G, _Float16, true>), dim3(e->v.B), dim3(64), 0, s, e->v, (_Float16 *)obs, row_of_slot)); }
    prof_end(e, s, 0, p);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_arena_rows(azg_engine *e, void *stream, const int32_t *p2i_host, int32_t *row_of_slot, int32_t *rows_per_model) {
    if (!e || !p2i_host || !row_of_slot || !rows_per_model) return fail(AZG_E_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    if (e->gi.num_players > 8) return fail(AZG_E_UNSUPPORTED, "more than 8 players");
    SeatMap seat{};
    for (int i = 0; i < e->gi.num_players; i++) seat.v[i] = p2i_host[i];
    hipLaunchKernelGGL(k_arena_rows, dim3(1), dim3(64), 0, s, e->v, seat, (const uint32_t *)nullptr, row_of_slot, rows_per_model);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_arena_rows_seats(azg_engine *e, void *stream, const uint32_t *seat_of_slot, int32_t *row_of_slot, int32_t *rows_per_model) {
    if (!e || !seat_of_slot || !row_of_slot || !rows_per_model) return fail(AZG_E_INVALID_ARG, "null argument");
    if (e->gi.num_players > 8) return fail(AZG_E_UNSUPPORTED, "more than 8 players");
    hipLaunchKernelGGL(k_arena_rows, dim3(1), dim3(64), 0, (hipStream_t)stream, e->v, SeatMap{}, seat_of_slot, row_of_slot, rows_per_model);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_backup(azg_engine *e, void *stream, const float *policy, const float *value, const int32_t *row_of_slot, int flags) {
    if (!e || !policy || !value) return fail(AZG_E_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    View v = e->v;
    if (flags >= 0) { v.add_noise = (flags & AZG_FLAG_NOISE) ? 1 : 0; v.add_temp = (flags & AZG_FLAG_TEMP) ? 1 : 0; }
    EvPair p; prof_begin(e, s, 1, p);
    GAME_SWITCH(e, AZG_LAUNCH((k_backup<G>), dim3(e->v.B), dim3(64), 0, s, v, policy, value, row_of_slot));
    prof_end(e, s, 1, p);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_backup_select(azg_engine *e, void *stream, const float *policy, const float *value, const int32_t *row_of_slot, int flags,
                                 void *obs, int obs_dtype) {
    if (!e || !policy || !value) return fail(AZG_E_INVALID_ARG, "null argument");
    if (obs_dtype < 0 || obs_dtype > 2) return fail(AZG_E_INVALID_ARG, "obs_dtype must be 0 (f32), 1 (f16) or 2 (f16 NHWC8)");
    hipStream_t s = (hipStream_t)stream;
    View v = e->v;
    if (flags >= 0) { v.add_noise = (flags & AZG_FLAG_NOISE) ? 1 : 0; v.add_temp = (flags & AZG_FLAG_TEMP) ? 1 : 0; }
    EvPair p; prof_begin(e, s, 1, p);
    const int A = e->gi.action_size;
    if (obs_dtype == 0) { GAME_SWITCH(e, AZG_LAUNCH((k_backup_select2<G, float, false, IN_PROBS>), dim3(v.B), dim3(128), 0, s, v, policy, value, A, (float *)obs, row_of_slot, 1, HeadRows{})); }
    else if (obs_dtype == 1) { GAME_SWITCH(e, AZG_LAUNCH((k_backup_select2<G, _Float16, false, IN_PROBS>), dim3(v.B), dim3(128), 0, s, v, policy, value, A, (_Float16 *)obs, row_of_slot, 1, HeadRows{})); }
    else { GAME_SWITCH(e, AZG_LAUNCH((k_backup_select2<G, _Float16, true, IN_PROBS>), dim3(v.B), dim3(128), 0, s, v, policy, value, A, (_Float16 *)obs, row_of_slot, 1, HeadRows{})); }
    prof_end(e, s, 1, p);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_backup_select_logits(azg_engine *e, void *stream, const float *logits, int logits_stride, const int32_t *row_of_slot,
                                        int flags, void *obs, int obs_dtype, int do_select) {
    if (!e || !logits) return fail(AZG_E_INVALID_ARG, "null argument");
    if (obs_dtype < 0 || obs_dtype > 2) return fail(AZG_E_INVALID_ARG, "obs_dtype must be 0 (f32), 1 (f16) or 2 (f16 NHWC8)");
    if (logits_stride < e->gi.action_size + e->gi.num_players + 1 || e->gi.action_size > 1024)
        return fail(AZG_E_INVALID_ARG, "logits_stride must hold A + P + 1 logits (A <= 1024)");
    hipStream_t s = (hipStream_t)stream;
    View v = e->v;
    if (flags >= 0) { v.add_noise = (flags & AZG_FLAG_NOISE) ? 1 : 0; v.add_temp = (flags & AZG_FLAG_TEMP) ? 1 : 0; }
    EvPair p; prof_begin(e, s, 1, p);
    const float *nov = nullptr;
    if (obs_dtype == 0) { GAME_SWITCH(e, AZG_LAUNCH((k_backup_select2<G, float, false, IN_LOGITS>), dim3(v.B), dim3(128), 0, s, v, logits, nov, logits_stride, (float *)obs, row_of_slot, do_select, HeadRows{})); }
    else if (obs_dtype == 1) { GAME_SWITCH(e, AZG_LAUNCH((k_backup_select2<G, _Float16, false, IN_LOGITS>), dim3(v.B), dim3(128), 0, s, v, logits, nov, logits_stride, (_Float16 *)obs, row_of_slot, do_select, HeadRows{})); }
    else { GAME_SWITCH(e, AZG_LAUNCH((k_backup_select2<G, _Float16, true, IN_LOGITS>), dim3(v.B), dim3(128), 0, s, v, logits, nov, logits_stride, (_Float16 *)obs, row_of_slot, do_select, HeadRows{})); }
    prof_end(e, s, 1, p);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_backup_select_features(azg_engine *e, void *stream, const void *feat, int feat_k, const void *head_rows, const float *head_b,
                                          const int32_t *row_of_slot, int flags, void *obs, int obs_dtype, int do_select) {
    if (!e || !feat || !head_rows || !head_b) return fail(AZG_E_INVALID_ARG, "null argument");
    if (obs_dtype < 0 || obs_dtype > 2) return fail(AZG_E_INVALID_ARG, "obs_dtype must be 0 (f32), 1 (f16) or 2 (f16 NHWC8)");
    if (e->gi.action_size > 1024) return fail(AZG_E_INVALID_ARG, "A <= 1024");
    hipStream_t s = (hipStream_t)stream;
    View v = e->v;
    if (flags >= 0) { v.add_noise = (flags & AZG_FLAG_NOISE) ? 1 : 0; v.add_temp = (flags & AZG_FLAG_TEMP) ? 1 : 0; }
    const HeadRows hd{(const _Float16 *)head_rows, head_b, feat_k};
    int want = 0;
    GAME_SWITCH(e, want = head_fk<G>());
    if (feat_k != want) return fail(AZG_E_INVALID_ARG, "feat_k must be cells x 16 rounded up to a multiple of 32 for this game");
    EvPair p; prof_begin(e, s, 1, p);
    const float *nov = nullptr, *f = (const float *)feat;
    if (obs_dtype == 0) { GAME_SWITCH(e, AZG_LAUNCH((k_backup_select2<G, float, false, IN_FEATURES>), dim3(v.B), dim3(128), 0, s, v, f, nov, 0, (float *)obs, row_of_slot, do_select, hd)); }
    else if (obs_dtype == 1) { GAME_SWITCH(e, AZG_LAUNCH((k_backup_select2<G, _Float16, false, IN_FEATURES>), dim3(v.B), dim3(128), 0, s, v, f, nov, 0, (_Float16 *)obs, row_of_slot, do_select, hd)); }
    else { GAME_SWITCH(e, AZG_LAUNCH((k_backup_select2<G, _Float16, true, IN_FEATURES>), dim3(v.B), dim3(128), 0, s, v, f, nov, 0, (_Float16 *)obs, row_of_slot, do_select, hd)); }
    prof_end(e, s, 1, p);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_leaf_heads_sparse_f16(azg_engine *e, void *stream, const void *feat, int feat_k, const void *head_rows, const float *head_b,
                                         const int32_t *row_of_slot, float *logits, int logits_stride) {
    if (!e || !feat || !head_rows || !head_b || !logits) return fail(AZG_E_INVALID_ARG, "null argument");
    if (logits_stride < e->gi.action_size + e->gi.num_players + 1 || e->gi.action_size > 1024)
        return fail(AZG_E_INVALID_ARG, "logits_stride must hold A + P + 1 logits (A <= 1024)");
    int want = 0;
    GAME_SWITCH(e, want = head_fk<G>());
    if (feat_k != want) return fail(AZG_E_INVALID_ARG, "feat_k must be cells x 16 rounded up to a multiple of 32 for this game");
    const HeadRows hd{(const _Float16 *)head_rows, head_b, feat_k};
    GAME_SWITCH(e, hipLaunchKernelGGL((k_leaf_heads_sparse<G>), dim3(e->v.B), dim3(64), 0, (hipStream_t)stream, e->v, (const _Float16 *)feat, hd,
                                      row_of_slot, logits, logits_stride));
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_advance(azg_engine *e, void *stream, int record_history) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    hipStream_t s = (hipStream_t)stream;
    EvPair p; prof_begin(e, s, 2, p);
    GAME_SWITCH(e, {
        hipLaunchKernelGGL((k_play<G>), dim3(e->v.B), dim3(64), 0, s, e->v, record_history);
        hipLaunchKernelGGL((k_finalize<G>), dim3(1), dim3(64), 0, s, e->v, (const int32_t *)nullptr);
        hipLaunchKernelGGL((k_emit_samples<G>), dim3(e->v.B, G::NSYM), dim3(64), 0, s, e->v);
        hipLaunchKernelGGL((k_emit<G>), dim3(e->v.B), dim3(64), 0, s, e->v);
        hipLaunchKernelGGL((k_compact<G>), dim3(e->v.B * e->v.T), dim3(64), 0, s, e->v, 0, 0);
    });
    prof_end(e, s, 2, p);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_advance_begin(azg_engine *e, void *stream, int record_history, int32_t *fin_host) {
    if (!e || !fin_host) return fail(AZG_E_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    GAME_SWITCH(e, hipLaunchKernelGGL((k_play<G>), dim3(e->v.B), dim3(64), 0, s, e->v, record_history));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(fin_host, e->v.fin_flag, sizeof(int32_t) * e->v.B, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return AZG_OK;
}

extern "C" int azg_advance_commit(azg_engine *e, void *stream, const int32_t *counted_host) {
    if (!e || !counted_host) return fail(AZG_E_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(e->v.fin_counted, counted_host, sizeof(int32_t) * e->v.B, hipMemcpyHostToDevice, s));
    GAME_SWITCH(e, {
        hipLaunchKernelGGL((k_finalize<G>), dim3(1), dim3(64), 0, s, e->v, (const int32_t *)e->v.fin_counted);
        hipLaunchKernelGGL((k_emit_samples<G>), dim3(e->v.B, G::NSYM), dim3(64), 0, s, e->v);
        hipLaunchKernelGGL((k_emit<G>), dim3(e->v.B), dim3(64), 0, s, e->v);
        hipLaunchKernelGGL((k_compact<G>), dim3(e->v.B * e->v.T), dim3(64), 0, s, e->v, 0, 0);
    });
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_root_counts(azg_engine *e, void *stream, int32_t *counts) {
    if (!e || !counts) return fail(AZG_E_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    GAME_SWITCH(e, hipLaunchKernelGGL((k_root_stats<G>), dim3(e->v.B), dim3(64), 0, s, e->v, 0, 1.0f, 0, counts, (float *)nullptr, (float *)nullptr));
    HIPCHK(hipGetLastError());
    return AZG_OK;
}
extern "C" int azg_root_probs(azg_engine *e, void *stream, float temp, float *probs) {
    if (!e || !probs) return fail(AZG_E_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    GAME_SWITCH(e, hipLaunchKernelGGL((k_root_stats<G>), dim3(e->v.B), dim3(64), 0, s, e->v, 1, temp, 0, (int32_t *)nullptr, probs, (float *)nullptr));
    HIPCHK(hipGetLastError());
    return AZG_OK;
}
extern "C" int azg_root_value(azg_engine *e, void *stream, int average, float *values) {
    if (!e || !values) return fail(AZG_E_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    GAME_SWITCH(e, hipLaunchKernelGGL((k_root_stats<G>), dim3(e->v.B), dim3(64), 0, s, e->v, 2, 1.0f, average, (int32_t *)nullptr, (float *)nullptr, values));
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_update_root(azg_engine *e, void *stream, int slot, int action) {
    int r = check_range(e, slot, 1); if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    GAME_SWITCH(e, hipLaunchKernelGGL((k_update_root<G>), dim3(1), dim3(64), 0, s, e->v, slot, action, e->d_ok));
    // (only this slot's trees: a compaction voids the tree's pending find_leaf, and another slot of a per-slot driven engine may
    //  sit between azg_select and azg_backup)
    GAME_SWITCH(e, hipLaunchKernelGGL((k_compact<G>), dim3(e->v.T), dim3(64), 0, s, e->v, 0, slot * e->v.T));
    int32_t ok = 0;
    HIPCHK(hipMemcpyAsync(&ok, e->d_ok, sizeof(ok), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (!ok) return fail(AZG_E_INVALID_ACTION, "Invalid action encountered while updating root: " + std::to_string(action));
    return AZG_OK;
}

extern "C" int azg_compact(azg_engine *e, void *stream, int slot, int force) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    if (slot >= e->v.B) return fail(AZG_E_INVALID_ARG, "slot out of range");
    hipStream_t s = (hipStream_t)stream;
    const int first = slot < 0 ? 0 : slot * e->v.T, count = slot < 0 ? e->v.B * e->v.T : e->v.T;
    GAME_SWITCH(e, hipLaunchKernelGGL((k_compact<G>), dim3(count), dim3(64), 0, s, e->v, force ? 1 : 0, first));
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_engine_info(azg_engine *e, int32_t *out8) {
    if (!e || !out8) return fail(AZG_E_INVALID_ARG, "null argument");
    out8[0] = e->v.cap; out8[1] = e->v.compact_reserve; out8[2] = e->v.T; out8[3] = e->v.maxd; out8[4] = e->v.B;
    out8[5] = e->v.ex_cap; out8[6] = e->v.res_cap; out8[7] = e->cfg.sims_per_move > 0 ? e->cfg.sims_per_move : 100;
    return AZG_OK;
}

extern "C" int azg_set_random_tape(azg_engine *e, void *stream, const int16_t *ranks_host, const double *u_host, const int32_t *noise_off_host,
                                   const float *noise_pool_host, int noise_len, int len) {
    (void)stream;
    if (!e || len < 0 || noise_len < 0) return fail(AZG_E_INVALID_ARG, "null engine or negative length");
    if ((u_host || noise_off_host) && !ranks_host) return fail(AZG_E_INVALID_ARG, "choice / noise tapes come with a shuffle tape (one counter indexes all three)");
    if ((noise_off_host != nullptr) != (noise_pool_host != nullptr)) return fail(AZG_E_INVALID_ARG, "noise offsets and noise pool come together");
    const size_t n = (size_t)e->v.B * (size_t)(len > 0 ? len : 0);
    // every rank is an offset into an expansion's block of k <= max_children nodes: refuse anything else here, on the host (the device
    // checks rank < k per expansion as well and raises the sticky AZG_E_INVALID_ARG)
    if (ranks_host) for (size_t i = 0; i < n; i++)
        if (ranks_host[i] < 0 || ranks_host[i] >= e->gi.max_children) return fail(AZG_E_INVALID_ARG, "shuffle tape: a rank is outside [0, max_children)");
    if (u_host) for (size_t i = 0; i < n; i++)
        if (!(u_host[i] >= 0.0 && u_host[i] < 1.0)) return fail(AZG_E_INVALID_ARG, "choice tape: a uniform is outside [0, 1)");
    if (noise_off_host) for (size_t i = 0; i < n; i++)
        if (noise_off_host[i] < -1 || noise_off_host[i] >= noise_len) return fail(AZG_E_INVALID_ARG, "noise tape: an offset is outside the pool");
    // the View -- and with it these pointers -- is passed BY VALUE to every launch, also to launches captured in a hipGraph: work of ANY
    // stream may still read the old tapes, so the whole device is drained before they are freed.  A graph captured while a tape was set
    // keeps replaying with the pointers it captured: re-capture after changing the tape
    HIPCHK(hipDeviceSynchronize());
    if (e->d_perm) { (void)hipFree(e->d_perm); e->d_perm = nullptr; }
    if (e->d_utape) { (void)hipFree(e->d_utape); e->d_utape = nullptr; }
    if (e->d_noff) { (void)hipFree(e->d_noff); e->d_noff = nullptr; }
    if (e->d_npool) { (void)hipFree(e->d_npool); e->d_npool = nullptr; }
    e->v.perm_tape = nullptr; e->v.perm_len = 0; e->v.u_tape = nullptr; e->v.noise_off = nullptr; e->v.noise_pool = nullptr; e->v.noise_len = 0;
    if (!ranks_host || len == 0) return AZG_OK;
    HIPCHK(hipMalloc((void **)&e->d_perm, n * sizeof(int16_t)));
    HIPCHK(hipMemcpy(e->d_perm, ranks_host, n * sizeof(int16_t), hipMemcpyHostToDevice));
    if (u_host) {
        HIPCHK(hipMalloc((void **)&e->d_utape, n * sizeof(double)));
        HIPCHK(hipMemcpy(e->d_utape, u_host, n * sizeof(double), hipMemcpyHostToDevice));
    }
    if (noise_off_host) {
        HIPCHK(hipMalloc((void **)&e->d_noff, n * sizeof(int32_t)));
        HIPCHK(hipMemcpy(e->d_noff, noise_off_host, n * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void **)&e->d_npool, (size_t)(noise_len > 0 ? noise_len : 1) * sizeof(float)));
        HIPCHK(hipMemcpy(e->d_npool, noise_pool_host, (size_t)noise_len * sizeof(float), hipMemcpyHostToDevice));
    }
    e->v.perm_tape = e->d_perm; e->v.perm_len = len;
    e->v.u_tape = e->d_utape; e->v.noise_off = e->d_noff; e->v.noise_pool = e->d_npool; e->v.noise_len = noise_len;
    return AZG_OK;
}

extern "C" int azg_set_shuffle_tape(azg_engine *e, void *stream, const int16_t *ranks_host, int len) {
    return azg_set_random_tape(e, stream, ranks_host, nullptr, nullptr, nullptr, 0, len);
}

// bounds-checked builds (-DAZG_DEBUG_BOUNDS): the first check that failed (0: none; site numbers: csrc/azg_kernels.h AZG_BOUNDS_OK).
// *checked = 1 if this binary carries the checks at all.
extern "C" int azg_debug_bounds_site(azg_engine *e, void *stream, int32_t *site, int32_t *checked) {
    if (!e || !site || !checked) return fail(AZG_E_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(site, e->v.gcount + GC_BOUNDS_SITE, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
#ifdef AZG_DEBUG_BOUNDS
    *checked = 1;
#else
    *checked = 0;
#endif
    return AZG_OK;
}

#ifdef AZG_DEBUG_BOUNDS
// bounds-checked builds only (not part of include/azg.h): overwrite the root's first_child of a slot's tree -- the positive control of
// tools/debug_soak.py (a child block outside the live allocation must be caught by the checks, not read)
extern "C" int azg_debug_poke_root_fc(azg_engine *e, void *stream, int slot, int32_t fc) {
    int r = check_range(e, slot, 1); if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(e->v.hdr + (size_t)slot * e->v.T) + 16, &fc, sizeof(fc), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return AZG_OK;
}
#endif

extern "C" int azg_set_root_flags(azg_engine *e, int flags) {
    if (!e || flags < 0) return fail(AZG_E_INVALID_ARG, "null engine or negative flags");
    e->v.add_noise = ((flags & AZG_FLAG_NOISE) && !e->v.arena) ? 1 : 0; e->v.add_temp = ((flags & AZG_FLAG_TEMP) && !e->v.arena) ? 1 : 0;
    return AZG_OK;
}

// ---- snapshot of one slot's search state (pickling of the single-tree MCTS class: MCTS.pyx:8 auto_pickle) ----
// layout: int64 magic, int32 game, T, maxd, pad | azg_state root, leaf | uint64 tape_ctr | per tree: TreeHdr, PathEnt[maxd], Node[alloc]
static const int64_t k_snap_magic = 0x315A4E53475A41LL;     // "AZGSNZ1"
struct SnapHead { int64_t magic; int32_t game, T, maxd, layout; azg_state root, leaf; uint64_t ctr; };
// (layout: the record sizes and the ABI version the snapshot was written with -- a change of Node / TreeHdr / PathEnt invalidates old pickles)
static const int32_t k_snap_layout = (int32_t)(sizeof(Node) | (sizeof(TreeHdr) << 8) | (sizeof(PathEnt) << 16) | ((unsigned)AZG_ABI_VERSION << 24));

extern "C" int64_t azg_slot_export(azg_engine *e, void *stream, int slot, void *host_buf, int64_t nbytes) {
    int r = check_range(e, slot, 1); if (r) return r;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipStreamSynchronize(s));
    const int T = e->v.T, maxd = e->v.maxd;
    std::vector<TreeHdr> hdr((size_t)T);
    HIPCHK(hipMemcpy(hdr.data(), e->v.hdr + (size_t)slot * T, sizeof(TreeHdr) * T, hipMemcpyDeviceToHost));
    int64_t need = (int64_t)sizeof(SnapHead);
    for (int t = 0; t < T; t++) need += (int64_t)sizeof(TreeHdr) + (int64_t)sizeof(PathEnt) * maxd + (int64_t)sizeof(Node) * hdr[t].alloc;
    if (!host_buf) return need;                              // size query
    if (nbytes < need) return fail(AZG_E_INVALID_ARG, "snapshot buffer too small");
    char *o = (char *)host_buf;
    SnapHead sh; memset(&sh, 0, sizeof(sh));
    sh.magic = k_snap_magic; sh.game = e->cfg.game; sh.T = T; sh.maxd = maxd; sh.layout = k_snap_layout;
    HIPCHK(hipMemcpy(&sh.root, e->v.states + slot, sizeof(azg_state), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&sh.leaf, e->v.leaf_states + slot, sizeof(azg_state), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&sh.ctr, e->v.tape_ctr + slot, sizeof(uint64_t), hipMemcpyDeviceToHost));
    memcpy(o, &sh, sizeof(sh)); o += sizeof(sh);
    for (int t = 0; t < T; t++) {
        const size_t tt = (size_t)slot * T + t;
        memcpy(o, &hdr[t], sizeof(TreeHdr)); o += sizeof(TreeHdr);
        HIPCHK(hipMemcpy(o, e->v.path + tt * maxd, sizeof(PathEnt) * maxd, hipMemcpyDeviceToHost)); o += sizeof(PathEnt) * maxd;
        if (hdr[t].alloc > 0) HIPCHK(hipMemcpy(o, e->v.nodes + tt * 2 * e->v.cap + hdr[t].base, sizeof(Node) * hdr[t].alloc, hipMemcpyDeviceToHost));
        o += sizeof(Node) * hdr[t].alloc;
    }
    return need;
}

extern "C" int azg_slot_import(azg_engine *e, void *stream, int slot, const void *host_buf, int64_t nbytes) {
    int r = check_range(e, slot, 1); if (r) return r;
    if (!host_buf || nbytes < (int64_t)sizeof(SnapHead)) return fail(AZG_E_INVALID_ARG, "snapshot truncated");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipStreamSynchronize(s));
    const char *o = (const char *)host_buf, *end = o + nbytes;
    SnapHead sh; memcpy(&sh, o, sizeof(sh)); o += sizeof(sh);
    if (sh.magic != k_snap_magic || sh.game != e->cfg.game || sh.T != e->v.T || sh.maxd != e->v.maxd)
        return fail(AZG_E_INVALID_ARG, "snapshot does not belong to this kind of engine (game / arena mode)");
    if (sh.layout != k_snap_layout) return fail(AZG_E_INVALID_ARG, "snapshot was written by a library with another record layout / ABI version");
    {                                                        // validate the whole snapshot before anything is written
        const char *q = o;
        for (int t = 0; t < sh.T; t++) {
            if (end - q < (int64_t)(sizeof(TreeHdr) + sizeof(PathEnt) * sh.maxd)) return fail(AZG_E_INVALID_ARG, "snapshot truncated");
            TreeHdr h; memcpy(&h, q, sizeof(h));
            if (h.alloc < 0 || h.alloc > e->v.cap) return fail(AZG_E_TREE_FULL, "snapshot holds more nodes than this engine's nodes_per_tree");
            q += sizeof(TreeHdr) + sizeof(PathEnt) * sh.maxd;
            if (end - q < (int64_t)sizeof(Node) * h.alloc) return fail(AZG_E_INVALID_ARG, "snapshot truncated");
            // every index the kernels will follow must stay inside the live nodes: a corrupted or hand-made pickle must not become an
            // out-of-bounds device access on the next find_leaf / process_results
            const unsigned lk = h.leaf_info & 0xFFFFu;
            if (h.depth < 0 || h.depth > sh.maxd || h.max_depth < 0 || h.leaf < LEAF_IS_ROOT || h.leaf >= h.alloc || h.leaf_fc < -1 ||
                (h.leaf_fc >= 0 && (int64_t)h.leaf_fc + (int64_t)lk > h.alloc))
                return fail(AZG_E_INVALID_ARG, "snapshot header points outside its nodes");
            if (h.root.first_child < -1 || (h.root.first_child >= 0 && (int64_t)h.root.first_child + h.root.nchild > h.alloc))
                return fail(AZG_E_INVALID_ARG, "snapshot root points outside its nodes");
            const PathEnt *pe = reinterpret_cast<const PathEnt *>(q - sizeof(PathEnt) * sh.maxd);
            for (int d = 0; d < h.depth; d++) {
                PathEnt ent; memcpy(&ent, pe + d, sizeof(ent));
                if ((int)(ent.idx_mover & 0x0FFFFFFFu) >= h.alloc) return fail(AZG_E_INVALID_ARG, "snapshot path points outside its nodes");
            }
            for (int i = 0; i < h.alloc; i++) {
                Node nd; memcpy(&nd, q + sizeof(Node) * (size_t)i, sizeof(nd));
                if (nd.first_child < -1 || (nd.first_child >= 0 && (int64_t)nd.first_child + nd.nchild > h.alloc))
                    return fail(AZG_E_INVALID_ARG, "snapshot node points outside its nodes");
            }
            q += sizeof(Node) * h.alloc;
        }
    }
    HIPCHK(hipMemcpy(e->v.states + slot, &sh.root, sizeof(azg_state), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->v.leaf_states + slot, &sh.leaf, sizeof(azg_state), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->v.tape_ctr + slot, &sh.ctr, sizeof(uint64_t), hipMemcpyHostToDevice));
    for (int t = 0; t < sh.T; t++) {
        const size_t tt = (size_t)slot * sh.T + t;
        if (end - o < (int64_t)(sizeof(TreeHdr) + sizeof(PathEnt) * sh.maxd)) return fail(AZG_E_INVALID_ARG, "snapshot truncated");
        TreeHdr h; memcpy(&h, o, sizeof(h)); o += sizeof(h);
        if (h.alloc < 0 || h.alloc > e->v.cap) return fail(AZG_E_TREE_FULL, "snapshot holds more nodes than this engine's nodes_per_tree");
        h.base = 0;                                          // the live nodes land in semi-space 0 (indices are relative to the base)
        HIPCHK(hipMemcpy(e->v.hdr + tt, &h, sizeof(h), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->v.path + tt * sh.maxd, o, sizeof(PathEnt) * sh.maxd, hipMemcpyHostToDevice)); o += sizeof(PathEnt) * sh.maxd;
        if (end - o < (int64_t)sizeof(Node) * h.alloc) return fail(AZG_E_INVALID_ARG, "snapshot truncated");
        if (h.alloc > 0) HIPCHK(hipMemcpy(e->v.nodes + tt * 2 * e->v.cap, o, sizeof(Node) * h.alloc, hipMemcpyHostToDevice));
        o += sizeof(Node) * h.alloc;
    }
    return AZG_OK;
}

static int tree_of(azg_engine *e, int slot, int tree) { return slot * e->v.T + tree; }

// node `node` of a tree (AZG_NODE_ROOT = the root, which lives in the header) and the base of the tree's live semi-space
static int fetch_node(azg_engine *e, int t, int node, TreeHdr &h, Node &nd, Node *&base) {
    HIPCHK(hipMemcpy(&h, e->v.hdr + t, sizeof(h), hipMemcpyDeviceToHost));
    base = e->v.nodes + (size_t)t * 2 * e->v.cap + h.base;
    if (node < 0) { nd = h.root; return AZG_OK; }
    if (node >= h.alloc) return fail(AZG_E_INVALID_ARG, "node index out of range");
    HIPCHK(hipMemcpy(&nd, base + node, sizeof(Node), hipMemcpyDeviceToHost));
    return AZG_OK;
}

extern "C" int azg_root_children(azg_engine *e, void *stream, int slot, int tree, int max_k, int32_t *a, int32_t *n, float *q, float *p, float *vv) {
    int r = check_range(e, slot, 1); if (r) return r;
    if (tree < 0 || tree >= e->v.T) return fail(AZG_E_INVALID_ARG, "tree index out of range");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    TreeHdr h; Node root; Node *base;
    r = fetch_node(e, tree_of(e, slot, tree), -1, h, root, base); if (r) return r;
    int k = root.nchild;
    if (k > max_k) return fail(AZG_E_INVALID_ARG, "max_k too small");
    if (k == 0) return 0;
    std::vector<Node> ch((size_t)k);
    HIPCHK(hipMemcpy(ch.data(), base + root.first_child, sizeof(Node) * k, hipMemcpyDeviceToHost));
    for (int i = 0; i < k; i++) { a[i] = ch[i].a; n[i] = ch[i].n; q[i] = ch[i].q; p[i] = ch[i].p; vv[i] = ch[i].v; }
    return k;
}

extern "C" int azg_node_children(azg_engine *e, void *stream, int slot, int tree, int node, int max_k, int32_t *idx, int32_t *a, int32_t *n, float *q, float *p, float *vv,
                                 int32_t *player, int32_t *e_bits) {
    int r = check_range(e, slot, 1); if (r) return r;
    if (tree < 0 || tree >= e->v.T) return fail(AZG_E_INVALID_ARG, "tree index out of range");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    TreeHdr h; Node nd; Node *base;
    r = fetch_node(e, tree_of(e, slot, tree), node, h, nd, base); if (r) return r;
    int k = nd.nchild;
    if (k > max_k) return fail(AZG_E_INVALID_ARG, "max_k too small");
    if (k == 0) return 0;
    std::vector<Node> ch((size_t)k);
    HIPCHK(hipMemcpy(ch.data(), base + nd.first_child, sizeof(Node) * k, hipMemcpyDeviceToHost));
    for (int i = 0; i < k; i++) {
        idx[i] = nd.first_child + i; a[i] = ch[i].a; n[i] = ch[i].n; q[i] = ch[i].q; p[i] = ch[i].p; vv[i] = ch[i].v;
        if (player) player[i] = ch[i].player;
        if (e_bits) e_bits[i] = ch[i].e;
    }
    return k;
}

extern "C" int azg_reset_max_depth(azg_engine *e, void *stream) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    hipLaunchKernelGGL(k_reset_max_depth, dim3(((int)(e->v.B * e->v.T) + 63) / 64), dim3(64), 0, (hipStream_t)stream, e->v);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_tree_info(azg_engine *e, void *stream, int slot, int tree, int32_t *out8) {
    int r = check_range(e, slot, 1); if (r) return r;
    if (tree < 0 || tree >= e->v.T) return fail(AZG_E_INVALID_ARG, "tree index out of range");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    TreeHdr h;
    HIPCHK(hipMemcpy(&h, e->v.hdr + tree_of(e, slot, tree), sizeof(h), hipMemcpyDeviceToHost));
    const Node &root = h.root;
    out8[0] = root.n; memcpy(&out8[1], &root.q, 4); memcpy(&out8[2], &root.v, 4);
    out8[3] = root.player; out8[4] = root.e; out8[5] = h.depth; out8[6] = h.max_depth; out8[7] = h.alloc;
    return AZG_OK;
}

extern "C" int azg_last_path(azg_engine *e, void *stream, int slot, int tree, int max_len, int32_t *actions) {
    int r = check_range(e, slot, 1); if (r) return r;
    if (tree < 0 || tree >= e->v.T) return fail(AZG_E_INVALID_ARG, "tree index out of range");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    const int t = tree_of(e, slot, tree);
    TreeHdr h;
    HIPCHK(hipMemcpy(&h, e->v.hdr + t, sizeof(h), hipMemcpyDeviceToHost));
    if (h.depth > max_len) return fail(AZG_E_INVALID_ARG, "max_len too small");
    std::vector<PathEnt> path((size_t)(h.depth > 0 ? h.depth : 1));
    if (h.depth > 0) HIPCHK(hipMemcpy(path.data(), e->v.path + (size_t)t * e->v.maxd, sizeof(PathEnt) * h.depth, hipMemcpyDeviceToHost));
    const Node *base = e->v.nodes + (size_t)t * 2 * e->v.cap + h.base;
    for (int d = 0; d < h.depth; d++) {
        Node nd;
        HIPCHK(hipMemcpy(&nd, base + (path[d].idx_mover & 0x0FFFFFFFu), sizeof(Node), hipMemcpyDeviceToHost));
        actions[d] = nd.a;
    }
    return h.depth;
}

extern "C" int azg_read_counters(azg_engine *e, void *stream, azg_counters *out) {
    if (!e || !out) return fail(AZG_E_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    int32_t gc[8];
    std::vector<int64_t> sims((size_t)e->v.B), exps((size_t)e->v.B);
    hipLaunchKernelGGL(k_max_nodes, dim3(1), dim3(256), 0, s, e->v);
    HIPCHK(hipMemcpyAsync(gc, e->v.gcount, sizeof(gc), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(sims.data(), e->v.slot_sims, sizeof(int64_t) * e->v.B, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(exps.data(), e->v.slot_exp, sizeof(int64_t) * e->v.B, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < e->v.B; i++) { out->sims += sims[i]; out->expansions += exps[i]; }
    out->games_played = gc[GC_GAMES]; out->num_results = gc[GC_RESULTS]; out->num_examples = gc[GC_EXAMPLES];
    out->error = gc[GC_ERROR]; out->max_nodes_used = gc[GC_MAXNODES]; out->max_nodes_kept = gc[GC_MAXLIVE];
    return AZG_OK;
}

extern "C" int azg_examples_dev(azg_engine *e, float **obs, float **pi, float **z) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    if (obs) *obs = e->v.ex_obs; if (pi) *pi = e->v.ex_pi; if (z) *z = e->v.ex_z;
    return AZG_OK;
}

extern "C" int azg_copy_examples(azg_engine *e, void *stream, int first, int count, float *obs, float *pi, float *z) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    if (first < 0 || count < 0 || first + count > e->v.ex_cap) return fail(AZG_E_INVALID_ARG, "example range out of bounds");
    if (count == 0) return AZG_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t A = e->gi.action_size, NV = e->gi.num_players + 1, O = (size_t)e->gi.obs_c * e->gi.obs_h * e->gi.obs_w;
    if (obs) HIPCHK(hipMemcpyAsync(obs, e->v.ex_obs + (size_t)first * O, sizeof(float) * count * O, hipMemcpyDeviceToDevice, s));
    if (pi) HIPCHK(hipMemcpyAsync(pi, e->v.ex_pi + (size_t)first * A, sizeof(float) * count * A, hipMemcpyDeviceToDevice, s));
    if (z) HIPCHK(hipMemcpyAsync(z, e->v.ex_z + (size_t)first * NV, sizeof(float) * count * NV, hipMemcpyDeviceToDevice, s));
    return AZG_OK;
}

extern "C" int azg_read_results(azg_engine *e, void *stream, int first, int count, uint8_t *ws, int32_t *turns, int32_t *slot) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    if (first < 0 || count < 0 || first + count > e->v.res_cap) return fail(AZG_E_INVALID_ARG, "result range out of bounds");
    hipStream_t s = (hipStream_t)stream;
    const int NV = e->gi.num_players + 1;
    if (count == 0) return AZG_OK;
    HIPCHK(hipMemcpyAsync(ws, e->v.res_ws + (size_t)first * NV, (size_t)count * NV, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(turns, e->v.res_turns + first, sizeof(int32_t) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(slot, e->v.res_slot + first, sizeof(int32_t) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return AZG_OK;
}

extern "C" int azg_clear_outputs(azg_engine *e, void *stream) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    HIPCHK(hipMemsetAsync(e->v.gcount, 0, sizeof(int32_t) * 3, (hipStream_t)stream));
    return AZG_OK;
}

extern "C" int azg_last_actions_dev(azg_engine *e, int32_t **actions) {
    if (!e || !actions) return fail(AZG_E_INVALID_ARG, "null argument");
    *actions = e->v.last_action;
    return AZG_OK;
}

// occ (optional): the workgroups of THIS instantiation a CU of this device holds at once (the occupancy query with the launch's own LDS
// size) -- what the tile choice of the persistent launches is derived from instead of a constant for one chip
template <int H, int W, int BOARDS, int C, int PSPLIT = 1, class SEARCH = NoSearch, int KSPLIT = 1>
static int launch_tower(hipStream_t s, const TowerParams &P, SEARCH sa = SEARCH{}, bool init_only = false, int *occ = nullptr) {
    constexpr bool IS_SEARCH = !__is_same(SEARCH, NoSearch);
    using GEO = TowerGeom<H, W, BOARDS, C>;
    constexpr size_t LDS_IMG = []() {                          // the image (+ the wide search mode's scratch behind it)
        if constexpr (IS_SEARCH) { if constexpr (SEARCH::WIDE) return (size_t)GEO::TILE + (size_t)WideLds<typename SEARCH::Game, H * W, BOARDS, wide_solo<C, PSPLIT, KSPLIT, BOARDS>()>::BYTES; }
        return (size_t)GEO::TILE;
    }();
    // (+ the k-split exchange area: per wave one 1 KB accumulator tile for each of the 2 x NSUB / 2 tiles its partner finishes)
    constexpr size_t LDS_FIXED = LDS_IMG + (KSPLIT == 2 ? (size_t)(C / 32 * PSPLIT * 2) * ((GEO::NSUB + PSPLIT - 1) / PSPLIT) * 1024 : 0);
    // (+ the layers' biases and affines of one-board tiles, tower_param_bytes: sized by the network's depth)
    constexpr bool WSOLO = []() { if constexpr (IS_SEARCH) { if constexpr (SEARCH::WIDE) return wide_solo<C, PSPLIT, KSPLIT, BOARDS>(); } return false; }();
    constexpr bool DYN_PARAMS = BOARDS == 1 || WSOLO;          // (tower_param_bytes: sized by the network's depth)
    const size_t LDS_BYTES = LDS_FIXED + (size_t)tower_param_bytes<C, BOARDS, WSOLO>(P.nblocks);
    // per (instantiation, device): the pixel -> (subtile, lane) table, a few hundred bytes that live as long as the process
    // (the table is a pure function of the template arguments); first use is serialised
    static int16_t *d_map[16] = {nullptr};
    static int d_lds[16] = {0};
    static std::mutex d_map_mu;
    int dev = 0, cus = 256;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return fail(AZG_E_INVALID_ARG, "device ordinal out of range");
    {
        std::lock_guard<std::mutex> lk(d_map_mu);
        if (!d_map[dev]) {
            int16_t map[GEO::NSUB * 16];
            if (!tower_pixmap<GEO>(map)) return fail(AZG_E_INTERNAL, "tower pixel map does not fit its subtiles");
            int16_t *d = nullptr;
            HIPCHK(hipMalloc((void **)&d, sizeof(map)));
            HIPCHK(hipMemcpy(d, map, sizeof(map), hipMemcpyHostToDevice));
            // (the LDS a workgroup may take on THIS device -- 160 KB on gfx950 --, not a constant: a device with less refuses loudly below)
            int lds_limit = 0;
            HIPCHK(hipDeviceGetAttribute(&lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
            if ((size_t)lds_limit < LDS_FIXED) return fail(AZG_E_UNSUPPORTED, "this tile shape needs more LDS per workgroup than the device has");
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tower2<H, W, BOARDS, C, PSPLIT, SEARCH, KSPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DYN_PARAMS ? (size_t)lds_limit : LDS_FIXED)));
            d_lds[dev] = lds_limit;
            d_map[dev] = d;
        }
    }
    if (LDS_BYTES > (size_t)d_lds[dev]) return fail(AZG_E_INVALID_ARG, "this tower does not fit the LDS of its tile (too many residual blocks)");
    if (occ) {
        int n = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(&k_tower2<H, W, BOARDS, C, PSPLIT, SEARCH, KSPLIT>),
                                                            C * 2 * PSPLIT * KSPLIT, LDS_BYTES));
        *occ = n > 0 ? n : 1;
    }
    if (init_only) return AZG_OK;                            // (first-use allocations must not happen inside a stream capture)
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int ntiles = (P.boards + BOARDS - 1) / BOARDS + (P.rows_per_model ? P.nmodels - 1 : 0);
    {                                                        // one LDS image, residual stream in registers, >= 2 workgroups per CU
        const int per_cu = (int)((size_t)d_lds[dev] / LDS_BYTES) > 0 ? (int)((size_t)d_lds[dev] / LDS_BYTES) : 1;
        // (search mode: every tile is its own workgroup for the whole launch -- they never synchronise, later ones just start later)
        const int grid = IS_SEARCH || ntiles < per_cu * cus ? ntiles : per_cu * cus;
#ifdef AZG_TOWER_TIMING
        static unsigned long long *dbg = nullptr; static int calls = 0;
        if (!dbg) { HIPCHK(hipMalloc((void **)&dbg, (2048 + 4096 * 8) * 8)); HIPCHK(hipMemset(dbg, 0, (2048 + 4096 * 8) * 8)); }
        TowerParams Q = P; Q.dbg = dbg;
        AZG_LAUNCH((k_tower2<H, W, BOARDS, C, PSPLIT, SEARCH, KSPLIT>), dim3(grid), dim3(C * 2 * PSPLIT * KSPLIT), LDS_BYTES, s, Q, (const int16_t *)d_map[dev], sa);
        if constexpr (IS_SEARCH) if constexpr (SEARCH::WIDE) {
            static unsigned long long w[512 * 4];
            HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipMemcpy(w, dbg + 2048 + 4096 * 4, sizeof(unsigned long long) * 4 * (grid < 512 ? grid : 512), hipMemcpyDeviceToHost));
            double ph[4] = {0, 0, 0, 0}; const int nb = grid < 512 ? grid : 512;
            for (int b = 0; b < nb; b++) for (int i = 0; i < 4; i++) ph[i] += (double)w[b * 4 + i];
            const double n = (double)nb * (sa.sims > 8 ? sa.sims - 8 : 1);
            fprintf(stderr, "wide search, cycles per simulation (mean over %d workgroups): tree %.0f tower %.0f headconv %.0f heads %.0f\n", nb, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n);
            HIPCHK(hipMemset(dbg + 2048 + 4096 * 4, 0, sizeof(unsigned long long) * 4 * 512));
            static unsigned long long hw_[512 * 8];
            HIPCHK(hipMemcpy(hw_, dbg + 2048 + 4096 * 5, sizeof(unsigned long long) * 8 * nb, hipMemcpyDeviceToHost));
            double hp[5] = {0, 0, 0, 0, 0}, hn = 0;
            for (int b = 0; b < nb; b++) { for (int i = 0; i < 5; i++) hp[i] += (double)hw_[b * 8 + i]; hn += (double)hw_[b * 8 + 5]; }
            if (hn > 0) fprintf(stderr, "  helper wavefront (cycles per simulation with a policy backup): header %.0f masks %.0f logits %.0f softmax %.0f priors %.0f\n",
                                hp[0] / hn, hp[1] / hn, hp[2] / hn, hp[3] / hn, hp[4] / hn);
            HIPCHK(hipMemset(dbg + 2048 + 4096 * 5, 0, sizeof(unsigned long long) * 8 * 512));
            {                                                       // the layers of workgroup 0's last simulation (all wavefronts of the workgroup)
                unsigned long long h[64 * 4 * 5];
                HIPCHK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
                for (int l = 0; l <= 2 * P.nblocks; l++) for (int w = 0; w < C / 32 * PSPLIT * KSPLIT && w < 4; w++) {
                    unsigned long long *t = h + (l * 4 + w) * 5;
                    fprintf(stderr, "  layer %2d wave %d: main %6llu wait %6llu epi %6llu bar %6llu | start %llu\n", l, w, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[0] - h[0]);
                }
            }
            calls = 100;
        }
        if (++calls == 8) {
            unsigned long long h[64 * 4 * 5];
            HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            for (int l = 0; l <= 2 * P.nblocks; l++) for (int w = 0; w < C / 32; w++) {
                unsigned long long *t = h + (l * 4 + w) * 5;
                fprintf(stderr, "layer %2d wave %d: main %6llu wait %6llu epi %6llu bar %6llu | start %llu\n", l, w, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[0] - h[0]);
            }
            static unsigned long long w[4096 * 8];
            HIPCHK(hipMemcpy(w, dbg + 2048, sizeof(unsigned long long) * 8 * grid, hipMemcpyDeviceToHost));
            unsigned long long x2[1024];
            HIPCHK(hipMemcpy(x2, dbg + 1024, sizeof(x2), hipMemcpyDeviceToHost));
            for (int b = 0; b < grid && b < 512; b++) fprintf(stderr, "wgx %4d xload+store %llu weights %llu\n", b, x2[b * 2] - w[b * 8], x2[b * 2 + 1] - x2[b * 2]);
            for (int b = 0; b < grid; b++) fprintf(stderr, "wg %4d xcc %llu hwid %08llx start %llu end %llu | prologue %llu layers %llu headmm %llu softmax %llu tail %llu\n", b, w[b * 8 + 3] & 15, w[b * 8 + 2], w[b * 8], w[b * 8 + 1],
                                                   w[b * 8 + 4] - w[b * 8], w[b * 8 + 5] - w[b * 8 + 4], w[b * 8 + 6] - w[b * 8 + 5], w[b * 8 + 7] - w[b * 8 + 6], w[b * 8 + 1] - w[b * 8 + 7]);
        }
#else
        AZG_LAUNCH((k_tower2<H, W, BOARDS, C, PSPLIT, SEARCH, KSPLIT>), dim3(grid), dim3(C * 2 * PSPLIT * KSPLIT), LDS_BYTES, s, P, (const int16_t *)d_map[dev], sa);
#endif
    }
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

// ---- timing hooks for the network launches (no engine handle there: process-wide, same event scheme as azg_profile_*) ----------
struct NetProf {
    std::mutex mu; bool on = false;
    std::vector<EvPair> ev[3], pool;
    double ms[3] = {0, 0, 0}; int64_t n[3] = {0, 0, 0};
};
static NetProf g_netprof;
static bool netprof_begin(hipStream_t s, EvPair &p) {
    std::lock_guard<std::mutex> lk(g_netprof.mu);
    if (!g_netprof.on) return false;
    if (!g_netprof.pool.empty()) { p = g_netprof.pool.back(); g_netprof.pool.pop_back(); }
    else { (void)hipEventCreate(&p.a); (void)hipEventCreate(&p.b); }
    p.ext = true; g_kev = &p; (void)s;
    return true;
}
static void netprof_end(hipStream_t s, int fam, bool on, EvPair &p) {
    if (!on) return;
    g_kev = nullptr; (void)s;
    std::lock_guard<std::mutex> lk(g_netprof.mu);
    g_netprof.ev[fam].push_back(p); g_netprof.n[fam]++;
}
static void netprof_drain() {
    for (int f = 0; f < 3; f++) {
        for (auto &p : g_netprof.ev[f]) {
            (void)hipEventSynchronize(p.b);
            float t = 0; (void)hipEventElapsedTime(&t, p.a, p.b);
            g_netprof.ms[f] += t; g_netprof.pool.push_back(p);
        }
        g_netprof.ev[f].clear();
    }
}
extern "C" int azg_profile_net_enable(int on) {
    std::lock_guard<std::mutex> lk(g_netprof.mu);
    netprof_drain();
    g_netprof.on = on != 0;
    if (on) for (int f = 0; f < 3; f++) { g_netprof.ms[f] = 0; g_netprof.n[f] = 0; }
    return AZG_OK;
}
extern "C" int azg_profile_net_read(double *ms3, int64_t *launches3) {
    if (!ms3 || !launches3) return fail(AZG_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(g_netprof.mu);
    netprof_drain();
    for (int f = 0; f < 3; f++) { ms3[f] = g_netprof.ms[f]; launches3[f] = g_netprof.n[f]; }
    return AZG_OK;
}

static int device_cus(int *cus) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (*cus <= 0) *cus = 1;
    return AZG_OK;
}

// Boards per workgroup tile: the big tile has the least MFMA padding, small ones fill the chip at small batches (the arena,
// the single-tree API, brandubh's 512 games per GPU).  Tuning builds (hipcc -DAZG_TUNING, tools/sweep_small.py) let the
// environment override the choice: AZG_TOWER_BOARDS, AZG_TOWER_PSPLIT; the product library reads no environment.
static int dispatch_tower(hipStream_t s, int game, int channels, const TowerParams &P) {
#ifdef AZG_TUNING
    static const int forced = getenv("AZG_TOWER_BOARDS") ? atoi(getenv("AZG_TOWER_BOARDS")) : 0;
    static const int psplit = getenv("AZG_TOWER_PSPLIT") ? atoi(getenv("AZG_TOWER_PSPLIT")) : 0;
#else
    constexpr int forced = 0, psplit = 0;
#endif
    const int n = P.boards;
    // (tile thresholds in boards per CU of THIS device -- measured on 256 CUs: 640 / 1280 boards = 2.5 / 5 per CU ... -- not in boards)
    int cus = 1;
    { const int r = device_cus(&cus); if (r != AZG_OK) return r; }
    if (game == AZG_GAME_CONNECT4 && channels == 128) {
        const int bt = forced ? forced : 2 * n <= 5 * cus ? 1 : n <= 5 * cus ? 2 : 4;
        if (bt == 1 && psplit == 2) return launch_tower<C4::H, C4::W, 1, 128, 2>(s, P);   // 8 waves, 2 + 1 pixel subtiles: measured slower (101 vs 69 us)
        if (bt == 1) return launch_tower<C4::H, C4::W, 1, 128>(s, P);
#ifdef AZG_TUNING
        if (bt == 2 && psplit == 2) return launch_tower<C4::H, C4::W, 2, 128, 2>(s, P);   // (sweep only: profiles/r03_arena_tile_sweep.txt)
#endif
        if (bt == 2) return launch_tower<C4::H, C4::W, 2, 128>(s, P);
        return launch_tower<C4::H, C4::W, 4, 128>(s, P);
    }
    if (game == AZG_GAME_CONNECT4 && channels == 64) return launch_tower<C4::H, C4::W, 4, 64>(s, P);
    if (game == AZG_GAME_CONNECT4 && channels == 32) {           // the default net of Coach.py:108-116 (BASELINE config 1): one cout group,
        const int bt = forced ? forced : n <= 4 * cus ? 2 : 4;   // the tile's pixel subtiles dealt to two waves
        if (bt == 2) return launch_tower<C4::H, C4::W, 2, 32, 2>(s, P);
        return launch_tower<C4::H, C4::W, 4, 32, 2>(s, P);
    }
    if (game == AZG_GAME_BRANDUBH && channels == 64) {           // two cout groups: split the pixels too at small batches
        // measured (us per evaluation incl. heads, 256 / 512 / 1024 / 2048 boards): 1 board, no split 39 / 47 / 66 / 107;
        // 1 board, split 35 / 46 / 80 / 113; 2 boards, split 39 / 43 / 63 / 113
        // round 3, the k-split 1-board tile (`sp == 3`): 28 / 36 / 64 / 114 -- the shape up to 512 boards
        const int bt = forced ? forced : n <= 2 * cus ? 1 : 2;
        const int sp = psplit ? psplit : n <= 2 * cus ? 3 : n <= 4 * cus ? 2 : 1;
        if (bt == 1 && sp == 3) return launch_tower<BR::H, BR::W, 1, 64, 1, NoSearch, 2>(s, P);   // k-split: 4 waves = (cout group, k group)
        if (bt == 1 && sp == 2) return launch_tower<BR::H, BR::W, 1, 64, 2>(s, P);
        if (bt == 1) return launch_tower<BR::H, BR::W, 1, 64>(s, P);
        if (sp == 2) return launch_tower<BR::H, BR::W, 2, 64, 2>(s, P);
        return launch_tower<BR::H, BR::W, 2, 64>(s, P);
    }
    if (game == AZG_GAME_BRANDUBH && channels == 128) return launch_tower<BR::H, BR::W, 2, 128>(s, P);
    if (game == AZG_GAME_TRIMOK && channels == 32) {             // one cout group
        const int bt = forced ? forced : n <= 8 * cus ? 2 : 5;
        const int sp = psplit ? psplit : 2;                      // (256 boards: 26 us unsplit, 23 us split in two)
        if (bt == 2 && sp == 4) return launch_tower<TM::H, TM::W, 2, 32, 4>(s, P);
        if (bt == 2 && sp == 2) return launch_tower<TM::H, TM::W, 2, 32, 2>(s, P);
        if (bt == 2) return launch_tower<TM::H, TM::W, 2, 32>(s, P);
        return launch_tower<TM::H, TM::W, 5, 32>(s, P);
    }
    return fail(AZG_E_UNSUPPORTED, "no MFMA tower for this game / channel count (supported: connect4 x {32,64,128}, brandubh x {64,128}, trimok x 32 channels)");
}

// [boards, C, H*W] f32 planes (what GameState.observation / the reference's batch tensors hold) -> the tower's input rows [boards * H*W][8] fp16
__global__ __launch_bounds__(256) void k_obs_to_nhwc8(const float *src, half8 *dst, int n, int C, int HW) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = i / HW, pos = i - b * HW;
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < C; c++) v[c] = (_Float16)src[((size_t)b * C + c) * HW + pos];
    dst[i] = v;
}
extern "C" int azg_obs_to_nhwc8_f16(void *stream, const float *obs, int boards, int channels, int hw, void *x) {
    if (!obs || !x || boards <= 0 || channels <= 0 || channels > 8 || hw <= 0) return fail(AZG_E_INVALID_ARG, "null or out-of-range argument (channels <= 8)");
    const int n = boards * hw;
    hipLaunchKernelGGL(k_obs_to_nhwc8, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, obs, (half8 *)x, n, channels, hw);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_resnet_tower_f16(void *stream, int game, const void *x, const void *w, const float *bias, const float *pre_scale,
                                    const float *pre_shift, void *y, int boards, int nblocks, int channels) {
    if (!x || !w || !bias || !y || boards <= 0 || nblocks < 0) return fail(AZG_E_INVALID_ARG, "null argument");
    if (nblocks > 0 && (!pre_scale || !pre_shift)) return fail(AZG_E_INVALID_ARG, "pre_scale/pre_shift required");
    TowerParams P{x, w, bias, pre_scale, pre_shift, y, boards, nblocks, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, {}};
    EvPair ep; const bool prof = netprof_begin((hipStream_t)stream, ep);
    const int r = dispatch_tower((hipStream_t)stream, game, channels, P);
    netprof_end((hipStream_t)stream, 0, prof, ep);
    return r;
}

extern "C" int azg_resnet_tower_features_f16(void *stream, int game, const void *x, const void *w, const float *bias, const float *pre_scale,
                                             const float *pre_shift, int boards, int nblocks, int channels, const void *head1_w, const float *head1_b,
                                             void *feat, int feat_k) {
    if (!x || !w || !bias || !head1_w || !head1_b || !feat || boards <= 0 || nblocks < 0) return fail(AZG_E_INVALID_ARG, "null argument");
    if (nblocks > 0 && (!pre_scale || !pre_shift)) return fail(AZG_E_INVALID_ARG, "pre_scale/pre_shift required");
    if (feat_k <= 0 || (feat_k & 31)) return fail(AZG_E_INVALID_ARG, "feat_k must be a positive multiple of 32");
    TowerParams P{x, w, bias, pre_scale, pre_shift, nullptr, boards, nblocks, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, head1_w, head1_b, feat, feat_k,
                  nullptr, 0, {}};
    EvPair ep; const bool prof = netprof_begin((hipStream_t)stream, ep);
    const int r = dispatch_tower((hipStream_t)stream, game, channels, P);
    netprof_end((hipStream_t)stream, 0, prof, ep);
    return r;
}

extern "C" int azg_resnet_policy_value_f16(void *stream, int game, const void *x, const void *w, const float *bias, const float *pre_scale,
                                           const float *pre_shift, int boards, int nblocks, const void *head_w, const float *head_b,
                                           int A, int NV, float *policy, float *value) {
    if (!x || !w || !bias || !head_w || !head_b || !policy || !value || boards <= 0 || nblocks < 0) return fail(AZG_E_INVALID_ARG, "null argument");
    if (A <= 0 || NV <= 0 || A + NV > 16) return fail(AZG_E_UNSUPPORTED, "fused heads need A + NV <= 16");
    if (nblocks > 0 && (!pre_scale || !pre_shift)) return fail(AZG_E_INVALID_ARG, "pre_scale/pre_shift required");
    TowerParams P{x, w, bias, pre_scale, pre_shift, nullptr, boards, nblocks, head_w, head_b, policy, value, A, NV, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, {}};
    EvPair ep; const bool prof = netprof_begin((hipStream_t)stream, ep);
    const int r = dispatch_tower((hipStream_t)stream, game, 128, P);
    netprof_end((hipStream_t)stream, 0, prof, ep);
    return r;
}

extern "C" int azg_resnet_policy_value_multi_f16(void *stream, int game, const void *x, int nmodels, const void *const *w, const float *const *bias,
                                                 const float *const *pre_scale, const float *const *pre_shift, int max_boards, int nblocks,
                                                 const void *const *head_w, const float *const *head_b, int A, int NV, float *policy, float *value,
                                                 const int32_t *rows_per_model) {
    if (!x || !w || !bias || !head_w || !head_b || !policy || !value || !rows_per_model || max_boards <= 0 || nblocks < 0)
        return fail(AZG_E_INVALID_ARG, "null or out-of-range argument");
    if (nmodels < 1 || nmodels > 4) return fail(AZG_E_UNSUPPORTED, "1 to 4 models per launch");
    if (A <= 0 || NV <= 0 || A + NV > 16) return fail(AZG_E_UNSUPPORTED, "fused heads need A + NV <= 16");
    if (nblocks > 0 && (!pre_scale || !pre_shift)) return fail(AZG_E_INVALID_ARG, "pre_scale/pre_shift required");
    for (int m = 0; m < nmodels; m++)
        if (!w[m] || !bias[m] || !head_w[m] || !head_b[m] || (nblocks > 0 && (!pre_scale[m] || !pre_shift[m]))) return fail(AZG_E_INVALID_ARG, "null model parameter");
    // the grid is sized for max_boards rows plus one partial tile per extra model
    TowerParams P{x, w[0], bias[0], nblocks ? pre_scale[0] : nullptr, nblocks ? pre_shift[0] : nullptr, nullptr, max_boards, nblocks,
                  head_w[0], head_b[0], policy, value, A, NV, nullptr, nullptr, nullptr, nullptr, 0, rows_per_model, nmodels, {}};
    for (int m = 1; m < nmodels; m++)
        P.alt[m - 1] = TowerParams::Model{w[m], bias[m], nblocks ? pre_scale[m] : nullptr, nblocks ? pre_shift[m] : nullptr, head_w[m], head_b[m]};
    EvPair ep; const bool prof = netprof_begin((hipStream_t)stream, ep);
    const int r = dispatch_tower((hipStream_t)stream, game, 128, P);
    netprof_end((hipStream_t)stream, 0, prof, ep);
    return r;
}

extern "C" int azg_search_f16(azg_engine *e, void *stream, const void *w, const float *bias, const float *pre_scale, const float *pre_shift,
                              int nblocks, const void *head_w, const float *head_b, int sims) {
    if (!e || !w || !bias || !head_w || !head_b || nblocks < 0 || sims < 0) return fail(AZG_E_INVALID_ARG, "null or out-of-range argument");
    if (nblocks > 0 && (!pre_scale || !pre_shift)) return fail(AZG_E_INVALID_ARG, "pre_scale/pre_shift required");
    if (e->cfg.game != AZG_GAME_CONNECT4 || e->v.arena)
        return fail(AZG_E_UNSUPPORTED, "the fused search kernel is built for connect4 self-play with a 128-channel tower (use azg_select / network / azg_backup)");
    const int A = e->gi.action_size, NV = e->gi.num_players + 1;
    TowerParams P{nullptr, w, bias, pre_scale, pre_shift, nullptr, e->v.B, nblocks, head_w, head_b, nullptr, nullptr, A, NV, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, {}};
    SearchArgs<C4> sa{e->v, sims};
    EvPair ep; const bool prof = sims > 0 && netprof_begin((hipStream_t)stream, ep);
    // games per workgroup like the stand-alone tower's tile (dispatch_tower): small engines -- the single-tree MCTS class, config 1's 32 games --
    // take one game per workgroup (a lone 4-board tile runs at the tower's latency for four boards: 170 us per simulation against ~60)
    int r, cus = 1;
    r = device_cus(&cus); if (r != AZG_OK) { g_kev = nullptr; return r; }
    if (2 * e->v.B <= 5 * cus) r = launch_tower<C4::H, C4::W, 1, 128, 1, SearchArgs<C4>>((hipStream_t)stream, P, sa, sims == 0);   // sims == 0: set up only
    else if (e->v.B <= 5 * cus) r = launch_tower<C4::H, C4::W, 2, 128, 1, SearchArgs<C4>>((hipStream_t)stream, P, sa, sims == 0);
    else r = launch_tower<C4::H, C4::W, 4, 128, 1, SearchArgs<C4>>((hipStream_t)stream, P, sa, sims == 0);
    netprof_end((hipStream_t)stream, 2, prof, ep);
    return r;
}

extern "C" int azg_search_arena_f16(azg_engine *e, void *stream, int nmodels, const void *const *w, const float *const *bias, const float *const *pre_scale,
                                    const float *const *pre_shift, int nblocks, const void *const *head_w, const float *const *head_b,
                                    const int32_t *p2i_host, const uint32_t *seat_of_slot, int sims) {
    if (!e || !w || !bias || !head_w || !head_b || nblocks < 0 || sims < 0 || (!p2i_host && !seat_of_slot)) return fail(AZG_E_INVALID_ARG, "null or out-of-range argument");
    if (e->cfg.game != AZG_GAME_CONNECT4 || !e->v.arena)
        return fail(AZG_E_UNSUPPORTED, "the persistent arena launch is built for connect4 arena engines with 128-channel towers (use azg_select / network / azg_backup)");
    if (nmodels < 1 || nmodels > 4 || nmodels < e->gi.num_players) return fail(AZG_E_UNSUPPORTED, "one model per player, at most 4");
    for (int m = 0; m < nmodels; m++)
        if (!w[m] || !bias[m] || !head_w[m] || !head_b[m] || (nblocks > 0 && (!pre_scale || !pre_shift || !pre_scale[m] || !pre_shift[m]))) return fail(AZG_E_INVALID_ARG, "null model parameter");
    const int A = e->gi.action_size, NV = e->gi.num_players + 1;
    TowerParams P{nullptr, w[0], bias[0], nblocks ? pre_scale[0] : nullptr, nblocks ? pre_shift[0] : nullptr, nullptr, e->v.B, nblocks,
                  head_w[0], head_b[0], nullptr, nullptr, A, NV, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nmodels, {}};
    for (int m = 1; m < nmodels; m++)
        P.alt[m - 1] = TowerParams::Model{w[m], bias[m], nblocks ? pre_scale[m] : nullptr, nblocks ? pre_shift[m] : nullptr, head_w[m], head_b[m]};
    SearchArena<C4> sa{e->v, sims, SeatMap{}, seat_of_slot};
    if (p2i_host) for (int i = 0; i < e->gi.num_players && i < 8; i++) {
        if (p2i_host[i] < 0 || p2i_host[i] >= nmodels) return fail(AZG_E_INVALID_ARG, "player_to_index entry out of range");
        sa.seat.v[i] = p2i_host[i];
    }
    EvPair ep; const bool prof = sims > 0 && netprof_begin((hipStream_t)stream, ep);
    const int r = launch_tower<C4::H, C4::W, 1, 128, 1, SearchArena<C4>>((hipStream_t)stream, P, sa, sims == 0);   // sims == 0: set up only
    netprof_end((hipStream_t)stream, 2, prof, ep);
    return r;
}

// ---- the persistent wide-head search launch: tiles, the device-derived tile model, the set-up autotune ----------------------------------
#ifdef AZG_TUNING
static int wide_forced_tile() { static const int f = getenv("AZG_WIDE_BOARDS") ? atoi(getenv("AZG_WIDE_BOARDS")) : 0; return f; }
#else
static constexpr int wide_forced_tile() { return 0; }
#endif

// ONE tile shape of the launch (bt games per workgroup) for (game, tower width); AZG_E_UNSUPPORTED: no such tile.  sparse heads (EXACT =
// false: hd) or full-width heads (EXACT = true: hf).  init: one-time set-up only.  occ: see launch_tower.
template <bool EXACT>
static int wide_tile_launch(azg_engine *e, hipStream_t s, const TowerParams &P, int channels, int bt, const HeadRows &hd, const HeadsFull &hf, int sims,
                            bool init, int *occ) {
    const int game = e->cfg.game;
    if (game == AZG_GAME_BRANDUBH && channels == 64) {
        // two workgroups of four wavefronts per CU in every shape
        using SW = SearchWide<BR, 2, EXACT>;
        const SW sa{e->v, sims, hd, hf};
        if (bt == 1) return launch_tower<BR::H, BR::W, 1, 64, 1, SW, 2>(s, P, sa, init, occ);      // four wavefronts per game (walk, priors, masks, rules), k-split tower
        if (bt == 2) return launch_tower<BR::H, BR::W, 2, 64, 2, SW>(s, P, sa, init, occ);         // walker + helper per game
        if (bt == 3) return launch_tower<BR::H, BR::W, 3, 64, 2, SW>(s, P, sa, init, occ);         // solo tree phase: one wavefront per game
        if (bt == 4) return launch_tower<BR::H, BR::W, 4, 64, 2, SW>(s, P, sa, init, occ);         // (two 4-game workgroups fill a CU's LDS with a 4-block tower's
                                                                                                    //  parameters beside them: a deeper tower does not fit -> AZG_E_INVALID_ARG)
#ifdef AZG_TUNING
        if (bt == 8) return launch_tower<BR::H, BR::W, 8, 64, 4, SearchWide<BR, 1, EXACT>>(s, P, SearchWide<BR, 1, EXACT>{e->v, sims, hd, hf}, init, occ);   // one workgroup of eight wavefronts per CU
        if (bt == 12) return launch_tower<BR::H, BR::W, 2, 64, 2, SearchWide<BR, 1, EXACT>, 2>(s, P, SearchWide<BR, 1, EXACT>{e->v, sims, hd, hf}, init, occ);  // 2 games, 8 wavefronts (k-split), one workgroup per CU
        if (bt == 14) return launch_tower<BR::H, BR::W, 4, 64, 2, SearchWide<BR, 1, EXACT>>(s, P, SearchWide<BR, 1, EXACT>{e->v, sims, hd, hf}, init, occ);  // (what the spills cost: the 4-board tile with the whole register file)
#endif
    } else if (game == AZG_GAME_TRIMOK && channels == 32) {
        if (bt == 1) return launch_tower<TM::H, TM::W, 1, 32, 2, SearchWide<TM, 1, EXACT>>(s, P, SearchWide<TM, 1, EXACT>{e->v, sims, hd, hf}, init, occ);
        if (bt == 2) return launch_tower<TM::H, TM::W, 2, 32, 4, SearchWide<TM, 2, EXACT>>(s, P, SearchWide<TM, 2, EXACT>{e->v, sims, hd, hf}, init, occ);   // walker + helper per game
#ifdef AZG_TUNING
        if (bt == 4) return launch_tower<TM::H, TM::W, 4, 32, 4, SearchWide<TM, 2, EXACT>>(s, P, SearchWide<TM, 2, EXACT>{e->v, sims, hd, hf}, init, occ);
#endif
    } else if (game == AZG_GAME_CONNECT4 && channels == 32) {
        // the reference's DEFAULT net (Coach.py:108-116: 32 channels x 4 blocks, 16 + 16 head channels -- BASELINE config 1's network and what an
        // unconfigured Coach trains) on connect4: factorised heads, so the wide search mode; tiles like the 3-player env's 32-channel tower
        if (bt == 1) return launch_tower<C4::H, C4::W, 1, 32, 2, SearchWide<C4, 1, EXACT>>(s, P, SearchWide<C4, 1, EXACT>{e->v, sims, hd, hf}, init, occ);
        if (bt == 2) return launch_tower<C4::H, C4::W, 2, 32, 4, SearchWide<C4, 2, EXACT>>(s, P, SearchWide<C4, 2, EXACT>{e->v, sims, hd, hf}, init, occ);   // walker + helper per game
    } else if (game == AZG_GAME_CONNECT4 && channels == 64) {
        if (bt == 1) return launch_tower<C4::H, C4::W, 1, 64, 2, SearchWide<C4, 2, EXACT>>(s, P, SearchWide<C4, 2, EXACT>{e->v, sims, hd, hf}, init, occ);   // four wavefronts per game
        if (bt == 2) return launch_tower<C4::H, C4::W, 2, 64, 2, SearchWide<C4, 2, EXACT>>(s, P, SearchWide<C4, 2, EXACT>{e->v, sims, hd, hf}, init, occ);   // walker + helper per game
    }
    return AZG_E_UNSUPPORTED;
}

static int wide_max_tile(int game, int channels) {
    if (game == AZG_GAME_BRANDUBH && channels == 64) return 4;
    if ((game == AZG_GAME_TRIMOK && channels == 32) || (game == AZG_GAME_CONNECT4 && (channels == 32 || channels == 64))) return 2;
    return 0;
}

// The tile model (used when no measurement exists: a launch that is being captured without a set-up call before it).  Nothing in it is
// a constant of one chip or one depth: a tile of t games gives n = ceil(B / t) workgroups; they run in rounds of cus x occ(t) (CU count of
// THIS device x the occupancy query of the tile's kernel with its LDS for THIS depth); a workgroup's chain per simulation is
// fixed(t) [tree phase + heads] + conv(t) x (2 nblocks + 1) [the tower], in k cycles, from the phase stamps of the measurement build
// (profiles/r05_phase_budget.json, r05_wide_tile_sweep.txt: brandubh 4-block chain exact 83 / 105 / 123 / 161, sparse 54 / 83 / 109 / 146 at
// 1 / 2 / 3 / 4 games per workgroup, of which the tower 37 / 62 / 78 / 100 -- i.e. per conv 4.1 / 6.9 / 8.7 / 11.1); a last round that
// fills at most half of the chip's slots runs without a neighbour on its CU (x 0.82: 70 vs 83 ... 124 vs 161 measured).
template <bool EXACT>
static int wide_tile_model(azg_engine *e, int channels, int nblocks, int cus, const int *occ) {
    const int game = e->cfg.game, B = e->v.B, tmax = wide_max_tile(game, channels);
    if (game != AZG_GAME_BRANDUBH) {
        // one game per workgroup up to two games per CU (3-player env, M expansions/s at 256 / 512 / 1024 games on 256 CUs: one game per
        // workgroup 20.7 / 37.4 / 39.4, two 16.3 / 31.3 / 48.0: profiles/r05_wide_tile_sweep.txt); beyond that the shared weight stream wins
        return (occ[0] > 0 && B <= 2 * cus) || tmax < 2 || occ[1] <= 0 ? 1 : 2;
    }
    static const double fixed_exact[4] = {46, 43, 45, 61}, fixed_sparse[4] = {17, 21, 31, 46}, conv[4] = {4.1, 6.9, 8.7, 11.1};
    int bt = 1; double best = -1;
    for (int t = 1; t <= tmax; t++) {
        if (occ[t - 1] <= 0) continue;                          // this tile does not fit (LDS: too deep a tower)
        const long round = (long)cus * occ[t - 1], n = (B + t - 1) / t, full = n / round, rem = n % round;
        const double chain = (EXACT ? fixed_exact : fixed_sparse)[t - 1] + conv[t - 1] * (2 * nblocks + 1);
        const double cost = full * chain + (rem == 0 ? 0 : rem * 2 <= round ? 0.82 * chain : chain);
        if (best < 0 || cost < best) { best = cost; bt = t; }
    }
    return bt;
}

// what was decided for (device, game, width, heads, engine size, depth): tile, where it came from (0 model, 1 measured at set-up, 2 forced)
struct WideTileKey { int dev, game, channels, exact, B, nblocks; bool operator<(const WideTileKey &o) const { return memcmp(this, &o, sizeof(*this)) < 0; } };
struct WideTilePick { int bt, source, occ; float us[4]; };
static std::mutex g_tile_mu;
static std::map<WideTileKey, WideTilePick> g_tile;

// Set-up autotune (sims == 0, never inside a capture): every tile shape this (game, width) has is timed ONCE on a scratch engine of the same
// size -- fresh games, a node store for the trial only -- with THIS network: a warm launch and a timed launch of 24 simulations each; the
// fastest is the engine's tile for (B, nblocks) from then on.  Tile shape changes no result (every shape is bit-identical to the
// launch-per-phase form: tests/test_gpu_fullsize.py), so the measurement only decides speed.  The model's tile is the prior: a trial has to
// beat it by more than 3 % to replace it.  Falls back to the model when the scratch engine cannot be had (memory).
template <bool EXACT>
static int wide_tile_autotune(azg_engine *e, hipStream_t s, const TowerParams &P, int channels, const HeadRows &hd, const HeadsFull &hf, const int *occ,
                              WideTilePick &pick) {
    const int tmax = wide_max_tile(e->cfg.game, channels);
    constexpr int TRIAL_SIMS = 24;                            // (8 simulations on fresh trees could not tell the 2- from the 3-game tile of an 8-block tower: 1 % apart
                                                              //  in the trial, 20 % at 40 simulations -- the solo tree phase slows down as the trees grow)
    azg_config cfg = e->cfg;
    cfg.sims_per_move = TRIAL_SIMS; cfg.nodes_per_tree = (2 * TRIAL_SIMS + 2) * e->gi.max_children + 64;
    cfg.example_capacity = 0; cfg.result_capacity = 0; cfg.temp_table = nullptr; cfg.temp_table_len = 0; cfg.arena = 0;
    azg_engine *tmp = nullptr;
    if (azg_engine_create(&cfg, &tmp) != AZG_OK || !tmp) return AZG_E_HIP;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { azg_engine_destroy(tmp); return AZG_E_HIP; }
    int best = 0; float best_ms = 0;
    for (int t = 1; t <= tmax; t++) {
        pick.us[t - 1] = 0;
        if (occ[t - 1] <= 0) continue;
        if (azg_engine_reset(tmp, s) != AZG_OK) continue;
        TowerParams Q = P; Q.boards = tmp->v.B;
        if (wide_tile_launch<EXACT>(tmp, s, Q, channels, t, hd, hf, TRIAL_SIMS, false, nullptr) != AZG_OK) continue;
        (void)hipEventRecord(a, s);
        const int r = wide_tile_launch<EXACT>(tmp, s, Q, channels, t, hd, hf, TRIAL_SIMS, false, nullptr);
        (void)hipEventRecord(b, s);
        if (r != AZG_OK || hipEventSynchronize(b) != hipSuccess) continue;
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        pick.us[t - 1] = ms * 1e3f;
        if (!best || ms < best_ms) { best = t; best_ms = ms; }
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    azg_engine_destroy(tmp);
    (void)hipGetLastError();
    if (!best) return AZG_E_HIP;
    // hysteresis: the model's tile (pick.bt on entry) stands unless the fastest trial beats ITS trial by more than 3 % -- near-ties (512
    // brandubh games: one and two games per workgroup are 1 % apart) would otherwise flip with the noise of a single timing, and under a
    // profiler, which serialises and slows the trials, the kernel a run is measured on must not depend on which pass it is
    const int prior = pick.bt;
    if (prior >= 1 && prior <= tmax && pick.us[prior - 1] > 0 && best_ms * 1e3f >= 0.97f * pick.us[prior - 1]) best = prior;
    pick.bt = best; pick.source = 1;
    return AZG_OK;
}

template <bool EXACT>
static int wide_tile_pick(azg_engine *e, hipStream_t s, const TowerParams &P, int channels, const HeadRows &hd, const HeadsFull &hf, bool setup, WideTilePick &out) {
    int dev = 0, cus = 1;
    HIPCHK(hipGetDevice(&dev));
    int r = device_cus(&cus); if (r != AZG_OK) return r;
    const WideTileKey key{dev, e->cfg.game, channels, EXACT ? 1 : 0, e->v.B, P.nblocks};
    {
        std::lock_guard<std::mutex> lk(g_tile_mu);
        auto it = g_tile.find(key);
        if (it != g_tile.end() && (it->second.source != 0 || !setup)) { out = it->second; return AZG_OK; }
    }
    WideTilePick pick{1, 0, 1, {0, 0, 0, 0}};
    const int tmax = wide_max_tile(e->cfg.game, channels);
    if (tmax == 0) return AZG_E_UNSUPPORTED;
    if (wide_forced_tile()) {
        pick.bt = wide_forced_tile(); pick.source = 2;
        r = wide_tile_launch<EXACT>(e, s, P, channels, pick.bt, hd, hf, 0, true, &pick.occ);
        if (r != AZG_OK) return r;
    } else {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cs);
        const bool capturing = cs != hipStreamCaptureStatusNone;
        int occ[4] = {0, 0, 0, 0};
        for (int t = 1; t <= tmax; t++) {                        // set up every tile this network fits; its occupancy on this device
            int o = 0;
            if (capturing) {                                     // (no first-use allocation inside a capture: a tile that was never set up is left out)
                continue;
            }
            if (wide_tile_launch<EXACT>(e, s, P, channels, t, hd, hf, 0, true, &o) == AZG_OK) occ[t - 1] = o;
        }
        if (capturing) {                                         // a captured launch without a set-up call before it: the model over what is set up
            return fail(AZG_E_INVALID_ARG, "persistent wide-head search: call it once with sims == 0 (one-time set-up) before capturing it in a graph");
        }
        bool any = false;
        for (int t = 0; t < tmax; t++) any |= occ[t] > 0;
        if (!any) return fail(AZG_E_INVALID_ARG, "this tower does not fit the LDS of any tile of the persistent launch (too many residual blocks)");
        pick.bt = wide_tile_model<EXACT>(e, channels, P.nblocks, cus, occ);
        if (setup && !e->v.perm_tape) (void)wide_tile_autotune<EXACT>(e, s, P, channels, hd, hf, occ, pick);   // (failure: the model's pick stands)
        pick.occ = occ[pick.bt - 1];
    }
    std::lock_guard<std::mutex> lk(g_tile_mu);
    g_tile[key] = pick;
    out = pick;
    return AZG_OK;
}

template <bool EXACT>
static int search_wide(azg_engine *e, void *stream, const void *w, const float *bias, const float *pre_scale, const float *pre_shift, int nblocks, int channels,
                       const void *head1_w, const float *head1_b, const HeadRows &hd, const HeadsFull &hf, int feat_k, int sims) {
    if (!e || !w || !bias || !head1_w || !head1_b || nblocks < 0 || sims < 0) return fail(AZG_E_INVALID_ARG, "null or out-of-range argument");
    if (nblocks > 0 && (!pre_scale || !pre_shift)) return fail(AZG_E_INVALID_ARG, "pre_scale/pre_shift required");
    if (e->v.arena) return fail(AZG_E_UNSUPPORTED, "the persistent search launches are built for self-play engines");
    const int A = e->gi.action_size, NV = e->gi.num_players + 1, hw = e->gi.obs_h * e->gi.obs_w;
    if (feat_k != (hw * 16 + 31) / 32 * 32) return fail(AZG_E_INVALID_ARG, "feat_k must be H*W*16 rounded up to 32");
    if (wide_max_tile(e->cfg.game, channels) == 0)
        return fail(AZG_E_UNSUPPORTED, "persistent wide-head search: brandubh x 64, the 3-player env x 32 and connect4 x {32, 64} channels (use azg_select / network / azg_backup)");
    TowerParams P{nullptr, w, bias, pre_scale, pre_shift, nullptr, e->v.B, nblocks, nullptr, nullptr, nullptr, nullptr, A, NV, nullptr, head1_w, head1_b, nullptr, feat_k,
                  nullptr, 0, {}};
    hipStream_t s = (hipStream_t)stream;
    // Games per workgroup by the engine's size (SelfPlayAgent.pyx:23-26: the batch is whatever the caller made it): measured at set-up
    // (sims == 0) on this device with this network, else the device-derived model above.  More boards per tile share every weight fragment
    // between the tile's boards, pad fewer pixel lanes (brandubh: 196 of 208 instead of 49 of 64) and have a main loop long enough to
    // amortise a layer's epilogue and barriers; fewer fill the chip at small engines.
    WideTilePick pick;
    int r = wide_tile_pick<EXACT>(e, s, P, channels, hd, hf, sims == 0, pick);
    if (r != AZG_OK) return r;
    if (sims == 0) return AZG_OK;                            // one-time set-up only
    EvPair ep; const bool prof = netprof_begin(s, ep);
    r = wide_tile_launch<EXACT>(e, s, P, channels, pick.bt, hd, hf, sims, false, nullptr);
    if (r == AZG_E_UNSUPPORTED) { g_kev = nullptr; return fail(r, "persistent wide-head search: no such tile for this game / width"); }
    netprof_end(s, 2, prof, ep);
    return r;
}

// the tile a wide-head engine searches with: info12 = {games per workgroup, workgroups of a launch, workgroups a CU holds at once, CUs of the
// device, source (0 device-derived model, 1 measured at set-up, 2 forced by a tuning build), trial simulations, 0, 0, then the set-up
// measurement in ns per trial launch for 1 / 2 / 3 / 4 games per workgroup (0: not measured)}; AZG_E_INVALID_ARG before the first call of the
// launch for this (engine size, depth)
extern "C" int azg_search_wide_tile_info(azg_engine *e, int channels, int nblocks, int exact, int32_t *info12) {
    int32_t *info8 = info12;
    if (!e || !info8) return fail(AZG_E_INVALID_ARG, "null argument");
    int dev = 0, cus = 1;
    HIPCHK(hipGetDevice(&dev));
    int r = device_cus(&cus); if (r != AZG_OK) return r;
    std::lock_guard<std::mutex> lk(g_tile_mu);
    auto it = g_tile.find(WideTileKey{dev, e->cfg.game, channels, exact ? 1 : 0, e->v.B, nblocks});
    if (it == g_tile.end()) return fail(AZG_E_INVALID_ARG, "no persistent wide-head launch has been set up for this engine size and depth");
    const WideTilePick &p = it->second;
    info8[0] = p.bt; info8[1] = (e->v.B + p.bt - 1) / p.bt; info8[2] = p.occ; info8[3] = cus; info8[4] = p.source;
    info8[5] = 24; info8[6] = info8[7] = 0;
    for (int t = 0; t < 4; t++) info8[8 + t] = (int32_t)(p.us[t] * 1e3f);
    return AZG_OK;
}

extern "C" int azg_search_wide_f16(azg_engine *e, void *stream, const void *w, const float *bias, const float *pre_scale, const float *pre_shift,
                                   int nblocks, int channels, const void *head1_w, const float *head1_b, const void *head_rows, const float *head_b,
                                   int feat_k, int sims) {
    if (!head_rows || !head_b) return fail(AZG_E_INVALID_ARG, "null argument");
    return search_wide<false>(e, stream, w, bias, pre_scale, pre_shift, nblocks, channels, head1_w, head1_b, HeadRows{(const _Float16 *)head_rows, head_b, feat_k},
                              HeadsFull{}, feat_k, sims);
}

extern "C" int azg_search_wide_exact_f16(azg_engine *e, void *stream, const void *w, const float *bias, const float *pre_scale, const float *pre_shift,
                                         int nblocks, int channels, const void *head1_w, const float *head1_b, const void *wps_packed, const void *wv_packed,
                                         const float *head_b, int feat_k, int sims) {
    if (!e || !wps_packed || !wv_packed || !head_b) return fail(AZG_E_INVALID_ARG, "null argument");
    return search_wide<true>(e, stream, w, bias, pre_scale, pre_shift, nblocks, channels, head1_w, head1_b, HeadRows{nullptr, head_b, feat_k},
                             HeadsFull{(const half8 *)wps_packed, (const half8 *)wv_packed, head_b}, feat_k, sims);
}

extern "C" int azg_policy_value_heads_f16(void *stream, const void *y, const void *head_w_packed, const float *head_b, int boards, int k,
                                          int A, int NV, float *logits_ws, float *policy, float *value) {
    if (!y || !head_w_packed || !head_b || !logits_ws || (!policy != !value)) return fail(AZG_E_INVALID_ARG, "null argument");
    if (boards <= 0 || k <= 0 || (k & 31) || A <= 0 || A > 1024 || NV <= 0 || NV > 64) return fail(AZG_E_INVALID_ARG, "boards > 0, k a multiple of 32, 0 < A <= 1024, 0 < NV <= 64");
    const int osub = (A + NV + 15) / 16, nchunks = (osub + HEAD_NS - 1) / HEAD_NS, groups = (boards + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    EvPair ep; const bool prof = netprof_begin(s, ep);
    AZG_LAUNCH(k_heads, dim3(groups * nchunks), dim3(HEAD_WAVES * 64), 0, s, (const _Float16 *)y, (const half8 *)head_w_packed, head_b, logits_ws,
                       boards, k / 32, osub);
    netprof_end(s, 1, prof, ep);
    if (policy)                                              // (NULL: leave the logits for azg_backup_select_logits)
        AZG_LAUNCH(k_heads_softmax, dim3((boards + 3) / 4), dim3(256), 0, s, (const float *)logits_ws, policy, value, boards, osub * 16, A, NV);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_policy_value_heads_fact_f16(void *stream, const void *feat, const void *wp_packed, const void *wv_packed, const float *head_b, int boards,
                                               int feat_k, int A, int NV, float *logits_ws, float *policy, float *value) {
    if (!feat || !wp_packed || !wv_packed || !head_b || !logits_ws || (!policy != !value)) return fail(AZG_E_INVALID_ARG, "null argument");
    if (boards <= 0 || feat_k <= 0 || (feat_k & 31) || A <= 0 || A > 1024 || NV <= 0 || NV > 16) return fail(AZG_E_INVALID_ARG, "boards > 0, feat_k a multiple of 32, 0 < A <= 1024, 0 < NV <= 16");
    const int osp = (A + 15) / 16, osub = (A + NV + 15) / 16, nchunks = (osp + HEADF_NS - 1) / HEADF_NS + 1, groups = (boards + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    EvPair ep; const bool prof = netprof_begin(s, ep);
    const HeadsFact hf{(const half8 *)wp_packed, (const half8 *)wv_packed, head_b, feat_k, osp, A, NV};
    AZG_LAUNCH(k_heads_fact, dim3(groups * nchunks), dim3(HEADF_Q * 64), 0, s, (const _Float16 *)feat, hf, logits_ws, boards, osub * 16);
    netprof_end(s, 1, prof, ep);
    if (policy)
        AZG_LAUNCH(k_heads_softmax, dim3((boards + 3) / 4), dim3(256), 0, s, (const float *)logits_ws, policy, value, boards, osub * 16, A, NV);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

extern "C" int azg_heads_softmax(void *stream, const float *logits, int boards, int logits_stride, int A, int NV, float *policy, float *value) {
    if (!logits || !policy || !value) return fail(AZG_E_INVALID_ARG, "null argument");
    if (boards <= 0 || A <= 0 || A > 1024 || NV <= 0 || NV > 64 || logits_stride < A + NV) return fail(AZG_E_INVALID_ARG, "boards > 0, 0 < A <= 1024, 0 < NV <= 64, stride >= A + NV");
    hipLaunchKernelGGL(k_heads_softmax, dim3((boards + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, policy, value, boards, logits_stride, A, NV);
    HIPCHK(hipGetLastError());
    return AZG_OK;
}

// halves of w_packed a tower launch may read: stem (3 k-steps) + 2 * nblocks convolutions (9 * C/32 k-steps each) + the slack the weight
// prefetch ring runs into behind the last layer.  The deepest ring is the k-split tile's (azg_conv.h conv_main2<..., WR = 9, KSTR = 2>):
// its last prefetch of the last layer is k-step 2 * (8 + 8) + 1 = 33 of an 18 k-step layer -- 16 k-steps past the end; 18 are required.
extern "C" int64_t azg_tower_weights_size(int channels, int nblocks) {
    if (channels <= 0 || (channels & 31) || nblocks < 0) return fail(AZG_E_INVALID_ARG, "channels must be a positive multiple of 32");
    const int64_t kstep = (int64_t)channels * 32;
    return ((int64_t)STEM_KSTEPS + (int64_t)2 * nblocks * 9 * (channels / 32) + AZG_TOWER_W_SLACK_KSTEPS) * kstep;
}

// host-side layout tables of the tower (no device needed): the pixel -> (subtile, lane) map and the padded LDS row of every pixel
template <int H, int W, int BOARDS, int C>
static int tower_layout_of(int16_t *map, int32_t *qrow, int32_t *info) {
    using GEO = TowerGeom<H, W, BOARDS, C>;
    if (map && !tower_pixmap<GEO>(map)) return fail(AZG_E_INTERNAL, "tower pixel map does not fit its subtiles");
    if (qrow) for (int p = 0; p < GEO::ROWS; p++) qrow[p] = GEO::qrow(p);
    info[0] = GEO::NSUB; info[1] = GEO::ROWS; info[2] = GEO::RSTRIDE; info[3] = GEO::TROWS; info[4] = GEO::TILE; info[5] = GEO::PW;
    info[6] = GEO::LEAD; info[7] = GEO::BSTRIDE;
    return AZG_OK;
}

extern "C" int azg_tower_layout(int game, int boards_per_tile, int channels, int16_t *pixmap, int32_t *qrow, int32_t *info8) {
    if (!info8) return fail(AZG_E_INVALID_ARG, "null argument");
#define AZG_LAYOUT(GM, BT, CH) if (game == GM::ID && boards_per_tile == BT && channels == CH) return tower_layout_of<GM::H, GM::W, BT, CH>(pixmap, qrow, info8)
    AZG_LAYOUT(C4, 1, 128); AZG_LAYOUT(C4, 2, 128); AZG_LAYOUT(C4, 4, 128); AZG_LAYOUT(C4, 4, 64); AZG_LAYOUT(C4, 2, 32); AZG_LAYOUT(C4, 4, 32);
    AZG_LAYOUT(BR, 1, 64); AZG_LAYOUT(BR, 2, 64); AZG_LAYOUT(BR, 2, 128);
    AZG_LAYOUT(TM, 2, 32); AZG_LAYOUT(TM, 5, 32);
#undef AZG_LAYOUT
    return fail(AZG_E_UNSUPPORTED, "no tower instantiation for this (game, boards per tile, channels)");
}

#ifdef AZG_TREE_TIMING
// measurement builds only (not part of include/azg.h): the s_memtime stamps [B][16] of every slot's last simulation
extern "C" int azg_debug_tree_timing(azg_engine *e, unsigned long long *host) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host, e->v.dbg, sizeof(unsigned long long) * 16 * (size_t)e->v.B, hipMemcpyDeviceToHost));
    return AZG_OK;
}
#endif

extern "C" int azg_profile_enable(azg_engine *e, int on) {
    if (!e) return fail(AZG_E_INVALID_ARG, "null engine");
    prof_drain(e);
    e->profile = on != 0;
    if (on) { for (int f = 0; f < 3; f++) { e->ms[f] = 0; e->launches[f] = 0; } }
    return AZG_OK;
}
extern "C" int azg_profile_read(azg_engine *e, double *ms3, int64_t *launches3) {
    if (!e || !ms3 || !launches3) return fail(AZG_E_INVALID_ARG, "null argument");
    prof_drain(e);
    for (int f = 0; f < 3; f++) { ms3[f] = e->ms[f]; launches3[f] = e->launches[f]; }
    return AZG_OK;
}
