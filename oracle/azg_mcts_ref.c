/*
 * azg_mcts_ref.c -- ORACLE restatement of alphazero/MCTS.pyx and alphazero/SelfPlayAgent.pyx.
 * TEST INFRASTRUCTURE ONLY (see azg_oracle.h).  Structure mirrors the reference: heap Node objects with a
 * (shuffled) child pointer list, a path stack, one MCTS object per game (per player in arena mode).
 * Arithmetic follows the C that Cython generates from the .pyx, which is the ground truth
 * for the mixed float/double steps.
 */
#include "azg_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXA 1024

/* ---- Node (MCTS.pyx:49-104) ------------------------------------------------------------------- */
typedef struct onode {
    struct onode **children; int nchildren;      /* _children */
    int a;                                        /* a  */
    uint8_t e[AZO_MAX_PLAYERS + 1];               /* e  */
    float q, v; int n; float p; int player;       /* q v n p player */
} onode;

static onode *node_new(int action) {              /* Node.__init__ :59-67 */
    onode *nd = (onode *)calloc(1, sizeof(onode));
    nd->a = action;
    return nd;
}
static void node_free(onode *nd) {
    if (!nd) return;
    for (int i = 0; i < nd->nchildren; i++) node_free(nd->children[i]);
    free(nd->children); free(nd);
}
static int e_any(const onode *nd, int np1) { for (int i = 0; i < np1; i++) if (nd->e[i]) return 1; return 0; }

struct azo_mcts {
    azo_mcts_args args;
    onode *root, *curnode;
    onode **path; int path_len, path_cap;
    int depth, max_depth, discount_max_depth;
    uint64_t own_ctr, *ctr;                       /* tape counter (own, or the agent slot's) */
    int32_t last_path[256]; int last_path_len;
    onode *tree_base;                             /* original root for freeing */
};

azo_mcts *azo_mcts_new(const azo_mcts_args *a) {  /* MCTS.__init__ :133-145 */
    azo_mcts *m = (azo_mcts *)calloc(1, sizeof(*m));
    m->args = *a;
    m->root = m->tree_base = node_new(-1);
    m->curnode = m->root;
    m->path_cap = 256; m->path = (onode **)malloc(sizeof(onode *) * m->path_cap);
    m->ctr = &m->own_ctr;
    return m;
}
void azo_mcts_free(azo_mcts *m) { if (!m) return; node_free(m->tree_base); free(m->path); free(m); }
void azo_mcts_reset(azo_mcts *m) {                /* :154-160 */
    node_free(m->tree_base);
    m->root = m->tree_base = node_new(-1); m->curnode = m->root;
    m->path_len = 0; m->depth = m->max_depth = m->discount_max_depth = 0;
}
uint64_t azo_mcts_tape_ctr(const azo_mcts *m) { return *m->ctr; }
void azo_mcts_set_tape_ctr(azo_mcts *m, uint64_t c) { *m->ctr = c; }
int azo_mcts_root_n(const azo_mcts *m) { return m->root->n; }
int azo_mcts_max_depth(const azo_mcts *m) { return m->max_depth; }
int azo_mcts_depth(const azo_mcts *m) { return m->depth; }

/* Node.add_children :76-79 (one stub per valid action ascending, then shuffle via the tape) */
static void add_children(azo_mcts *m, onode *nd, const uint8_t *valid, int A) {
    int idx[MAXA], k = 0; int32_t pos[MAXA];
    for (int a = 0; a < A; a++) if (valid[a]) idx[k++] = a;
    int base = nd->nchildren;                      /* list.extend: appended after existing children */
    nd->children = (onode **)realloc(nd->children, sizeof(onode *) * (size_t)(base + k + 1));
    onode **tmp = (onode **)malloc(sizeof(onode *) * (size_t)(base + k + 1));
    for (int i = 0; i < base; i++) tmp[i] = nd->children[i];
    for (int i = 0; i < k; i++) tmp[base + i] = node_new(idx[i]);
    int tot = base + k;
    azo_tape_shuffle_pos(m->args.tape_seed, m->args.tape_stream, *m->ctr, tot, pos);
    *m->ctr += (uint64_t)tot;
    for (int i = 0; i < tot; i++) nd->children[pos[i]] = tmp[i];
    nd->nchildren = tot;
    free(tmp);
}

/* Node.best_child + Node.uct :86-104 */
static onode *best_child(const onode *self, float fpu_reduction, float cpuct) {
    double seen = 0.0;                             /* python sum(): double, list order :91 */
    for (int i = 0; i < self->nchildren; i++) if (self->children[i]->n > 0) seen += (double)self->children[i]->p;
    float seen_policy = (float)seen;
    float fpu_value = (float)((double)self->v - ((double)fpu_reduction * sqrt((double)seen_policy)));   /* :92 */
    float cur_best = -INFINITY;
    float sqrt_n = (float)sqrt((double)self->n);   /* :94 */
    onode *child = NULL;
    for (int i = 0; i < self->nchildren; i++) {
        const onode *c = self->children[i];
        float t = c->n == 0 ? fpu_value : c->q;
        float uct = t + (((cpuct * c->p) * sqrt_n) / ((float)(1 + c->n)));     /* :87 gen-C */
        if (uct > cur_best) { cur_best = uct; child = self->children[i]; }
    }
    return child;
}

static void path_push(azo_mcts *m, onode *nd) {
    if (m->path_len == m->path_cap) { m->path_cap *= 2; m->path = (onode **)realloc(m->path, sizeof(onode *) * m->path_cap); }
    m->path[m->path_len++] = nd;
}

/* MCTS.find_leaf :208-228 */
int azo_mcts_find_leaf(azo_mcts *m, int game, const azo_state *gs, azo_state *leaf) {
    azo_game_info gi; azo_game_info_get(game, &gi);
    int np1 = m->args.num_players_plus_draw;
    m->depth = 0; m->curnode = m->root; *leaf = *gs;
    m->last_path_len = 0;
    while (m->curnode->n > 0 && !e_any(m->curnode, np1)) {
        path_push(m, m->curnode);
        m->curnode = best_child(m->curnode, m->args.fpu_reduction, m->args.cpuct);
        azo_game_play(game, leaf, m->curnode->a);
        if (m->last_path_len < 256) m->last_path[m->last_path_len++] = m->curnode->a;
        m->depth += 1;
    }
    if (m->depth > m->max_depth) { m->max_depth = m->depth; m->discount_max_depth = m->depth; }
    if (m->curnode->n == 0) {
        uint8_t valid[MAXA], ws[AZO_MAX_PLAYERS + 1];
        m->curnode->player = leaf->player;
        azo_game_win_state(game, leaf, ws);
        memset(m->curnode->e, 0, sizeof(m->curnode->e));
        for (int i = 0; i < gi.num_players + 1; i++) m->curnode->e[i] = ws[i];
        azo_game_valid_moves(game, leaf, valid);
        add_children(m, m->curnode, valid, gi.action_size);
        return 1;
    }
    return 0;
}

/* MCTS._get_value :291-295 */
static float get_value(const float *value, int value_size, int player, int num_players) {
    if (value_size > num_players) return value[player] + (value[num_players] / ((float)num_players));
    return value[player];
}

/* MCTS._add_root_noise :197-206 (gen-C line: c.p = (float)((c.p * (1.0 - frac)) + (frac * n))) */
static void add_root_noise(azo_mcts *m) {
    int k = m->root->nchildren;
    double noise_d[MAXA];
    azo_tape_dirichlet(m->args.tape_seed, m->args.tape_stream, *m->ctr, k, 10.83 / (double)k, noise_d);
    *m->ctr += 1;
    for (int i = 0; i < k; i++) {
        float n = (float)noise_d[i];
        onode *c = m->root->children[i];
        c->p = (float)(((double)c->p * (1.0 - (double)m->args.root_noise_frac)) + (double)(m->args.root_noise_frac * n));
    }
}

/* MCTS.process_results :230-289 */
void azo_mcts_process_results(azo_mcts *m, int game, float *value_in, float *pi, int add_noise, int add_temp) {
    azo_game_info gi; azo_game_info_get(game, &gi);
    int A = gi.action_size, P = gi.num_players, np1 = m->args.num_players_plus_draw;
    int value_size = P + 1;
    float value[AZO_MAX_PLAYERS + 1];
    onode *cur = m->curnode;
    if (e_any(cur, np1)) {
        for (int i = 0; i < np1; i++) value[i] = (float)cur->e[i];      /* :235 */
        value_size = np1;
    } else {
        for (int i = 0; i < value_size; i++) value[i] = value_in[i];
        float valids[MAXA];
        memset(valids, 0, sizeof(float) * (size_t)A);
        for (int i = 0; i < cur->nchildren; i++) valids[cur->children[i]->a] = 1.f;   /* :239-241 */
        for (int a = 0; a < A; a++) pi[a] = pi[a] * valids[a];                         /* :244 */
        float s = azo_np_sum_f32(pi, A);
        for (int a = 0; a < A; a++) pi[a] = pi[a] / s;                                 /* :245 */
        if (cur == m->root) {
            if (add_temp) {                                                            /* :249-252 */
                double ex = 1.0 / (double)m->args.root_policy_temp;
                for (int a = 0; a < A; a++) pi[a] = azo_np_pow_f32(pi[a], ex);
                float s2 = azo_np_sum_f32(pi, A);
                for (int a = 0; a < A; a++) pi[a] = pi[a] / s2;
            }
            for (int i = 0; i < cur->nchildren; i++) cur->children[i]->p = pi[cur->children[i]->a];   /* :254 */
            if (add_noise) add_root_noise(m);                                          /* :255-256 */
        } else {
            for (int i = 0; i < cur->nchildren; i++) cur->children[i]->p = pi[cur->children[i]->a];   /* :258 */
        }
    }
    int i = 0;
    while (m->path_len > 0) {                                                          /* :265-287 */
        onode *parent = m->path[--m->path_len];
        float v = get_value(value, value_size, parent->player, P);
        float discount = (float)pow((double)m->args.min_discount, (double)(i / m->discount_max_depth)); /* :270 cdivision */
        if ((double)v < 0.5) discount = (float)(2.0 - (double)discount);
        else if ((double)v == 0.5) discount = 1.0f;
        cur->q = (((cur->q * (float)cur->n) + (v * discount)) / ((float)(cur->n + 1)));            /* :282 */
        if (cur->n == 0) cur->v = get_value(value, value_size, cur->player, P);                     /* :283-284 */
        cur->n += 1;
        cur = parent;
        i += 1;
    }
    m->curnode = cur;
    m->root->n += 1;                                                                                /* :289 */
}

/* MCTS.update_root :185-195 */
int azo_mcts_update_root(azo_mcts *m, int game, const azo_state *gs, int a) {
    azo_game_info gi; azo_game_info_get(game, &gi);
    if (m->root->nchildren == 0) {
        uint8_t valid[MAXA];
        azo_game_valid_moves(game, gs, valid);
        add_children(m, m->root, valid, gi.action_size);
    }
    for (int i = 0; i < m->root->nchildren; i++)
        if (m->root->children[i]->a == a) { m->root = m->root->children[i]; return 0; }
    return -1;
}

/* MCTS.counts :297-303 */
void azo_mcts_counts(const azo_mcts *m, int game, int32_t *counts) {
    azo_game_info gi; azo_game_info_get(game, &gi);
    memset(counts, 0, sizeof(int32_t) * (size_t)gi.action_size);
    for (int i = 0; i < m->root->nchildren; i++) counts[m->root->children[i]->a] = m->root->children[i]->n;
}

static int argmax_f32(const float *x, int n) { int b = 0; for (int i = 1; i < n; i++) if (x[i] > x[b]) b = i; return b; }

/* MCTS.probs :308-329 */
void azo_mcts_probs(const azo_mcts *m, int game, float temp, float *probs) {
    azo_game_info gi; azo_game_info_get(game, &gi);
    int A = gi.action_size; int32_t ci[MAXA]; float counts[MAXA];
    azo_mcts_counts(m, game, ci);
    for (int a = 0; a < A; a++) counts[a] = (float)ci[a];
    if (temp == 0) {
        int b = argmax_f32(counts, A);
        for (int a = 0; a < A; a++) probs[a] = 0.f;
        probs[b] = 1.f; return;
    }
    float s = azo_np_sum_f32(counts, A);
    double ex = 1.0 / (double)temp;
    for (int a = 0; a < A; a++) probs[a] = azo_np_pow_f32(counts[a] / s, ex);
    float s2 = azo_np_sum_f32(probs, A);
    for (int a = 0; a < A; a++) probs[a] = probs[a] / s2;
}

/* MCTS.value :331-344 */
float azo_mcts_value(const azo_mcts *m, int average) {
    float value = 0;
    if (average) {
        double s = 0.0;
        for (int i = 0; i < m->root->nchildren; i++) if (m->root->children[i]->n > 0) s += (double)m->root->children[i]->q;
        value = (float)(s / (double)m->root->nchildren);
    } else {
        for (int i = 0; i < m->root->nchildren; i++) {
            const onode *c = m->root->children[i];
            if (c->q > value && c->n > 0) value = c->q;
        }
    }
    return value;
}

/* MCTS.raw_search :175-183 */
void azo_mcts_raw_search(azo_mcts *m, int game, const azo_state *gs, int sims, int noise, int temp) {
    azo_game_info gi; azo_game_info_get(game, &gi);
    float v[AZO_MAX_PLAYERS + 1], p[MAXA]; azo_state leaf;
    m->max_depth = 0;
    for (int s = 0; s < sims; s++) {
        /* `pi *= valids` on a typed memoryview rebinds pi to a NEW array (memoryview has no __imul__, numpy's
           __rmul__ allocates), so the caller's p/v buffers are never modified: refill them every iteration */
        for (int i = 0; i < gi.num_players + 1; i++) v[i] = 0.f;
        for (int a = 0; a < gi.action_size; a++) p[a] = 1.f;
        azo_mcts_find_leaf(m, game, gs, &leaf);
        azo_mcts_process_results(m, game, v, p, noise, temp);
    }
}

int azo_mcts_root_children(const azo_mcts *m, int32_t *a, int32_t *n, float *q, float *p, float *v) {
    for (int i = 0; i < m->root->nchildren; i++) {
        const onode *c = m->root->children[i];
        a[i] = c->a; n[i] = c->n; q[i] = c->q; p[i] = c->p; v[i] = c->v;
    }
    return m->root->nchildren;
}
int azo_mcts_last_path(const azo_mcts *m, int32_t *actions) {
    for (int i = 0; i < m->last_path_len; i++) actions[i] = m->last_path[i];
    return m->last_path_len;
}
void azo_mcts_root_header(const azo_mcts *m, int32_t *n, float *q, float *v, int32_t *player, uint8_t *e) {
    *n = m->root->n; *q = m->root->q; *v = m->root->v; *player = m->root->player;
    for (int i = 0; i < AZO_MAX_PLAYERS + 1; i++) e[i] = m->root->e[i];
}

/* ================================================================================================
 * SelfPlayAgent (alphazero/SelfPlayAgent.pyx)
 * ================================================================================================ */
typedef struct { azo_state st; float *pi; } hist_entry;

struct azo_agent {
    int game; azo_game_info gi; azo_agent_args args;
    int B, P, A, O;
    azo_state *games;
    hist_entry **hist; int *hist_len, *hist_cap;
    float *temps; int *next_reset;
    azo_mcts **mcts;                 /* [B * (is_arena ? P : 1)] */
    uint64_t *slot_ctr;              /* per-slot tape counters */
    uint64_t agent_ctr;              /* agent-level stream (fast coin, seat shuffle) */
    int fast;
    int32_t player_to_index[AZO_MAX_PLAYERS];
    int32_t *batch_indices;          /* row -> game (arena) */
    int games_played;
    /* output_queue */
    float *s_obs, *s_pi, *s_z; int n_samples, cap_samples;
    /* result_queue */
    uint8_t *r_ws; int32_t *r_turns, *r_slot; int n_results, cap_results;
    int32_t *last_actions;
    uint64_t sims_done, expansions;
};

#define AGENT_STREAM(ag) (0x4000000000000000ULL + (ag)->args.slot_base)

static azo_mcts *agent_new_mcts(azo_agent *ag, int slot) {
    azo_mcts_args ma = ag->args.mcts;
    ma.tape_stream = ag->args.slot_base + (uint64_t)slot;
    azo_mcts *m = azo_mcts_new(&ma);
    m->ctr = &ag->slot_ctr[slot];
    return m;
}
static void agent_reset_mcts(azo_agent *ag, int slot) {             /* _get_mcts :60-66 */
    int T = ag->args.is_arena ? ag->P : 1;
    for (int t = 0; t < T; t++) {
        azo_mcts_free(ag->mcts[slot * T + t]);
        ag->mcts[slot * T + t] = agent_new_mcts(ag, slot);
    }
}
static azo_mcts *agent_mcts(azo_agent *ag, int slot) {              /* _mcts :68-73 */
    if (ag->args.is_arena) return ag->mcts[slot * ag->P + ag->games[slot].player];
    return ag->mcts[slot];
}
azo_mcts *azo_agent_mcts(azo_agent *ag, int slot, int player) {
    return ag->args.is_arena ? ag->mcts[slot * ag->P + player] : ag->mcts[slot];
}

azo_agent *azo_agent_new(int game, const azo_agent_args *a) {       /* __init__ :14-58 */
    azo_agent *ag = (azo_agent *)calloc(1, sizeof(*ag));
    ag->game = game; ag->args = *a; azo_game_info_get(game, &ag->gi);
    ag->B = a->batch_size; ag->P = ag->gi.num_players; ag->A = ag->gi.action_size;
    ag->O = ag->gi.obs_c * ag->gi.obs_h * ag->gi.obs_w;
    int B = ag->B, T = a->is_arena ? ag->P : 1;
    float *tt = (float *)malloc(sizeof(float) * (size_t)(a->temp_table_len > 0 ? a->temp_table_len : 1));
    for (int i = 0; i < a->temp_table_len; i++) tt[i] = a->temp_table[i];
    ag->args.temp_table = tt;
    ag->games = (azo_state *)calloc((size_t)B, sizeof(azo_state));
    ag->hist = (hist_entry **)calloc((size_t)B, sizeof(hist_entry *));
    ag->hist_len = (int *)calloc((size_t)B, sizeof(int)); ag->hist_cap = (int *)calloc((size_t)B, sizeof(int));
    ag->temps = (float *)calloc((size_t)B, sizeof(float)); ag->next_reset = (int *)calloc((size_t)B, sizeof(int));
    ag->mcts = (azo_mcts **)calloc((size_t)(B * T), sizeof(azo_mcts *));
    ag->slot_ctr = (uint64_t *)calloc((size_t)B, sizeof(uint64_t));
    ag->batch_indices = (int32_t *)calloc((size_t)B, sizeof(int32_t));
    ag->last_actions = (int32_t *)calloc((size_t)B, sizeof(int32_t));
    for (int p = 0; p < ag->P; p++) ag->player_to_index[p] = p;
    if (a->is_arena) {                                              /* :44-47 seat shuffle (agent stream) */
        int32_t pos[AZO_MAX_PLAYERS], tmp[AZO_MAX_PLAYERS];
        azo_tape_shuffle_pos(a->mcts.tape_seed, AGENT_STREAM(ag), ag->agent_ctr, ag->P, pos);
        ag->agent_ctr += (uint64_t)ag->P;
        for (int p = 0; p < ag->P; p++) tmp[pos[p]] = p;
        for (int p = 0; p < ag->P; p++) ag->player_to_index[p] = tmp[p];
    }
    for (int i = 0; i < B; i++) {                                   /* :54-59 */
        azo_game_init(game, &ag->games[i]);
        ag->temps[i] = a->startTemp;
        for (int t = 0; t < T; t++) ag->mcts[i * T + t] = agent_new_mcts(ag, i);
    }
    return ag;
}

void azo_agent_free(azo_agent *ag) {
    if (!ag) return;
    int T = ag->args.is_arena ? ag->P : 1;
    for (int i = 0; i < ag->B * T; i++) azo_mcts_free(ag->mcts[i]);
    for (int i = 0; i < ag->B; i++) { for (int h = 0; h < ag->hist_len[i]; h++) free(ag->hist[i][h].pi); free(ag->hist[i]); }
    free((void *)ag->args.temp_table);
    free(ag->games); free(ag->hist); free(ag->hist_len); free(ag->hist_cap); free(ag->temps); free(ag->next_reset);
    free(ag->mcts); free(ag->slot_ctr); free(ag->batch_indices); free(ag->last_actions);
    free(ag->s_obs); free(ag->s_pi); free(ag->s_z); free(ag->r_ws); free(ag->r_turns); free(ag->r_slot);
    free(ag);
}

/* run() :82-86: the per-round fast coin and sims count */
int azo_agent_begin_round(azo_agent *ag) {
    double u = azo_tape_uniform(ag->args.mcts.tape_seed, AGENT_STREAM(ag), ag->agent_ctr); ag->agent_ctr += 1;
    ag->fast = u < (double)ag->args.probFastSim;
    if (ag->fast) return ag->args.numFastSims;
    return ag->args.is_warmup ? ag->args.numWarmupSims : ag->args.numMCTSSims;
}

/* generateBatch :103-135 */
void azo_agent_generate_batch(azo_agent *ag, float *obs, int32_t *row_game, int32_t *row_model) {
    int B = ag->B; azo_state leaf;
    if (!ag->args.is_arena) {
        for (int i = 0; i < B; i++) {
            ag->expansions += (uint64_t)azo_mcts_find_leaf(agent_mcts(ag, i), ag->game, &ag->games[i], &leaf);
            if (!ag->args.is_warmup) azo_game_observation(ag->game, &leaf, obs + (size_t)i * ag->O);
            if (row_game) row_game[i] = i;
            if (row_model) row_model[i] = 0;
        }
        return;
    }
    /* arena :117-132: rows grouped by model index, games in slot order inside a group */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)B * ag->O);
    int *model = (int *)malloc(sizeof(int) * (size_t)B);
    for (int i = 0; i < B; i++) {
        int mover = ag->games[i].player;
        ag->expansions += (uint64_t)azo_mcts_find_leaf(agent_mcts(ag, i), ag->game, &ag->games[i], &leaf);
        azo_game_observation(ag->game, &leaf, tmp + (size_t)i * ag->O);
        model[i] = ag->player_to_index[mover];
    }
    int row = 0;
    for (int mi = 0; mi < ag->P; mi++)
        for (int i = 0; i < B; i++) if (model[i] == mi) {
            memcpy(obs + (size_t)row * ag->O, tmp + (size_t)i * ag->O, sizeof(float) * (size_t)ag->O);
            ag->batch_indices[row] = i;
            if (row_game) row_game[row] = i;
            if (row_model) row_model[row] = mi;
            row++;
        }
    free(tmp); free(model);
}

/* processBatch :137-151 */
void azo_agent_process_batch(azo_agent *ag, const float *policy, const float *value) {
    int B = ag->B, A = ag->A, V = ag->P + 1;
    int32_t *inv = NULL;
    if (ag->args.is_arena && !ag->args.arena_ref_misroute) {
        inv = (int32_t *)malloc(sizeof(int32_t) * (size_t)B);
        for (int r = 0; r < B; r++) inv[ag->batch_indices[r]] = r;
    }
    float pi[MAXA], val[AZO_MAX_PLAYERS + 1];
    for (int i = 0; i < B; i++) {
        int index = i;
        if (ag->args.is_arena) index = inv ? inv[i] : ag->batch_indices[i];       /* :144 (Q15) */
        if (ag->args.is_warmup) {                                                 /* :48-52,111-114 */
            float wp = (float)(1.0 / (double)A), wv = (float)(1.0 / (double)V);
            for (int a = 0; a < A; a++) pi[a] = wp;
            for (int j = 0; j < V; j++) val[j] = wv;
        } else {
            memcpy(pi, policy + (size_t)index * A, sizeof(float) * (size_t)A);
            memcpy(val, value + (size_t)index * V, sizeof(float) * (size_t)V);
        }
        azo_mcts_process_results(agent_mcts(ag, i), ag->game, val, pi,
                                 ag->args.is_arena ? 0 : ag->args.add_root_noise,
                                 ag->args.is_arena ? 0 : ag->args.add_root_temp);
        ag->sims_done++;
    }
    free(inv);
}

static void push_sample(azo_agent *ag, const float *obs, const float *pi, const uint8_t *ws) {
    int V = ag->P + 1;
    if (ag->n_samples == ag->cap_samples) {
        ag->cap_samples = ag->cap_samples ? ag->cap_samples * 2 : 1024;
        ag->s_obs = (float *)realloc(ag->s_obs, sizeof(float) * (size_t)ag->cap_samples * ag->O);
        ag->s_pi = (float *)realloc(ag->s_pi, sizeof(float) * (size_t)ag->cap_samples * ag->A);
        ag->s_z = (float *)realloc(ag->s_z, sizeof(float) * (size_t)ag->cap_samples * V);
    }
    memcpy(ag->s_obs + (size_t)ag->n_samples * ag->O, obs, sizeof(float) * (size_t)ag->O);
    memcpy(ag->s_pi + (size_t)ag->n_samples * ag->A, pi, sizeof(float) * (size_t)ag->A);
    for (int j = 0; j < V; j++) ag->s_z[(size_t)ag->n_samples * V + j] = (float)ws[j];
    ag->n_samples++;
}

/* playMoves :153-202 */
int azo_agent_play_moves(azo_agent *ag) {
    int B = ag->B, A = ag->A, finished = 0;
    int T = ag->args.is_arena ? ag->P : 1;
    float policy[MAXA], pi1[MAXA], obs[4096], pis[MAXA];
    for (int i = 0; i < B; i++) {
        if (ag->args.is_arena) ag->temps[i] = ag->args.arenaTemp;                     /* :158 */
        else {                                                                        /* :156-157 temp_scaling_fn */
            int t = ag->games[i].turns;
            if (t >= ag->args.temp_table_len) t = ag->args.temp_table_len - 1;
            ag->temps[i] = ag->args.temp_table[t];
        }
        azo_mcts *mc = agent_mcts(ag, i);
        azo_mcts_probs(mc, ag->game, ag->temps[i], policy);                           /* :159 */
        int action = azo_tape_choice(ag->args.mcts.tape_seed, ag->args.slot_base + (uint64_t)i, ag->slot_ctr[i], policy, A);
        ag->slot_ctr[i] += 1;                                                         /* :160 */
        ag->last_actions[i] = action;
        if (!ag->fast && !ag->args.is_arena) {                                        /* :161-165 */
            if (ag->hist_len[i] == ag->hist_cap[i]) {
                ag->hist_cap[i] = ag->hist_cap[i] ? ag->hist_cap[i] * 2 : 64;
                ag->hist[i] = (hist_entry *)realloc(ag->hist[i], sizeof(hist_entry) * (size_t)ag->hist_cap[i]);
            }
            azo_mcts_probs(mc, ag->game, 1.0f, pi1);
            hist_entry *h = &ag->hist[i][ag->hist_len[i]++];
            h->st = ag->games[i]; h->pi = (float *)malloc(sizeof(float) * (size_t)A);
            memcpy(h->pi, pi1, sizeof(float) * (size_t)A);
        }
        if (ag->args.is_arena) { for (int t = 0; t < T; t++) azo_mcts_update_root(ag->mcts[i * T + t], ag->game, &ag->games[i], action); }
        else azo_mcts_update_root(mc, ag->game, &ag->games[i], action);               /* :167-170 */
        azo_game_play(ag->game, &ag->games[i], action);                               /* :171 */
        if (ag->args.mctsResetThreshold && ag->games[i].turns >= ag->next_reset[i]) { /* :172-174 */
            agent_reset_mcts(ag, i);
            ag->next_reset[i] = ag->games[i].turns + ag->args.mctsResetThreshold;
        }
        uint8_t ws[AZO_MAX_PLAYERS + 1];
        azo_game_win_state(ag->game, &ag->games[i], ws);                              /* :176 */
        int any = 0; for (int j = 0; j < ag->P + 1; j++) any |= ws[j];
        if (any) {
            if (ag->n_results == ag->cap_results) {                                   /* :178 result_queue.put */
                ag->cap_results = ag->cap_results ? ag->cap_results * 2 : 256;
                ag->r_ws = (uint8_t *)realloc(ag->r_ws, (size_t)ag->cap_results * (AZO_MAX_PLAYERS + 1));
                ag->r_turns = (int32_t *)realloc(ag->r_turns, sizeof(int32_t) * (size_t)ag->cap_results);
                ag->r_slot = (int32_t *)realloc(ag->r_slot, sizeof(int32_t) * (size_t)ag->cap_results);
            }
            memcpy(ag->r_ws + (size_t)ag->n_results * (AZO_MAX_PLAYERS + 1), ws, AZO_MAX_PLAYERS + 1);
            ag->r_turns[ag->n_results] = ag->games[i].turns; ag->r_slot[ag->n_results] = i; ag->n_results++;
            if (ag->games_played < ag->args.gamesPerIteration) {                      /* :179-183 */
                ag->games_played += 1; finished++;
                if (!ag->args.is_arena) {                                             /* :184-196 */
                    for (int h = 0; h < ag->hist_len[i]; h++) {
                        int nsym = ag->args.symmetricSamples ? ag->gi.num_symmetries : 1;
                        for (int k = 0; k < nsym; k++) {
                            azo_state ss;
                            azo_game_symmetry(ag->game, &ag->hist[i][h].st, ag->hist[i][h].pi, k, &ss, pis);
                            azo_game_observation(ag->game, &ss, obs);
                            push_sample(ag, obs, pis, ws);
                        }
                    }
                }
                azo_game_init(ag->game, &ag->games[i]);                               /* :197-200 */
                for (int h = 0; h < ag->hist_len[i]; h++) free(ag->hist[i][h].pi);
                ag->hist_len[i] = 0;
                ag->temps[i] = ag->args.startTemp;
                agent_reset_mcts(ag, i);
            }
        }
    }
    return finished;
}

int azo_agent_games_played(const azo_agent *ag) { return ag->games_played; }
int azo_agent_num_samples(const azo_agent *ag) { return ag->n_samples; }
int azo_agent_num_results(const azo_agent *ag) { return ag->n_results; }
void azo_agent_get_samples(const azo_agent *ag, float *obs, float *pi, float *z) {
    memcpy(obs, ag->s_obs, sizeof(float) * (size_t)ag->n_samples * ag->O);
    memcpy(pi, ag->s_pi, sizeof(float) * (size_t)ag->n_samples * ag->A);
    memcpy(z, ag->s_z, sizeof(float) * (size_t)ag->n_samples * (ag->P + 1));
}
void azo_agent_get_results(const azo_agent *ag, uint8_t *ws, int32_t *turns, int32_t *slot) {
    for (int i = 0; i < ag->n_results; i++) {
        for (int j = 0; j < ag->P + 1; j++) ws[i * (ag->P + 1) + j] = ag->r_ws[(size_t)i * (AZO_MAX_PLAYERS + 1) + j];
        turns[i] = ag->r_turns[i]; slot[i] = ag->r_slot[i];
    }
}
void azo_agent_get_state(const azo_agent *ag, int slot, azo_state *out) { *out = ag->games[slot]; }
void azo_agent_last_actions(const azo_agent *ag, int32_t *actions) { memcpy(actions, ag->last_actions, sizeof(int32_t) * (size_t)ag->B); }
const int32_t *azo_agent_player_to_index(const azo_agent *ag) { return ag->player_to_index; }
uint64_t azo_agent_sims_done(const azo_agent *ag) { return ag->sims_done; }
uint64_t azo_agent_expansions(const azo_agent *ag) { return ag->expansions; }
