/*
 * azg_pool_ref.c -- CPU ORACLE, TEST INFRASTRUCTURE ONLY: many oracle agents on many host cores.
 *
 * The reference spreads self-play over `workers` agent PROCESSES, one per core, each searching its own batch of games and all of
 * them sharing one network (alphazero/Coach.py:291-342).  This file is that arrangement for the C restatement: N azo_agents, one
 * POSIX thread each, stepped in lock step so that their leaf batches can be evaluated as ONE network batch (bench.py's
 * cpu_baseline leg), plus a free-running tree-only loop with the warm-up evaluator's uniform policy / value
 * (SelfPlayAgent.pyx:48-52,111-114).  Nothing here is part of the algorithm; parity tests never touch it.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "azg_oracle.h"

enum { CMD_NONE = 0, CMD_BEGIN, CMD_GENERATE, CMD_PROCESS, CMD_PLAY, CMD_TREE_ONLY, CMD_EXIT };

struct azo_pool {
    int game, n, B, A, NV, O;
    azo_agent **ag;
    pthread_t *th;
    pthread_mutex_t mu; pthread_cond_t go, done;
    int cmd, gen, pending;
    float *obs; const float *pol, *val;          /* whole-pool buffers of the current command */
    int32_t *scratch_i32;                        /* [n][2][B] row_game / row_model (unused in self-play) */
    double seconds; int sims_round;
    int *finished;                               /* per agent: games finished by the last CMD_PLAY */
};

typedef struct { struct azo_pool *p; int i; } worker_arg;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void run_cmd(struct azo_pool *p, int i, int cmd) {
    azo_agent *a = p->ag[i];
    int32_t *rg = p->scratch_i32 + (size_t)i * 2 * (size_t)p->B, *rm = rg + p->B;
    switch (cmd) {
    case CMD_BEGIN: azo_agent_begin_round(a); break;
    case CMD_GENERATE: azo_agent_generate_batch(a, p->obs + (size_t)i * (size_t)p->B * (size_t)p->O, rg, rm); break;
    case CMD_PROCESS: azo_agent_process_batch(a, p->pol + (size_t)i * (size_t)p->B * (size_t)p->A, p->val + (size_t)i * (size_t)p->B * (size_t)p->NV); break;
    case CMD_PLAY: p->finished[i] = azo_agent_play_moves(a); break;
    case CMD_TREE_ONLY: {                        /* whole rounds with the uniform evaluator until the time is up; no coupling */
        float *obs = (float *)malloc(sizeof(float) * (size_t)p->B * (size_t)p->O);
        float *pol = (float *)malloc(sizeof(float) * (size_t)p->B * (size_t)p->A), *val = (float *)malloc(sizeof(float) * (size_t)p->B * (size_t)p->NV);
        const double t_end = now_s() + p->seconds;
        while (now_s() < t_end) {
            const int sims = azo_agent_begin_round(a);
            for (int s = 0; s < sims; s++) {
                azo_agent_generate_batch(a, obs, rg, rm);
                for (int j = 0; j < p->B * p->A; j++) pol[j] = 1.0f / (float)p->A;      /* (process_results scales pi in place) */
                for (int j = 0; j < p->B * p->NV; j++) val[j] = 1.0f / (float)p->NV;
                azo_agent_process_batch(a, pol, val);
            }
            azo_agent_play_moves(a);
        }
        free(obs); free(pol); free(val);
    } break;
    default: break;
    }
}

static void *worker(void *arg_) {
    worker_arg *wa = (worker_arg *)arg_;
    struct azo_pool *p = wa->p; const int i = wa->i;
    free(wa);
    int seen = 0;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (p->gen == seen) pthread_cond_wait(&p->go, &p->mu);
        seen = p->gen;
        const int cmd = p->cmd;
        pthread_mutex_unlock(&p->mu);
        if (cmd == CMD_EXIT) return NULL;
        run_cmd(p, i, cmd);
        pthread_mutex_lock(&p->mu);
        if (--p->pending == 0) pthread_cond_signal(&p->done);
        pthread_mutex_unlock(&p->mu);
    }
}

static void dispatch(struct azo_pool *p, int cmd) {
    pthread_mutex_lock(&p->mu);
    p->cmd = cmd; p->pending = p->n; p->gen++;
    pthread_cond_broadcast(&p->go);
    while (p->pending > 0) pthread_cond_wait(&p->done, &p->mu);
    pthread_mutex_unlock(&p->mu);
}

/* n agents of a->batch_size games each; agent i owns the global slots [slot_base + i * B, + B) of the same random tape */
struct azo_pool *azo_pool_new(int game, const azo_agent_args *a, int n) {
    struct azo_pool *p = (struct azo_pool *)calloc(1, sizeof(*p));
    azo_game_info gi; azo_game_info_get(game, &gi);
    p->game = game; p->n = n; p->B = a->batch_size; p->A = gi.action_size; p->NV = gi.num_players + 1; p->O = gi.obs_c * gi.obs_h * gi.obs_w;
    p->ag = (azo_agent **)calloc((size_t)n, sizeof(azo_agent *));
    p->th = (pthread_t *)calloc((size_t)n, sizeof(pthread_t));
    p->scratch_i32 = (int32_t *)calloc((size_t)n * 2 * (size_t)p->B, sizeof(int32_t));
    p->finished = (int *)calloc((size_t)n, sizeof(int));
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->go, NULL); pthread_cond_init(&p->done, NULL);
    for (int i = 0; i < n; i++) {
        azo_agent_args ai = *a;
        ai.slot_base = a->slot_base + (uint64_t)i * (uint64_t)p->B;
        p->ag[i] = azo_agent_new(game, &ai);
    }
    for (int i = 0; i < n; i++) {
        worker_arg *wa = (worker_arg *)malloc(sizeof(*wa)); wa->p = p; wa->i = i;
        pthread_create(&p->th[i], NULL, worker, wa);
    }
    return p;
}

void azo_pool_free(struct azo_pool *p) {
    if (!p) return;
    pthread_mutex_lock(&p->mu); p->cmd = CMD_EXIT; p->gen++; pthread_cond_broadcast(&p->go); pthread_mutex_unlock(&p->mu);
    for (int i = 0; i < p->n; i++) pthread_join(p->th[i], NULL);
    for (int i = 0; i < p->n; i++) azo_agent_free(p->ag[i]);
    free(p->ag); free(p->th); free(p->scratch_i32); free(p->finished);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->go); pthread_cond_destroy(&p->done);
    free(p);
}

void azo_pool_begin_round(struct azo_pool *p) { dispatch(p, CMD_BEGIN); }
/* obs [n * B, C*H*W]: agent i writes rows [i * B, (i + 1) * B) */
void azo_pool_generate(struct azo_pool *p, float *obs) { p->obs = obs; dispatch(p, CMD_GENERATE); }
void azo_pool_process(struct azo_pool *p, const float *policy, const float *value) { p->pol = policy; p->val = value; dispatch(p, CMD_PROCESS); }
int  azo_pool_play(struct azo_pool *p) { dispatch(p, CMD_PLAY); int f = 0; for (int i = 0; i < p->n; i++) f += p->finished[i]; return f; }
/* every agent plays whole rounds on its own thread with the uniform evaluator for `seconds`; returns the wall time spent */
double azo_pool_run_tree_only(struct azo_pool *p, double seconds) { p->seconds = seconds; const double t0 = now_s(); dispatch(p, CMD_TREE_ONLY); return now_s() - t0; }
uint64_t azo_pool_expansions(const struct azo_pool *p) { uint64_t s = 0; for (int i = 0; i < p->n; i++) s += azo_agent_expansions(p->ag[i]); return s; }
uint64_t azo_pool_sims(const struct azo_pool *p) { uint64_t s = 0; for (int i = 0; i < p->n; i++) s += azo_agent_sims_done(p->ag[i]); return s; }
int      azo_pool_games_played(const struct azo_pool *p) { int s = 0; for (int i = 0; i < p->n; i++) s += azo_agent_games_played(p->ag[i]); return s; }
azo_agent *azo_pool_agent(struct azo_pool *p, int i) { return p->ag[i]; }
/* arena mode: the model (player_to_index[mover]) that evaluates each row of the last azo_pool_generate, [n * B] */
void azo_pool_row_models(const struct azo_pool *p, int32_t *out) {
    for (int i = 0; i < p->n; i++) memcpy(out + (size_t)i * (size_t)p->B, p->scratch_i32 + (size_t)i * 2 * (size_t)p->B + p->B, sizeof(int32_t) * (size_t)p->B);
}
