/*
 * azg_oracle.h -- CPU ORACLE for the self-play / MCTS hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (kevaday/alphazero-general):
 *   alphazero/MCTS.pyx, alphazero/SelfPlayAgent.pyx, alphazero/envs/connect4/, alphazero/envs/brandubh/,
 *   fastafl/cengine.pyx, boardgame/board.pyx.
 * Every function cites the reference file:line it follows.  It deliberately keeps the reference's own
 * structure (heap-allocated Node objects holding a shuffled child list, a Python-style path stack) and
 * NOT the product's SoA/arena layout, so that the two are independent implementations.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (alphazero_general_amd/csrc, libazg_hip.so) never links, includes or calls anything here.
 *
 * Parity pin: tests/golden/ holds vectors produced by the *actual* reference (Cython, imported from
 * /root/reference in the build container by tests/golden/make_goldens.py); tests/test_oracle_golden.py checks
 * this oracle against them bit-for-bit.
 */
#ifndef AZG_ORACLE_H
#define AZG_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- games ---------------------------------------------------------------------------------- */
enum { AZO_GAME_CONNECT4 = 0, AZO_GAME_BRANDUBH = 1, AZO_GAME_TRIMOK = 2, AZO_NUM_GAMES = 3 };
#define AZO_MAX_PLAYERS 4
#define AZO_MAX_CELLS 64

typedef struct azo_state {
    int8_t  cells[AZO_MAX_CELLS]; /* connect4: 6x7 row-major, 1 / -1 / 0 (Connect4Logic.pyx:34)       */
                                  /* brandubh: 7x7 row-major piece codes 0..8 (fastafl/cengine.pyx:24-32) */
    int32_t player;               /* GameState._player (Game.py:10)                                   */
    int32_t turns;                /* GameState._turns  (Game.py:11)                                   */
    int32_t aux[4];               /* game specific (brandubh: winner cache etc.)                      */
} azo_state;

typedef struct azo_game_info {
    int32_t action_size, obs_c, obs_h, obs_w, num_players, has_draw, max_turns, num_symmetries, cells;
} azo_game_info;

int  azo_game_info_get(int game, azo_game_info *out);
void azo_game_init(int game, azo_state *s);
int  azo_game_play(int game, azo_state *s, int action);               /* 0 ok, <0 illegal            */
void azo_game_valid_moves(int game, const azo_state *s, uint8_t *valid /*[A]*/);
void azo_game_win_state(int game, const azo_state *s, uint8_t *ws /*[P+1]*/);
void azo_game_observation(int game, const azo_state *s, float *obs /*[C*H*W]*/);
/* k-th symmetry of (state, pi): writes the transformed state and policy (k=0 is the identity).      */
void azo_game_symmetry(int game, const azo_state *s, const float *pi, int k, azo_state *s_out, float *pi_out);

/* ---- random tape (definition shared by oracle + product; see DESIGN.md "Random tape") -------- */
uint64_t azo_tape_u64(uint64_t seed, uint64_t stream, uint64_t ctr);
/* shuffle: pos[i] = position in the shuffled list of the i-th element (ascending order in). k draws */
void     azo_tape_shuffle_pos(uint64_t seed, uint64_t stream, uint64_t ctr, int k, int32_t *pos);
/* choice: one draw; numpy-legacy semantics (double cdf, cdf/=cdf[-1], searchsorted right)           */
int      azo_tape_choice(uint64_t seed, uint64_t stream, uint64_t ctr, const float *p, int n);
/* dirichlet([alpha]*k): one draw from the stream (event key) + per-element sub-streams              */
void     azo_tape_dirichlet(uint64_t seed, uint64_t stream, uint64_t ctr, int k, double alpha, double *out);
double   azo_tape_uniform(uint64_t seed, uint64_t stream, uint64_t ctr);   /* [0,1) 53-bit           */
/* replay of recorded draws for one stream (the MT19937 tier): ranks int16[len], u double[len], noise_off int32[len] into noise_pool  */
void     azo_tape_set_replay(uint64_t stream, const int16_t *ranks, const double *u, const int32_t *noise_off, const float *noise_pool, int len);
void     azo_tape_clear_replay(void);
double   azo_det_log(double x);
double   azo_det_exp(double x);

/* numpy float32 pairwise np.sum restatement (numpy/_core/src/umath/loops_utils.h.src pairwise sum)  */
float    azo_np_sum_f32(const float *a, int n);
/* numpy float32 `a ** python_float` restatement used by MCTS.pyx:250,320                            */
float    azo_np_pow_f32(float x, double e);

/* deterministic synthetic evaluator (stands in for the network in tree-parity tests)               */
void     azo_fake_eval(uint64_t seed, uint64_t slot, uint64_t sim, int A, int nv, float *p, float *v);

/* ---- MCTS (alphazero/MCTS.pyx) --------------------------------------------------------------- */
typedef struct azo_mcts_args {
    float root_noise_frac, root_policy_temp, min_discount, fpu_reduction, cpuct; /* MCTS.pyx:134-138 */
    int32_t num_players_plus_draw;                                              /* args._num_players */
    uint64_t tape_seed, tape_stream;                                            /* random tape key    */
} azo_mcts_args;

typedef struct azo_mcts azo_mcts;

azo_mcts *azo_mcts_new(const azo_mcts_args *a);
void      azo_mcts_free(azo_mcts *m);
void      azo_mcts_reset(azo_mcts *m);                                          /* MCTS.pyx:154-160 */
uint64_t  azo_mcts_tape_ctr(const azo_mcts *m);
void      azo_mcts_set_tape_ctr(azo_mcts *m, uint64_t c);
/* MCTS.pyx:208-228. leaf_out receives the leaf state. returns 1 if an expansion happened, else 0.   */
int       azo_mcts_find_leaf(azo_mcts *m, int game, const azo_state *gs, azo_state *leaf_out);
/* MCTS.pyx:230-289. pi is modified in place like the reference does.                                */
void      azo_mcts_process_results(azo_mcts *m, int game, float *value, float *pi, int add_root_noise, int add_root_temp);
int       azo_mcts_update_root(azo_mcts *m, int game, const azo_state *gs, int a); /* :185-195; -1 = ValueError */
void      azo_mcts_counts(const azo_mcts *m, int game, int32_t *counts);           /* :297-303        */
void      azo_mcts_probs(const azo_mcts *m, int game, float temp, float *probs);   /* :308-329        */
float     azo_mcts_value(const azo_mcts *m, int average);                          /* :331-344        */
void      azo_mcts_raw_search(azo_mcts *m, int game, const azo_state *gs, int sims, int noise, int temp); /* :175-183 */
int       azo_mcts_root_n(const azo_mcts *m);
int       azo_mcts_max_depth(const azo_mcts *m);
int       azo_mcts_depth(const azo_mcts *m);
/* dump the root's children in list order: a, n, q, p, v (arrays of length >= nchildren). returns k  */
int       azo_mcts_root_children(const azo_mcts *m, int32_t *a, int32_t *n, float *q, float *p, float *v);
/* the path of the last find_leaf as actions from the root (length = depth)                          */
int       azo_mcts_last_path(const azo_mcts *m, int32_t *actions);
void      azo_mcts_root_header(const azo_mcts *m, int32_t *n, float *q, float *v, int32_t *player, uint8_t *e);

/* ---- SelfPlayAgent (alphazero/SelfPlayAgent.pyx) --------------------------------------------- */
typedef struct azo_agent_args {
    azo_mcts_args mcts;
    int32_t batch_size;
    int32_t numMCTSSims, numFastSims, numWarmupSims;
    float   probFastSim;
    int32_t gamesPerIteration;
    int32_t add_root_noise, add_root_temp, symmetricSamples;
    int32_t mctsResetThreshold;           /* 0 = None                                           */
    float   startTemp, arenaTemp;
    int32_t temp_table_len;               /* temp_by_turn[t] precomputed from args.temp_scaling_fn */
    const float *temp_table;
    int32_t is_arena, is_warmup;
    int32_t arena_ref_misroute;           /* 1: reproduce reference Q15 (row->game used as game->row) */
    uint64_t slot_base;                   /* global slot id of local slot 0 (multi-GPU sharding)   */
} azo_agent_args;

typedef struct azo_agent azo_agent;
azo_agent *azo_agent_new(int game, const azo_agent_args *a);
void       azo_agent_free(azo_agent *ag);
/* one round header: draws the `fast` coin (SelfPlayAgent.pyx:84); returns number of sims for the round */
int        azo_agent_begin_round(azo_agent *ag);
/* SelfPlayAgent.pyx:103-135. obs [B, C*H*W] f32. In arena mode also fills model_of_row / row order  */
void       azo_agent_generate_batch(azo_agent *ag, float *obs, int32_t *row_game, int32_t *row_model);
/* SelfPlayAgent.pyx:137-151. policy [B,A], value [B,P+1] indexed by ROW (self-play: row == game)    */
void       azo_agent_process_batch(azo_agent *ag, const float *policy, const float *value);
/* SelfPlayAgent.pyx:153-202. returns number of games finished in this call                         */
int        azo_agent_play_moves(azo_agent *ag);
int        azo_agent_games_played(const azo_agent *ag);
int        azo_agent_num_samples(const azo_agent *ag);
int        azo_agent_num_results(const azo_agent *ag);
/* copy out accumulated (obs, pi, z) samples in output_queue order                                   */
void       azo_agent_get_samples(const azo_agent *ag, float *obs, float *pi, float *z);
/* results in result_queue order: winstate u8[P+1], turns, slot                                     */
void       azo_agent_get_results(const azo_agent *ag, uint8_t *winstate, int32_t *turns, int32_t *slot);
void       azo_agent_get_state(const azo_agent *ag, int slot, azo_state *out);
void       azo_agent_last_actions(const azo_agent *ag, int32_t *actions /*[B]*/);
const int32_t *azo_agent_player_to_index(const azo_agent *ag);
azo_mcts  *azo_agent_mcts(azo_agent *ag, int slot, int player);
uint64_t   azo_agent_sims_done(const azo_agent *ag);
uint64_t   azo_agent_expansions(const azo_agent *ag);

/* ---- many agents on many host cores (azg_pool_ref.c; bench.py's cpu_baseline leg) ------------------ */
typedef struct azo_pool azo_pool;
azo_pool  *azo_pool_new(int game, const azo_agent_args *a, int n_agents);   /* one thread per agent; agent i: slot_base + i * B */
void       azo_pool_free(azo_pool *p);
void       azo_pool_begin_round(azo_pool *p);
void       azo_pool_generate(azo_pool *p, float *obs /*[n*B, C*H*W]*/);
void       azo_pool_process(azo_pool *p, const float *policy /*[n*B, A]*/, const float *value /*[n*B, P+1]*/);
int        azo_pool_play(azo_pool *p);
double     azo_pool_run_tree_only(azo_pool *p, double seconds);
uint64_t   azo_pool_expansions(const azo_pool *p);
uint64_t   azo_pool_sims(const azo_pool *p);
int        azo_pool_games_played(const azo_pool *p);
azo_agent *azo_pool_agent(azo_pool *p, int i);
void       azo_pool_row_models(const azo_pool *p, int32_t *out /*[n*B]*/);   /* arena: model of every row of the last generate */

#ifdef __cplusplus
}
#endif
#endif
