/*
 * azg.h -- C ABI of libazg_hip.so, the MI355X-native batched self-play / MCTS engine.
 *
 * The reference (kevaday/alphazero-general) has no FFI: its hot path is two Cython extension classes,
 * alphazero/MCTS.pyx and alphazero/SelfPlayAgent.pyx.  This header is what a binding for that path would bind;
 * each entry point names the reference interface it replaces (file:line relative to the reference root).
 * The Python classes in alphazero_general_amd/{MCTS,SelfPlayAgent}.py are thin ctypes callers of exactly these
 * functions (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C types only; every buffer is caller-owned.  "dev" pointers are device (HBM) pointers of the calling
 *     process' HIP context, "host" pointers are ordinary host memory.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  All work is stream-ordered; nothing
 *     synchronises unless the function name says `_sync`/`_read` or the doc says "blocking".
 *   - one engine per GPU per process; an engine is not thread-safe.
 *   - return 0 on success, a negative azg_status otherwise; azg_last_error() gives the message
 *     (the Python layer maps AZG_E_INVALID_ACTION to ValueError like MCTS.pyx:195, the rest to RuntimeError).
 *   - there is NO CPU fallback: every compute entry point launches HIP kernels and fails with AZG_E_HIP when no
 *     device is available.
 */
#ifndef AZG_H
#define AZG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZG_ABI_VERSION 6

typedef enum azg_status {
    AZG_OK = 0,
    AZG_E_INVALID_ARG = -1,
    AZG_E_HIP = -2,             /* HIP runtime error / no device                                   */
    AZG_E_INVALID_ACTION = -3,  /* update_root on an action that is not a child (MCTS.pyx:195)      */
    AZG_E_TREE_FULL = -4,       /* a tree's node arena overflowed (raise nodes_per_tree)            */
    AZG_E_EXAMPLES_FULL = -5,   /* the training-example buffer overflowed (raise example_capacity)  */
    AZG_E_UNSUPPORTED = -6,
    AZG_E_INTERNAL = -7,        /* a bounded device-side wait expired (never expected; reported instead of hanging) */
    AZG_E_FLOATING_POINT = -8   /* a division the reference turns into FloatingPointError (np.seterr(all='raise'), MCTS.pyx:23):
                                   MCTS.probs at a root without a visited child (counts / 0, :320; numMCTSSims < 2), or a policy
                                   whose valid entries sum to 0 (:245).  The Python layer raises FloatingPointError too        */
} azg_status;

/* games with device-side rules (Game plugin API, alphazero/Game.py:7-113) */
typedef enum azg_game {
    AZG_GAME_CONNECT4 = 0,      /* alphazero/envs/connect4/connect4.pyx + Connect4Logic.pyx          */
    AZG_GAME_BRANDUBH = 1,      /* alphazero/envs/brandubh/fastafl.pyx + fastafl/cengine.pyx         */
    AZG_GAME_TRIMOK = 2         /* build-defined 3-player env (N-player path, BASELINE config 5)     */
} azg_game;

/* Game state as it crosses the ABI (all games): the reference's board array, row-major, one int8 per cell
 * (connect4: 1/-1/0 as Connect4Logic.pyx:34; brandubh: piece codes of fastafl/cengine.pyx:24-32),
 * GameState._player / _turns (Game.py:10-11) and two game-specific words. */
typedef struct azg_state {
    int8_t  cells[64];
    int32_t player;
    int32_t turns;
    int32_t aux[2];
} azg_state;                    /* 80 bytes */

typedef struct azg_game_info {
    int32_t action_size;        /* GameState.action_size()       Game.py:26-30   */
    int32_t obs_c, obs_h, obs_w;/* GameState.observation_size()  Game.py:32-42   */
    int32_t num_players;        /* GameState.num_players()       Game.py:49-53   */
    int32_t has_draw;           /* GameState.has_draw()          Game.py:60-63   */
    int32_t max_turns;          /* GameState.max_turns()         Game.py:55-58   */
    int32_t num_symmetries;     /* len(GameState.symmetries())   Game.py:99-113  */
    int32_t cells;              /* board cells used in azg_state.cells            */
    int32_t max_children;       /* upper bound on legal moves in one position     */
} azg_game_info;

/* Engine configuration.  Search keys are the ones MCTS.__init__ reads (MCTS.pyx:133-139); agent keys the ones
 * SelfPlayAgent reads (SelfPlayAgent.pyx:58,82-86,149-150,156-158,172,181,187). */
typedef struct azg_config {
    int32_t  abi_version;       /* AZG_ABI_VERSION */
    int32_t  game;              /* azg_game */
    int32_t  device;            /* HIP device ordinal */
    int32_t  num_slots;         /* B: concurrent games (batch_tensor.shape[0], SelfPlayAgent.pyx:23-26) */
    int32_t  arena;             /* 1: one tree per player per game (SelfPlayAgent.pyx:60-73)      */
    int32_t  nodes_per_tree;    /* capacity of each of a tree's two node semi-spaces; 0 = min(max_turns, 16) * sims_per_move * max_children + 64
                                 * (the value in effect: azg_engine_info) */
    int32_t  example_capacity;  /* max (obs, pi, z) samples held; 0 = no sample recording         */
    int32_t  result_capacity;   /* max finished-game records held                                  */
    float    cpuct, fpu_reduction, root_noise_frac, root_policy_temp, min_discount;   /* MCTS.pyx:134-138 */
    int32_t  add_root_noise, add_root_temp;        /* SelfPlayAgent.pyx:149-150 (forced off in arena) */
    int32_t  symmetric_samples;                    /* args.symmetricSamples  :187                      */
    int32_t  mcts_reset_threshold;                 /* args.mctsResetThreshold :172-174, 0 = None       */
    int32_t  games_per_iteration;                  /* args.gamesPerIteration :179-183                  */
    float    start_temp, arena_temp;               /* args.startTemp :58, args.arenaTemp :158          */
    int32_t  temp_table_len;                       /* temp used for the move at turn t, i.e.           */
    int32_t  sims_per_move;                        /* max(numMCTSSims, numFastSims, numWarmupSims): sizes the node store and the
                                                    * free-node reserve below which a tree is compacted after a move; 0 = 100 */
    const float *temp_table;                       /*   args.temp_scaling_fn iterated (:156-157); host */
    uint64_t tape_seed;                            /* random tape                          */
    uint64_t slot_base;                            /* global id of slot 0 (multi-GPU sharding)         */
} azg_config;

typedef struct azg_engine azg_engine;

typedef struct azg_counters {
    int64_t  sims;              /* find_leaf + process_results pairs                              */
    int64_t  expansions;        /* find_leaf calls that took the n == 0 branch (MCTS.pyx:223-226)  */
    int32_t  games_played;      /* SelfPlayAgent.games_played (capped at games_per_iteration)     */
    int32_t  num_results;       /* result_queue length (every finished game)           */
    int32_t  num_examples;      /* output_queue length                                             */
    int32_t  error;             /* sticky azg_status raised on device (tree/example overflow ...)  */
    int32_t  max_nodes_used;    /* high-water mark of any tree's live node space (garbage included)  */
    int32_t  max_nodes_kept;    /* largest subtree a compaction has kept: + one move must fit nodes_per_tree */
} azg_counters;

/* ---- library ------------------------------------------------------------------------------------------- */
int          azg_abi_version(void);
/* content hash of the kernel sources (the headers under csrc/) this binary was built from, stamped by alphazero_general_amd/build.py ("unstamped" for a
 * build made another way): bench.py prints it and quotes profile counters next to a run only when they were taken on the same sources */
const char  *azg_source_sha(void);
const char  *azg_last_error(void);
int          azg_game_info_get(int game, azg_game_info *out);              /* Game.py static methods       */
int          azg_device_count(void);

/* ---- engine life cycle (SelfPlayAgent.__init__ :14-58 / MCTS.__init__ :133-145) -------------------------- */
int  azg_engine_create(const azg_config *cfg, azg_engine **out);
int  azg_engine_destroy(azg_engine *e);
/* the sizes in effect (defaults resolved): out8 = {nodes_per_tree, compaction reserve (free nodes below which a move's end compacts),
 * trees per slot, path entries per tree, num_slots, example_capacity, result_capacity, sims_per_move} */
int  azg_engine_info(azg_engine *e, int32_t *out8);
/* reset every slot to the initial position with fresh trees (SelfPlayAgent.pyx:54-59; MCTS.reset :154-160) */
int  azg_engine_reset(azg_engine *e, void *stream);
/* overwrite the game state of `count` slots starting at `first` (host array) and give them fresh trees */
int  azg_set_states(azg_engine *e, void *stream, int first, int count, const azg_state *host_states, int reset_trees);
int  azg_get_states(azg_engine *e, void *stream, int first, int count, azg_state *host_states);      /* blocking */
int  azg_get_leaf_states(azg_engine *e, void *stream, int first, int count, azg_state *host_states); /* blocking */
int  azg_set_tape_counters(azg_engine *e, void *stream, int first, int count, const uint64_t *host_ctr);
int  azg_get_tape_counters(azg_engine *e, void *stream, int first, int count, uint64_t *host_ctr);   /* blocking */

/* ---- one simulation step ---------------------------------------------------------------------------------- */
/* SelfPlayAgent.generateBatch (:103-135) = MCTS.find_leaf (:208-228) on every slot + leaf observation
 * (GameState.observation) written as row `row_of_slot[slot]` (NULL: row = slot) of the dense NN input
 * obs_dev[rows, C, H, W]; obs_dtype 0 = float32, 1 = float16 (both NCHW like the
 * reference), 2 = float16 NHWC with channels padded to 8: obs_dev[rows, H*W, 8], the input format of azg_resnet_tower_f16. */
int  azg_select(azg_engine *e, void *stream, void *obs_dev, int obs_dtype, const int32_t *row_of_slot_dev);
/* arena helper (:117-132): rows grouped by model index = player_to_index[mover]; writes row_of_slot_dev[B] and
 * rows_per_model_dev[P] (device), to be called before azg_select */
int  azg_arena_rows(azg_engine *e, void *stream, const int32_t *player_to_index_host, int32_t *row_of_slot_dev,
                    int32_t *rows_per_model_dev);
/* the same with one seating per SLOT instead of per agent: seat_of_slot_dev[B], 4 bits per player -- the model of player p in
 * slot s is (seat_of_slot_dev[s] >> 4p) & 15 (every concurrent game draws its own seats) */
int  azg_arena_rows_seats(azg_engine *e, void *stream, const uint32_t *seat_of_slot_dev, int32_t *row_of_slot_dev,
                          int32_t *rows_per_model_dev);
/* SelfPlayAgent.processBatch (:137-151) = MCTS.process_results (:230-289) on every slot with row
 * row_of_slot[slot] of policy_dev[rows, A] / value_dev[rows, P+1] (float32 probabilities). */
int  azg_backup(azg_engine *e, void *stream, const float *policy_dev, const float *value_dev,
                const int32_t *row_of_slot_dev, int flags);
/* azg_backup of simulation k followed by azg_select of simulation k + 1 in one launch (the two are adjacent in the loop and
 * touch the same trees); arguments as for the two calls, identical results. */
int  azg_backup_select(azg_engine *e, void *stream, const float *policy_dev, const float *value_dev,
                       const int32_t *row_of_slot_dev, int flags, void *obs_dev, int obs_dtype);
/* azg_backup (do_select = 0) or azg_backup_select (do_select = 1) fed with LOGITS: row r of logits_dev (stride logits_stride
 * floats) holds the A policy logits, then the P+1 value logits, of network row r -- the workspace azg_policy_value_heads_f16
 * fills; the two softmaxes of NNetArchitecture.py:112-118 run inside this launch.  Same results as softmax + azg_backup. */
int  azg_backup_select_logits(azg_engine *e, void *stream, const float *logits_dev, int logits_stride,
                              const int32_t *row_of_slot_dev, int flags, void *obs_dev, int obs_dtype, int do_select);
/* The same fed with the head FEATURES of a factorised-heads network: row r of feat_dev holds fp16 [2][feat_k] -- the 16 policy-
 * head and 16 value-head channels per pixel that azg_resnet_tower_features_f16 stores (index pixel * 16 + channel, zero padded
 * to feat_k = ceil(H*W*16 / 32) * 32).  The launch applies the collapsed Linear chains (NNetArchitecture.py:92-102) itself, and
 * only where process_results looks: the P+1 value logits and the policy logits of the leaf's VALID actions (the reference
 * masks and renormalises the policy, MCTS.pyx:239-245, so the other A - k logits never reach the tree; brandubh: ~40 rows of
 * the 588).  head_rows_dev: fp16 [A + P + 1][feat_k], row o = output o's weights over the policy (o < A) or value features;
 * head_b_dev: f32 [A + P + 1].  Priors equal softmax-over-all-A + mask + renormalise up to rounding (the full normaliser
 * cancels): ~1e-7 relative, not bit-identical to azg_backup_select_logits. */
int  azg_backup_select_features(azg_engine *e, void *stream, const void *feat_dev, int feat_k, const void *head_rows_dev,
                                const float *head_b_dev, const int32_t *row_of_slot_dev, int flags, void *obs_dev, int obs_dtype,
                                int do_select);
/* replace the engine's default root flags (azg_config.add_root_noise / add_root_temp) from now on -- what AZG_FLAGS_DEFAULT and the
 * persistent search launches use: MCTS.search(gs, nn, sims, add_root_noise, add_root_temp) passes them per call (MCTS.pyx:165) */
int  azg_set_root_flags(azg_engine *e, int flags /* AZG_FLAG_NOISE | AZG_FLAG_TEMP */);
#define AZG_FLAGS_DEFAULT (-1)   /* use azg_config.add_root_noise / add_root_temp                          */
#define AZG_FLAG_NOISE 1          /* process_results(..., add_root_noise, add_root_temp) per call (:230)    */
#define AZG_FLAG_TEMP  2
/* SelfPlayAgent.playMoves (:153-202): temperature, sample the move from MCTS.probs, record history,
 * MCTS.update_root + GameState.play_action, end-of-game bookkeeping (results, samples x symmetries, reset).
 * record_history = not fast (:161). */
int  azg_advance(azg_engine *e, void *stream, int record_history);
/* The same in two steps, for callers that share one games_played counter between several agents
 * (Coach with workers > 1: the `games_played < gamesPerIteration` test is made under the caller's lock, :179-183):
 *   azg_advance_begin  plays the moves and reports which slots finished: fin_host[B] = winstate bits (0 = running);
 *                      the finished slots still hold their final state (azg_get_states) until the commit.  blocking.
 *   azg_advance_commit counted_host[B] != 0 marks the finished games that count: results for every finished game,
 *                      samples + reset for the counted ones. */
int  azg_advance_begin(azg_engine *e, void *stream, int record_history, int32_t *fin_host);
int  azg_advance_commit(azg_engine *e, void *stream, const int32_t *counted_host);

/* ---- single-tree API (MCTS.pyx public methods; slot-wise) ------------------------------------------------ */
int  azg_root_counts(azg_engine *e, void *stream, int32_t *counts_dev /*[B, A]*/);                 /* MCTS.counts :297-303 */
int  azg_root_probs(azg_engine *e, void *stream, float temp, float *probs_dev /*[B, A]*/);        /* MCTS.probs  :308-329 */
int  azg_root_value(azg_engine *e, void *stream, int average, float *value_dev /*[B]*/);          /* MCTS.value  :331-344 */
/* MCTS.update_root(gs, a) (:185-195) on one slot's tree(s); the game state itself is NOT advanced. blocking. */
int  azg_update_root(azg_engine *e, void *stream, int slot, int action);
/* Node reclamation on demand -- the reference's counterpart is Python's GC freeing the siblings of the played move once
 * MCTS.update_root rebinds _root (alphazero/MCTS.pyx:185-195): copy the subtree under the root of the slot's tree(s) (slot < 0:
 * every slot) into the other semi-space and drop everything else -- what azg_advance / azg_update_root do by themselves once less
 * than one move's worth of nodes is free.  force = 0: only trees that are that full.  Must not be called between a find_leaf and its process_results
 * (azg_select .. azg_backup): the pending leaf's indices are void afterwards.  Results never change. */
int  azg_compact(azg_engine *e, void *stream, int slot, int force);
/* Snapshot of ONE slot's search state -- every tree of the slot (header with the root, the path of the last find_leaf, the live
 * nodes), the root and leaf game states and the slot's tape counter -- into / from a host buffer: what pickling an `MCTS` object
 * carries in the reference (alphazero/MCTS.pyx:8 `auto_pickle=True`: _root and its Node tree, _curnode, _path, depth, max_depth).
 * azg_slot_export(host_buf = NULL) returns the bytes needed; otherwise the bytes written (or < 0).  azg_slot_import needs an engine of
 * the same game / arena mode whose nodes_per_tree holds the snapshot's nodes (AZG_E_TREE_FULL otherwise).  A restored tree continues
 * bit for bit (node indices are relative to the semi-space base; the tape counter travels with it).  blocking. */
int64_t azg_slot_export(azg_engine *e, void *stream, int slot, void *host_buf, int64_t nbytes);
int  azg_slot_import(azg_engine *e, void *stream, int slot, const void *host_buf, int64_t nbytes);
/* children of a slot's root in list order (Node._children): a, n, q, p, v.  blocking; returns k or <0. */
int  azg_root_children(azg_engine *e, void *stream, int slot, int tree, int max_k, int32_t *a, int32_t *n, float *q, float *p, float *v);
/* children of an arbitrary node (node < 0: the root) incl. their node indices, for tree walks
 * (utils.plot_mcts_tree, alphazero/utils.py:57-83). blocking; returns k or <0.  player / e (optional, may be NULL; ABI v6): Node.player and
 * Node.e (MCTS.pyx:52,57) -- the player to move at the child and its win state as a bit per entry of the reference's uint8 vector, both
 * meaningful once the child has been expanded (the reference fills them in add_children's caller, :223-226). */
int  azg_node_children(azg_engine *e, void *stream, int slot, int tree, int node, int max_k, int32_t *idx, int32_t *a, int32_t *n, float *q, float *p, float *v,
                       int32_t *player, int32_t *e_bits);
/* MCTS.search resets max_depth at the start of every search (MCTS.pyx:168,179) */
int  azg_reset_max_depth(azg_engine *e, void *stream);
/* root header / search statistics of a slot's tree: n, q, v, player, e, depth, max_depth. blocking. */
int  azg_tree_info(azg_engine *e, void *stream, int slot, int tree, int32_t *out8);
/* path of the last find_leaf as actions (length = depth). blocking; returns depth or <0. */
int  azg_last_path(azg_engine *e, void *stream, int slot, int tree, int max_len, int32_t *actions);

/* ---- outputs ------------------------------------------------------------------------------------------------ */
int  azg_read_counters(azg_engine *e, void *stream, azg_counters *host_out);                      /* blocking */
/* device views of the accumulated examples, in the reference's output_queue order (Coach.py:364-386 layout):
 * obs f32 [n, C, H, W], pi f32 [n, A], z f32 [n, P+1] */
int  azg_examples_dev(azg_engine *e, float **obs_dev, float **pi_dev, float **z_dev);
/* stream-ordered device-to-device copy of examples [first, first+count) into caller-owned device buffers */
int  azg_copy_examples(azg_engine *e, void *stream, int first, int count, float *obs_dst, float *pi_dst, float *z_dst);
/* finished games in result_queue order: winstate u8 [n, P+1], turns i32 [n], slot i32 [n]  (host, blocking) */
int  azg_read_results(azg_engine *e, void *stream, int first, int count, uint8_t *winstate, int32_t *turns, int32_t *slot);
int  azg_clear_outputs(azg_engine *e, void *stream);      /* new iteration: games_played = 0, queues emptied */
/* per-slot action chosen by the last azg_advance (device [B]) */
int  azg_last_actions_dev(azg_engine *e, int32_t **actions_dev);

/* ---- network hot op: the policy/value ResNet on MFMA (csrc/azg_conv.h) --------------------------------------
 * alphazero/NNetArchitecture.py:36-120 in eval mode, BatchNorm folded.  Activations are fp16 NHWC rows; x: [boards*H*W, 8]
 * fp16 rows (the engine's obs_dtype 2); w_packed: MFMA fragment order [9 taps][KS = C/32][C/16][64 lanes][8 halves] per
 * convolution (nnet.pack_conv_weight), except the stem: [3 k-steps][C/16][64][8] with four taps of the 8 input channels per
 * k-step (nnet.pack_stem_weight); bias / pre_scale / pre_shift: f32 [C].  Board geometry from `game`. */

/* The whole residual tower (stem + 2*nblocks convolutions) in ONE persistent launch with activations resident in
 * LDS (csrc/azg_conv.h k_tower2), `channels` = 64 or 128 wide (C below).  x: [boards*H*W, 8] fp16; w_packed: stem fragments
 * then conv1, conv2 of every block, followed by AZG_TOWER_W_SLACK_KSTEPS = 18 k-steps (18 * C*32 halves) of readable slack -- the
 * weight prefetch ring runs past the last layer (the k-split tile: up to 16 k-steps); azg_tower_weights_size(C, nblocks) is the
 * number of halves every tower / search launch may read from w_packed; bias: f32 [1 + 2*nblocks][C] (stem, then b1, b2 per block); pre_scale/pre_shift: f32 [nblocks][C];
 * y: [boards*H*W, C] fp16 = the final residual stream (input of the collapsed heads GEMM).
 * The kernel evaluates 1, 2 or 4 boards per workgroup tile, chosen by `boards` (small batches: small tiles, more workgroups).
 * One-board tiles keep the layers' biases and affines in LDS ((12 * nblocks + 8) * channels bytes beside the image): a tower too deep
 * for that (about 90 blocks at 128 channels) is refused with AZG_E_INVALID_ARG at small batch sizes. */
#define AZG_TOWER_W_SLACK_KSTEPS 18
int64_t azg_tower_weights_size(int channels, int nblocks);      /* host only, no device needed */
int  azg_resnet_tower_f16(void *stream, int game, const void *x_dev, const void *w_packed_dev, const float *bias_dev,
                          const float *pre_scale_dev, const float *pre_shift_dev, void *y_dev, int boards, int nblocks,
                          int channels /* 128 or 64; every array above is sized by it instead of 128 */);
/* The tower with both heads fused behind it (A + NV <= 16): head_w_packed = the collapsed heads matrix
 * Wfull[H*W*128, A+NV] in MFMA fragment order [H*W][4][64 lanes][8 halves] (nnet.pack_head_weight), head_b f32[16];
 * writes softmax probabilities policy f32[boards, A] and value f32[boards, NV] -- what NNetWrapper.process returns
 * (alphazero/NNetWrapper.py:225-232) -- without storing the final stream. */
int  azg_resnet_policy_value_f16(void *stream, int game, const void *x_dev, const void *w_packed_dev, const float *bias_dev,
                                 const float *pre_scale_dev, const float *pre_shift_dev, int boards, int nblocks,
                                 const void *head_w_packed_dev, const float *head_b_dev, int A, int NV,
                                 float *policy_dev, float *value_dev);

/* Several models in ONE launch, each on a row range only the device knows -- the arena (Arena.pyx:262-281,
 * SelfPlayAgent.pyx:117-132): model m evaluates rows [sum(rows_per_model_dev[0..m)), + rows_per_model_dev[m]) of
 * x_dev / policy_dev / value_dev (whole-batch base pointers; rows_per_model_dev as written by azg_arena_rows) with
 * its own parameters w[m], bias[m], ... (host arrays of nmodels device pointers, nmodels <= 4, same architecture).
 * max_boards bounds the launch.  No host read of the split: rows -> select -> all models -> backup is one graph. */
int  azg_resnet_policy_value_multi_f16(void *stream, int game, const void *x_dev, int nmodels, const void *const *w_packed_dev,
                                       const float *const *bias_dev, const float *const *pre_scale_dev,
                                       const float *const *pre_shift_dev, int max_boards, int nblocks,
                                       const void *const *head_w_packed_dev, const float *const *head_b_dev, int A, int NV,
                                       float *policy_dev, float *value_dev, const int32_t *rows_per_model_dev);

/* `sims` whole simulations -- SelfPlayAgent.run's inner loop (:87-92): generateBatch, NNetWrapper.process, processBatch --
 * on every slot of the engine in ONE persistent launch: a workgroup owns four games, one wavefront each walks its tree
 * (azg_select's code), the leaf observations go straight into the tower's LDS image, the workgroup evaluates them
 * (azg_resnet_policy_value_f16's code) and each wavefront backs its game up (azg_backup's code, engine flags) from the
 * probabilities left in LDS.  Results are identical to `sims` x [azg_select, azg_resnet_policy_value_f16, azg_backup].
 * connect4 self-play engines with a 128-channel tower only (AZG_E_UNSUPPORTED otherwise); parameters as above.
 * sims == 0 only performs the one-time setup (call it once before capturing the launch in a graph). */
int  azg_search_f16(azg_engine *e, void *stream, const void *w_packed_dev, const float *bias_dev, const float *pre_scale_dev,
                    const float *pre_shift_dev, int nblocks, const void *head_w_packed_dev, const float *head_b_dev, int sims);

/* The same for networks with FACTORISED heads (wide action spaces): brandubh x 64 channels and the 3-player env x 32 channels, ONE
 * game per workgroup (the games of a CU then never wait for each other) of four wavefronts for brandubh (cout group x k group: two
 * waves of different games on every SIMD), two for the 3-player env.  Per simulation: one wavefront walks the game's tree and a
 * second prepares the priors and the shuffle, all of them run the tower on the leaf planes left in LDS, the 1x1 head convolutions
 * leave their features in LDS, and the next tree phase computes the logits it needs from them (azg_backup_select_features' code).  Results are
 * identical to `sims` x [azg_select / azg_backup_select_features, azg_resnet_tower_features_f16] + a final
 * azg_backup_select_features without select.  Parameters as those two functions take them; sims == 0: one-time setup only. */
int  azg_search_wide_f16(azg_engine *e, void *stream, const void *w_packed_dev, const float *bias_dev, const float *pre_scale_dev,
                         const float *pre_shift_dev, int nblocks, int channels, const void *head1_w_packed_dev,
                         const float *head1_b_dev, const void *head_rows_dev, const float *head_b_dev, int feat_k, int sims);

/* The same persistent launch with the BIT-EXACT hand-over between network and tree (NNetWrapper.py:225-232 -> MCTS.pyx:239-245): after
 * the head convolutions the launch computes ALL A + P+1 logits of its boards -- the collapsed Linear chains of
 * azg_policy_value_heads_fact_f16, same fragments, same summation order, so the same bits (wv_packed / head_b as there; wps_packed =
 * the policy chain's fragments SUBTILE-major, [ceil(A/16)][feat_k/32][64 lanes][8 halves], i.e. wp_packed with its first two axes
 * swapped: a wavefront of the launch streams one output subtile's fragments contiguously) --
 * and the next tree phase takes the softmax over all A, masks and renormalises.  Results are identical to `sims` x [azg_select /
 * azg_backup_select_logits, azg_resnet_tower_features_f16 + azg_policy_value_heads_fact_f16 (logits only)] + a final
 * azg_backup_select_logits without select, i.e. to the reference's evaluation fed NNetWrapper.process.  sims == 0: one-time setup only.
 * The engine's size picks the tile like azg_search_wide_f16 (games per workgroup). */
int  azg_search_wide_exact_f16(azg_engine *e, void *stream, const void *w_packed_dev, const float *bias_dev, const float *pre_scale_dev,
                               const float *pre_shift_dev, int nblocks, int channels, const void *head1_w_packed_dev,
                               const float *head1_b_dev, const void *wps_packed_dev, const void *wv_packed_dev, const float *head_b_dev,
                               int feat_k, int sims);

/* Bounds-checked builds (alphazero_general_amd/build.py --variant debug, -DAZG_DEBUG_BOUNDS; the reference compiles its own checks out,
 * MCTS.pyx:2-6): every node / child-block / path index of the tree kernels is checked before use; a violation raises the sticky
 * AZG_E_INTERNAL and is skipped.  *site = the first check that failed (0: none), *checked = 1 if this binary carries the checks. */
int  azg_debug_bounds_site(azg_engine *e, void *stream, int32_t *site, int32_t *checked);

/* The tile the persistent wide-head launches of `e` run with -- games per workgroup, picked per (device, game, tower width, heads, engine
 * size, depth): MEASURED at the launch's one-time set-up (sims == 0: every tile shape this game / width has is timed once, 24 + 24 simulations on
 * a scratch engine of the same size with the caller's network; a trial replaces the model's tile when it is more than 3 % faster; tile shape
 * changes no result), else a model derived from the device (CU
 * count, the occupancy query of each tile's kernel with its LDS at this depth) and the depth.  info12 = {games per workgroup, workgroups of a
 * launch, workgroups a CU holds at once, CUs, source (0 model, 1 measured, 2 forced by a tuning build), simulations per trial launch, 0, 0,
 * ns per trial launch at 1 / 2 / 3 / 4 games per workgroup (0: not measured)}.  AZG_E_INVALID_ARG before the first call of the launch for this engine size and depth.
 * (SelfPlayAgent.pyx:23-26: the batch size is whatever the caller made it; NNetArchitecture.py:78-84: depth is a free argument.) */
int  azg_search_wide_tile_info(azg_engine *e, int channels, int nblocks, int exact, int32_t *info12);

/* Collapsed heads for action spaces too wide to fuse behind the tower (A + NV > 16; brandubh: 588 + 3): the same
 * [k = H*W*C, A+NV] matrix applied to the final stream y [boards, k] fp16 that azg_resnet_tower_f16 stores, then the
 * two softmaxes.  head_w_packed: fragment order [k/32][OS = ceil((A+NV)/16)][64 lanes][8 halves], lane g*16+i, half j
 * = Wfull[ks*32 + g*8 + j, sub*16 + i] (zero beyond A+NV); head_b: f32[OS*16]; logits_ws: f32[boards * OS*16] scratch.
 * policy_dev = value_dev = NULL: stop at the logits (rows of stride OS*16 in logits_ws, for azg_backup_select_logits). */
int  azg_policy_value_heads_f16(void *stream, const void *y_dev, const void *head_w_packed_dev, const float *head_b_dev,
                                int boards, int k, int A, int NV, float *logits_ws_dev, float *policy_dev, float *value_dev);

/* The same heads FACTORISED the way the reference computes them (NNetArchitecture.py:88-102): stage 1 inside the tower launch --
 * the two 1x1 head convolutions (+ BatchNorm, folded), 16 policy and 16 value channels per pixel, written as feature rows
 * feat_dev[boards][2][feat_k] fp16 (policy half then value half, feature index pos * 16 + channel, feat_k = H*W*16 rounded up to
 * 32; the buffer must start zeroed, the padding is never written) instead of the final stream; head1_w_packed: fragments
 * [C/32][2][64 lanes][8 halves], lane g*16+i, half j = W1[out = ms*16 + i][cin = ks*32 + g*8 + j], rows 0-15 policy, 16-31 value;
 * head1_b: f32[32].  Stage 2 = azg_policy_value_heads_fact_f16: the collapsed Linear chains, policy logits from the policy half
 * (wp_packed [feat_k/32][ceil(A/16)][64][8], half j of lane g*16+i = Wp[k = ks*32 + g*8 + j][out = sub*16 + i]), value logits from
 * the value half (wv_packed [feat_k/32][64][8]); head_b / logits_ws / policy / value as azg_policy_value_heads_f16.  For 64
 * channels a quarter of the weight traffic of the fully collapsed matrix.  Needs 16 + 16 head channels. */
int  azg_resnet_tower_features_f16(void *stream, int game, const void *x_dev, const void *w_packed_dev, const float *bias_dev,
                                   const float *pre_scale_dev, const float *pre_shift_dev, int boards, int nblocks, int channels,
                                   const void *head1_w_packed_dev, const float *head1_b_dev, void *feat_dev, int feat_k);
int  azg_policy_value_heads_fact_f16(void *stream, const void *feat_dev, const void *wp_packed_dev, const void *wv_packed_dev,
                                     const float *head_b_dev, int boards, int feat_k, int A, int NV, float *logits_ws_dev,
                                     float *policy_dev, float *value_dev);

/* The sparse heads as their own launch: for every slot's last leaf (the one azg_select / azg_backup_select* left), the logits row
 * azg_backup_select_features computes internally -- the P + 1 value logits and the policy logits of the leaf's valid actions from
 * the board's head features, -inf for every other action -- into logits_dev[row][logits_stride] (A policy logits, then P + 1 value
 * logits; a terminal leaf, which takes no evaluation, gets zeros).  Same arithmetic, so azg_heads_softmax + azg_backup /
 * azg_backup_select_logits on these rows reproduce azg_backup_select_features bit for bit: the evaluation a leaf receives on
 * the sparse-heads path -- NNetWrapper.process (alphazero/NNetWrapper.py:225-232) restricted to the leaf's valid moves, which is
 * all MCTS.process_results keeps of it (alphazero/MCTS.pyx:239-245) -- exposed for callers and for parity tests against the oracle.  Arguments as azg_backup_select_features. */
int  azg_leaf_heads_sparse_f16(azg_engine *e, void *stream, const void *feat_dev, int feat_k, const void *head_rows_dev,
                               const float *head_b_dev, const int32_t *row_of_slot, float *logits_dev, int logits_stride);
/* exp(log_softmax) of NNetArchitecture.py:112-118 on rows of logits (A policy logits then NV value logits, stride
 * logits_stride): policy f32 [boards, A], value f32 [boards, NV] -- the second launch of azg_policy_value_heads_*_f16 by itself. */
int  azg_heads_softmax(void *stream, const float *logits_dev, int boards, int logits_stride, int A, int NV, float *policy_dev,
                       float *value_dev);

/* Host-side layout tables of one tower instantiation (no device needed; for tests and tooling): pixmap[NSUB*16] = pixel of
 * (subtile, lane & 15) or -1, qrow[ROWS] = padded LDS row of pixel p, info8 = {NSUB, ROWS, row stride B, tile rows, tile bytes,
 * padded width, lead rows, board stride}.  pixmap / qrow may be NULL.  AZG_E_UNSUPPORTED for a shape that is not instantiated. */
int  azg_tower_layout(int game, int boards_per_tile, int channels, int16_t *pixmap, int32_t *qrow, int32_t *info8);

/* ---- timing hooks for bench.py (HIP events on `stream` around the engine's own kernels) -------------------- */
int  azg_profile_enable(azg_engine *e, int on);
/* accumulated GPU ms + launch counts per kernel family: select, backup, advance. blocking. */
int  azg_profile_read(azg_engine *e, double *ms3, int64_t *launches3);

/* the same for the network launches of this process (they take no engine): GPU ms + launch counts per family: 0 = the tower
 * (azg_resnet_tower_f16 / _policy_value_f16 / _multi_f16), 1 = the wide heads GEMM (azg_policy_value_heads_f16, its softmax
 * launch excluded), 2 = the persistent search launch (azg_search_f16).  Events are recorded on the launch's own stream. */
int  azg_profile_net_enable(int on);
int  azg_profile_net_read(double *ms3, int64_t *launches3);                                        /* blocking */

/* ---- random tape (exposed so that host code can draw the agent-level numbers) ------------------------------- */
uint64_t azg_tape_u64(uint64_t seed, uint64_t stream, uint64_t ctr);
double   azg_tape_uniform(uint64_t seed, uint64_t stream, uint64_t ctr);
void     azg_tape_shuffle_pos(uint64_t seed, uint64_t stream, uint64_t ctr, int k, int32_t *pos);

/* The batched Arena's move as ONE persistent launch (Arena.pyx:208-328 + SelfPlayAgent.pyx arena mode :62-73,117-132): connect4 arena
 * engines, fused-heads 128-channel towers.  One game per workgroup: `sims` x [find_leaf on the MOVER's tree, the MOVER's model on MFMA,
 * process_results]; model of a game = p2i_host[mover] (player_to_index, SelfPlayAgent.pyx:44-47; host int32[P], read at launch) or, when
 * seat_of_slot_dev != NULL, the slot's own seating (4 bits per player, as azg_arena_rows_seats).  Model parameters as
 * azg_resnet_policy_value_multi_f16 takes them.  Results identical to `sims` x [azg_select / azg_backup_select with the row map of
 * azg_arena_rows, azg_resnet_policy_value_multi_f16] + a final azg_backup.  sims == 0: one-time setup only. */
int  azg_search_arena_f16(azg_engine *e, void *stream, int nmodels, const void *const *w_packed_dev, const float *const *bias_dev,
                          const float *const *pre_scale_dev, const float *const *pre_shift_dev, int nblocks, const void *const *head_w_packed_dev,
                          const float *const *head_b_dev, const int32_t *p2i_host, const uint32_t *seat_of_slot_dev, int sims);

/* Observation planes as callers of the reference hold them -- f32 [boards][C][H*W] (GameState.observation, the batch tensors of
 * Coach.py:294-300) -- into the tower's input rows [boards * H*W][8] fp16 (channels padded to 8), one launch.  obs_dev may be device
 * memory or page-locked host memory the device can read (a registered shared batch tensor: the H2D and the conversion are then one pass). */
int  azg_obs_to_nhwc8_f16(void *stream, const float *obs_dev, int boards, int channels, int hw, void *x_dev);

/* "Identical seeds", literally (second tier): Node.add_children shuffles with np.random.shuffle on numpy's global MT19937
 * stream (MCTS.pyx:76-79).  The engine's own shuffles come from the counter-based tape; this call makes the engine REPLAY recorded
 * permutations instead: ranks_host int16 [num_slots][len], where the rank (position in the shuffled list) of child i -- children in
 * ascending action order -- of the expansion that begins at tape counter c of a slot is ranks_host[slot][c + i] (the counter advances
 * by the number of children per expansion, so a slot's recorded permutations are simply concatenated in expansion order).  The
 * fixtures under tests/golden/c4_mt19937.npz were recorded from the reference running under np.random.seed(s) with np.random.shuffle
 * observed, not replaced.  ranks_host == NULL or len == 0: back to the counter-based tape.  An expansion past the end of the tape
 * raises the sticky AZG_E_INVALID_ARG.  Ranks outside [0, max_children) are refused here; a rank >= the expansion's k raises the sticky
 * AZG_E_INVALID_ARG on the device.  The device is drained before an old tape is freed; a hipGraph captured while a tape was set replays
 * with the tape it captured (re-capture after changing it).  Root noise is not replayed by this call: azg_set_random_tape. */
int  azg_set_shuffle_tape(azg_engine *e, void *stream, const int16_t *ranks_host, int len);

/* ALL THREE draws of a self-play game replayed (round 6: a whole SelfPlayAgent game under np.random.seed(s), move for move): besides the
 * shuffles, u_host double [num_slots][len] -- the uniform np.random.choice(A, p = policy) drew for the move made at tape counter c
 * (SelfPlayAgent.pyx:160; the legacy choice draws exactly one random_sample and searches the cdf: what azg_advance does with it) -- and the
 * Dirichlet noise of MCTS._add_root_noise (MCTS.pyx:197-206): noise_off_host int32 [num_slots][len] = offset into noise_pool_host (float32,
 * as :198-200 casts it; noise_len entries) of the vector mixed into the root's priors at counter c, one value per child in list order (-1:
 * no noise event at that counter).  One counter indexes all three tapes: a shuffle of k children advances it by k, a noise event and a move
 * by 1 each -- the order the reference makes the calls in for one game slot.  NULL u_host / noise_off_host: that draw stays on the
 * counter-based tape.  tests/golden/c4_mt19937_agent.npz was recorded from the reference's SelfPlayAgent running on numpy's own stream
 * with the three functions observed, not replaced. */
int  azg_set_random_tape(azg_engine *e, void *stream, const int16_t *ranks_host, const double *u_host, const int32_t *noise_off_host,
                         const float *noise_pool_host, int noise_len, int len);

#ifdef __cplusplus
}
#endif
#endif /* AZG_H */
