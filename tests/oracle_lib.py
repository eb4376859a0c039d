"""ctypes bindings to oracle/libazg_oracle.so -- the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), 'oracle')
_LIB = None

GAME_CONNECT4, GAME_BRANDUBH, GAME_TRIMOK = 0, 1, 2
MAX_PLAYERS = 4


class State(C.Structure):
    _fields_ = [('cells', C.c_int8 * 64), ('player', C.c_int32), ('turns', C.c_int32), ('aux', C.c_int32 * 4)]

    def copy(self):
        s = State()
        C.memmove(C.byref(s), C.byref(self), C.sizeof(State))
        return s

    def cells_np(self, n):
        return np.frombuffer(bytes(self.cells), dtype=np.int8)[:n].copy()


class GameInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('action_size', 'obs_c', 'obs_h', 'obs_w', 'num_players', 'has_draw',
                                          'max_turns', 'num_symmetries', 'cells')]


class MctsArgs(C.Structure):
    _fields_ = [('root_noise_frac', C.c_float), ('root_policy_temp', C.c_float), ('min_discount', C.c_float),
                ('fpu_reduction', C.c_float), ('cpuct', C.c_float), ('num_players_plus_draw', C.c_int32),
                ('tape_seed', C.c_uint64), ('tape_stream', C.c_uint64)]


class AgentArgs(C.Structure):
    _fields_ = [('mcts', MctsArgs), ('batch_size', C.c_int32), ('numMCTSSims', C.c_int32), ('numFastSims', C.c_int32),
                ('numWarmupSims', C.c_int32), ('probFastSim', C.c_float), ('gamesPerIteration', C.c_int32),
                ('add_root_noise', C.c_int32), ('add_root_temp', C.c_int32), ('symmetricSamples', C.c_int32),
                ('mctsResetThreshold', C.c_int32), ('startTemp', C.c_float), ('arenaTemp', C.c_float),
                ('temp_table_len', C.c_int32), ('temp_table', C.POINTER(C.c_float)), ('is_arena', C.c_int32),
                ('is_warmup', C.c_int32), ('arena_ref_misroute', C.c_int32), ('slot_base', C.c_uint64)]


def build():
    subprocess.run(['make', '-s', '-C', ORACLE_DIR], check=True)
    return os.path.join(ORACLE_DIR, 'libazg_oracle.so')


def _fp(dtype):
    return np.ctypeslib.ndpointer(dtype=dtype, flags='C_CONTIGUOUS')


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, 'libazg_oracle.so')
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    u64, i32, f32, f64, vp = C.c_uint64, C.c_int32, C.c_float, C.c_double, C.c_void_p
    SP = C.POINTER(State)
    sig = {
        'azo_game_info_get': (C.c_int, [C.c_int, C.POINTER(GameInfo)]),
        'azo_game_init': (None, [C.c_int, SP]),
        'azo_game_play': (C.c_int, [C.c_int, SP, C.c_int]),
        'azo_game_valid_moves': (None, [C.c_int, SP, _fp(np.uint8)]),
        'azo_game_win_state': (None, [C.c_int, SP, _fp(np.uint8)]),
        'azo_game_observation': (None, [C.c_int, SP, _fp(np.float32)]),
        'azo_game_symmetry': (None, [C.c_int, SP, _fp(np.float32), C.c_int, SP, _fp(np.float32)]),
        'azo_tape_u64': (u64, [u64, u64, u64]),
        'azo_tape_shuffle_pos': (None, [u64, u64, u64, C.c_int, _fp(np.int32)]),
        'azo_tape_choice': (C.c_int, [u64, u64, u64, _fp(np.float32), C.c_int]),
        'azo_tape_dirichlet': (None, [u64, u64, u64, C.c_int, f64, _fp(np.float64)]),
        'azo_tape_uniform': (f64, [u64, u64, u64]),
        'azo_tape_set_replay': (None, [u64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
        'azo_tape_clear_replay': (None, []),
        'azo_det_log': (f64, [f64]), 'azo_det_exp': (f64, [f64]),
        'azo_np_sum_f32': (f32, [_fp(np.float32), C.c_int]),
        'azo_np_pow_f32': (f32, [f32, f64]),
        'azo_fake_eval': (None, [u64, u64, u64, C.c_int, C.c_int, _fp(np.float32), _fp(np.float32)]),
        'azo_mcts_new': (vp, [C.POINTER(MctsArgs)]),
        'azo_mcts_free': (None, [vp]), 'azo_mcts_reset': (None, [vp]),
        'azo_mcts_tape_ctr': (u64, [vp]), 'azo_mcts_set_tape_ctr': (None, [vp, u64]),
        'azo_mcts_find_leaf': (C.c_int, [vp, C.c_int, SP, SP]),
        'azo_mcts_process_results': (None, [vp, C.c_int, _fp(np.float32), _fp(np.float32), C.c_int, C.c_int]),
        'azo_mcts_update_root': (C.c_int, [vp, C.c_int, SP, C.c_int]),
        'azo_mcts_counts': (None, [vp, C.c_int, _fp(np.int32)]),
        'azo_mcts_probs': (None, [vp, C.c_int, f32, _fp(np.float32)]),
        'azo_mcts_value': (f32, [vp, C.c_int]),
        'azo_mcts_raw_search': (None, [vp, C.c_int, SP, C.c_int, C.c_int, C.c_int]),
        'azo_mcts_root_n': (C.c_int, [vp]), 'azo_mcts_max_depth': (C.c_int, [vp]), 'azo_mcts_depth': (C.c_int, [vp]),
        'azo_mcts_root_children': (C.c_int, [vp, _fp(np.int32), _fp(np.int32), _fp(np.float32), _fp(np.float32), _fp(np.float32)]),
        'azo_mcts_last_path': (C.c_int, [vp, _fp(np.int32)]),
        'azo_mcts_root_header': (None, [vp, C.POINTER(i32), C.POINTER(f32), C.POINTER(f32), C.POINTER(i32), _fp(np.uint8)]),
        'azo_agent_new': (vp, [C.c_int, C.POINTER(AgentArgs)]), 'azo_agent_free': (None, [vp]),
        'azo_agent_begin_round': (C.c_int, [vp]),
        'azo_agent_generate_batch': (None, [vp, _fp(np.float32), _fp(np.int32), _fp(np.int32)]),
        'azo_agent_process_batch': (None, [vp, _fp(np.float32), _fp(np.float32)]),
        'azo_agent_play_moves': (C.c_int, [vp]),
        'azo_agent_games_played': (C.c_int, [vp]), 'azo_agent_num_samples': (C.c_int, [vp]),
        'azo_agent_num_results': (C.c_int, [vp]),
        'azo_agent_get_samples': (None, [vp, _fp(np.float32), _fp(np.float32), _fp(np.float32)]),
        'azo_agent_get_results': (None, [vp, _fp(np.uint8), _fp(np.int32), _fp(np.int32)]),
        'azo_agent_get_state': (None, [vp, C.c_int, SP]),
        'azo_agent_last_actions': (None, [vp, _fp(np.int32)]),
        'azo_agent_player_to_index': (C.POINTER(i32), [vp]),
        'azo_agent_mcts': (vp, [vp, C.c_int, C.c_int]),
        'azo_agent_sims_done': (u64, [vp]), 'azo_agent_expansions': (u64, [vp]),
        'azo_pool_new': (vp, [C.c_int, C.POINTER(AgentArgs), C.c_int]), 'azo_pool_free': (None, [vp]),
        'azo_pool_begin_round': (None, [vp]),
        'azo_pool_generate': (None, [vp, _fp(np.float32)]),
        'azo_pool_process': (None, [vp, _fp(np.float32), _fp(np.float32)]),
        'azo_pool_play': (C.c_int, [vp]),
        'azo_pool_run_tree_only': (f64, [vp, f64]),
        'azo_pool_expansions': (u64, [vp]), 'azo_pool_sims': (u64, [vp]), 'azo_pool_games_played': (C.c_int, [vp]),
        'azo_pool_agent': (vp, [vp, C.c_int]),
        'azo_pool_row_models': (None, [vp, _fp(np.int32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _LIB = L
    return L


def game_info(game):
    gi = GameInfo()
    assert lib().azo_game_info_get(game, C.byref(gi)) == 0
    return gi


# ---- thin pythonic wrappers --------------------------------------------------------------------------------
class OGame:
    """One game state held by the oracle."""

    def __init__(self, game, state=None):
        self.game, self.gi = game, game_info(game)
        self.s = State()
        if state is None:
            lib().azo_game_init(game, C.byref(self.s))
        else:
            self.s = state.copy()

    def clone(self):
        return OGame(self.game, self.s)

    def play(self, a):
        r = lib().azo_game_play(self.game, C.byref(self.s), int(a))
        if r != 0:
            raise ValueError('illegal action %d' % a)

    def valid_moves(self):
        v = np.zeros(self.gi.action_size, np.uint8)
        lib().azo_game_valid_moves(self.game, C.byref(self.s), v)
        return v

    def win_state(self):
        w = np.zeros(MAX_PLAYERS + 1, np.uint8)
        lib().azo_game_win_state(self.game, C.byref(self.s), w)
        return w[:self.gi.num_players + 1]

    def observation(self):
        o = np.zeros(self.gi.obs_c * self.gi.obs_h * self.gi.obs_w, np.float32)
        lib().azo_game_observation(self.game, C.byref(self.s), o)
        return o.reshape(self.gi.obs_c, self.gi.obs_h, self.gi.obs_w)

    def symmetry(self, pi, k):
        so = State()
        po = np.zeros(self.gi.action_size, np.float32)
        lib().azo_game_symmetry(self.game, C.byref(self.s), np.ascontiguousarray(pi, np.float32), k, C.byref(so), po)
        return OGame(self.game, so), po

    @property
    def player(self):
        return self.s.player

    @property
    def turns(self):
        return self.s.turns

    def cells(self):
        return self.s.cells_np(self.gi.cells)


def br_wide_positions(n, seed):
    """positions of the 7x7 tafl env in which the side to move has MORE THAN 64 legal moves (playouts from the start position stay
    at <= 63, tests/golden/br_rules.npz max_k; the rules allow up to 96): eight pieces of the side that moves first scattered
    over an almost empty board, the king and up to two of its men somewhere else."""
    rng = np.random.RandomState(seed)
    base = np.zeros(49, np.int8); base[[0, 6, 42, 48]] = 5; base[24] = 4        # escape corners, empty throne (cengine.pyx:24-32)
    free = np.flatnonzero(base == 0)
    out = []
    for _ in range(400000):
        if len(out) >= n:
            break
        c = base.copy()
        pick = rng.choice(free, 9 + rng.randint(0, 3), replace=False)
        c[pick[:8]] = 2; c[pick[8]] = 3; c[pick[9:]] = 1
        st = State()
        for i in range(49):
            st.cells[i] = int(c[i])
        g = OGame(GAME_BRANDUBH, state=st)
        if not g.win_state().any() and int(g.valid_moves().sum()) > 64:
            out.append(g)
    assert len(out) == n
    return out


def mcts_args(game, cpuct=1.25, fpu_reduction=0.2, root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1.0,
              seed=0, stream=0):
    gi = game_info(game)
    return MctsArgs(root_noise_frac, root_policy_temp, min_discount, fpu_reduction, cpuct,
                    gi.num_players + gi.has_draw, seed, stream)


class OMCTS:
    def __init__(self, game, **kw):
        self.game, self.gi = game, game_info(game)
        self.args = mcts_args(game, **kw)
        self.h = lib().azo_mcts_new(C.byref(self.args))

    def __del__(self):
        if getattr(self, 'h', None):
            lib().azo_mcts_free(self.h)
            self.h = None

    def find_leaf(self, g):
        leaf = State()
        exp = lib().azo_mcts_find_leaf(self.h, self.game, C.byref(g.s), C.byref(leaf))
        return OGame(self.game, leaf), exp

    def process_results(self, value, pi, noise=False, temp=False):
        v = np.zeros(MAX_PLAYERS + 1, np.float32)
        v[:len(value)] = value
        lib().azo_mcts_process_results(self.h, self.game, v, np.array(pi, np.float32), int(noise), int(temp))

    def update_root(self, g, a):
        if lib().azo_mcts_update_root(self.h, self.game, C.byref(g.s), int(a)) != 0:
            raise ValueError('Invalid action encountered while updating root')

    def counts(self):
        c = np.zeros(self.gi.action_size, np.int32)
        lib().azo_mcts_counts(self.h, self.game, c)
        return c

    def probs(self, temp=1.0):
        p = np.zeros(self.gi.action_size, np.float32)
        lib().azo_mcts_probs(self.h, self.game, temp, p)
        return p

    def value(self, average=False):
        return lib().azo_mcts_value(self.h, int(average))

    def raw_search(self, g, sims, noise=False, temp=False):
        lib().azo_mcts_raw_search(self.h, self.game, C.byref(g.s), sims, int(noise), int(temp))

    def root_children(self):
        return _root_children(self.h)

    def last_path(self):
        a = np.zeros(256, np.int32)
        n = lib().azo_mcts_last_path(self.h, a)
        return a[:n].copy()

    @property
    def root_n(self):
        return lib().azo_mcts_root_n(self.h)

    @property
    def max_depth(self):
        return lib().azo_mcts_max_depth(self.h)


def _root_children(h):
    a = np.zeros(1024, np.int32); n = np.zeros(1024, np.int32)
    q = np.zeros(1024, np.float32); p = np.zeros(1024, np.float32); v = np.zeros(1024, np.float32)
    k = lib().azo_mcts_root_children(h, a, n, q, p, v)
    return dict(a=a[:k].copy(), n=n[:k].copy(), q=q[:k].copy(), p=p[:k].copy(), v=v[:k].copy())


def temp_table(temp_fn, start_temp, max_turns):
    """temp_by_turn[t] = temp used for the move made at turn t (SelfPlayAgent.pyx:156-157 iterated)."""
    out, t = [], float(start_temp)
    for turn in range(max(int(max_turns or 0), 1) + 2):
        t = temp_fn(t, turn, max_turns)
        out.append(t)
    return np.array(out, np.float32)


def default_temp_scaling(cur_temp, turns, const_max_turns):
    """alphazero/utils.py:19-27 restated (scale 0.15, floor 0.2)."""
    if const_max_turns and (turns + 1) % int(0.15 * const_max_turns) == 0:
        return max(0.2, cur_temp / 2)
    return cur_temp


class OAgent:
    def __init__(self, game, batch_size, sims=25, games_per_iteration=32, seed=0, cpuct=1.25, fpu_reduction=0.2,
                 root_noise_frac=0.1, root_policy_temp=1.1, add_root_noise=False, add_root_temp=False,
                 symmetric=True, prob_fast=0.0, fast_sims=20, warmup_sims=5, start_temp=1.0, arena_temp=0.25,
                 temp_fn=default_temp_scaling, reset_threshold=0, is_arena=False, is_warmup=False,
                 ref_misroute=False, slot_base=0, _pool_of=0):
        self.game, self.gi = game, game_info(game)
        self.B = batch_size
        self._tt = temp_table(temp_fn, start_temp, self.gi.max_turns)
        a = AgentArgs()
        a.mcts = mcts_args(game, cpuct=cpuct, fpu_reduction=fpu_reduction, root_noise_frac=root_noise_frac,
                           root_policy_temp=root_policy_temp, seed=seed)
        a.batch_size = batch_size
        a.numMCTSSims, a.numFastSims, a.numWarmupSims = sims, fast_sims, warmup_sims
        a.probFastSim, a.gamesPerIteration = prob_fast, games_per_iteration
        a.add_root_noise, a.add_root_temp, a.symmetricSamples = int(add_root_noise), int(add_root_temp), int(symmetric)
        a.mctsResetThreshold = reset_threshold
        a.startTemp, a.arenaTemp = start_temp, arena_temp
        a.temp_table_len = len(self._tt)
        a.temp_table = self._tt.ctypes.data_as(C.POINTER(C.c_float))
        a.is_arena, a.is_warmup, a.arena_ref_misroute, a.slot_base = int(is_arena), int(is_warmup), int(ref_misroute), slot_base
        self.args = a
        self.O = self.gi.obs_c * self.gi.obs_h * self.gi.obs_w
        self.h = None if _pool_of else lib().azo_agent_new(game, C.byref(a))

    def __del__(self):
        if getattr(self, 'h', None):
            lib().azo_agent_free(self.h)
            self.h = None

    def begin_round(self):
        return lib().azo_agent_begin_round(self.h)

    def generate_batch(self):
        obs = np.zeros((self.B, self.O), np.float32)
        rg = np.zeros(self.B, np.int32); rm = np.zeros(self.B, np.int32)
        lib().azo_agent_generate_batch(self.h, obs, rg, rm)
        return obs.reshape(self.B, self.gi.obs_c, self.gi.obs_h, self.gi.obs_w), rg, rm

    def process_batch(self, policy, value):
        lib().azo_agent_process_batch(self.h, np.ascontiguousarray(policy, np.float32), np.ascontiguousarray(value, np.float32))

    def play_moves(self):
        return lib().azo_agent_play_moves(self.h)

    @property
    def games_played(self):
        return lib().azo_agent_games_played(self.h)

    def samples(self):
        n = lib().azo_agent_num_samples(self.h)
        obs = np.zeros((n, self.O), np.float32); pi = np.zeros((n, self.gi.action_size), np.float32)
        z = np.zeros((n, self.gi.num_players + 1), np.float32)
        if n:
            lib().azo_agent_get_samples(self.h, obs, pi, z)
        return obs.reshape(n, self.gi.obs_c, self.gi.obs_h, self.gi.obs_w), pi, z

    def results(self):
        n = lib().azo_agent_num_results(self.h)
        ws = np.zeros((n, self.gi.num_players + 1), np.uint8); turns = np.zeros(n, np.int32); slot = np.zeros(n, np.int32)
        if n:
            lib().azo_agent_get_results(self.h, ws, turns, slot)
        return ws, turns, slot

    def state(self, slot):
        s = State()
        lib().azo_agent_get_state(self.h, slot, C.byref(s))
        return OGame(self.game, s)

    def last_actions(self):
        a = np.zeros(self.B, np.int32)
        lib().azo_agent_last_actions(self.h, a)
        return a

    def player_to_index(self):
        p = lib().azo_agent_player_to_index(self.h)
        return [p[i] for i in range(self.gi.num_players)]

    def root_children(self, slot, player=0):
        return _root_children(lib().azo_agent_mcts(self.h, slot, player))

    def root_n(self, slot, player=0):
        return lib().azo_mcts_root_n(lib().azo_agent_mcts(self.h, slot, player))

    @property
    def sims_done(self):
        return lib().azo_agent_sims_done(self.h)

    @property
    def expansions(self):
        return lib().azo_agent_expansions(self.h)


class OPool:
    """n oracle agents of `batch_size` games, one host thread each (oracle/azg_pool_ref.c): the reference's `workers` agent
    processes (Coach.py:291-342) for the C restatement.  Agent i owns the global slots [slot_base + i * B, + B).  Keyword
    arguments as OAgent."""

    def __init__(self, game, n_agents, batch_size, **kw):
        tmpl = OAgent(game, batch_size, _pool_of=1, **kw)                 # (only builds the argument block)
        self._keep = tmpl
        self.game, self.gi, self.n, self.B, self.O = game, tmpl.gi, int(n_agents), int(batch_size), tmpl.O
        self.A, self.NV = self.gi.action_size, self.gi.num_players + 1
        self.h = lib().azo_pool_new(game, C.byref(tmpl.args), self.n)
        self.obs = np.zeros((self.n * self.B, self.O), np.float32)

    def __del__(self):
        if getattr(self, 'h', None):
            lib().azo_pool_free(self.h)
            self.h = None

    def begin_round(self):
        lib().azo_pool_begin_round(self.h)

    def generate(self):
        """leaf observations of every agent's games: [n * B, C, H, W] (a view of one reused buffer)"""
        lib().azo_pool_generate(self.h, self.obs)
        return self.obs.reshape(self.n * self.B, self.gi.obs_c, self.gi.obs_h, self.gi.obs_w)

    def process(self, policy, value):
        lib().azo_pool_process(self.h, np.ascontiguousarray(policy, np.float32), np.ascontiguousarray(value, np.float32))

    def play(self):
        return lib().azo_pool_play(self.h)

    def row_models(self):
        """arena mode: which model evaluates each row of the last generate() ([n * B] int32)"""
        out = np.zeros(self.n * self.B, np.int32)
        lib().azo_pool_row_models(self.h, out)
        return out

    def run_tree_only(self, seconds):
        """every agent plays whole rounds with the uniform evaluator on its own thread for `seconds`; returns the wall time"""
        return lib().azo_pool_run_tree_only(self.h, float(seconds))

    def agent_last_actions(self, i):
        a = np.zeros(self.B, np.int32)
        lib().azo_agent_last_actions(lib().azo_pool_agent(self.h, i), a)
        return a

    # ---- whole-pool views in GLOBAL slot order (agent i owns slots [i * B, (i + 1) * B)) ----
    def last_actions(self):
        return np.concatenate([self.agent_last_actions(i) for i in range(self.n)])

    def root_counts(self, slots=None):
        """int32 [n * B, A] visit counts of every game's root children (MCTS.counts), or of the listed global slots only"""
        L = lib()
        slots = range(self.n * self.B) if slots is None else slots
        out = np.zeros((len(slots), self.A), np.int32)
        for r, gs in enumerate(slots):
            ch = _root_children(L.azo_agent_mcts(L.azo_pool_agent(self.h, gs // self.B), gs % self.B, 0))
            out[r, ch['a']] = ch['n']
        return out

    def root_probs(self, slots, temp=1.0):
        L = lib()
        out = np.zeros((len(slots), self.A), np.float32)
        for r, gs in enumerate(slots):
            L.azo_mcts_probs(L.azo_agent_mcts(L.azo_pool_agent(self.h, gs // self.B), gs % self.B, 0), self.game, temp, out[r])
        return out

    def samples(self):
        """(obs, pi, z) of every agent, agent after agent (each agent's own output_queue order)"""
        L = lib()
        obs, pi, z = [], [], []
        for i in range(self.n):
            ah = L.azo_pool_agent(self.h, i)
            n = L.azo_agent_num_samples(ah)
            o = np.zeros((n, self.O), np.float32); p = np.zeros((n, self.A), np.float32); zz = np.zeros((n, self.NV), np.float32)
            if n:
                L.azo_agent_get_samples(ah, o, p, zz)
            obs.append(o); pi.append(p); z.append(zz)
        return np.concatenate(obs), np.concatenate(pi), np.concatenate(z)

    def results(self):
        """(winstate, turns, global slot) of every finished game, agent after agent"""
        L = lib()
        ws, turns, slot = [], [], []
        for i in range(self.n):
            ah = L.azo_pool_agent(self.h, i)
            n = L.azo_agent_num_results(ah)
            w = np.zeros((max(n, 1), self.NV), np.uint8); t = np.zeros(max(n, 1), np.int32); s = np.zeros(max(n, 1), np.int32)
            if n:
                L.azo_agent_get_results(ah, w, t, s)
            ws.append(w[:n]); turns.append(t[:n]); slot.append(s[:n] + i * self.B)
        return np.concatenate(ws), np.concatenate(turns), np.concatenate(slot)

    @property
    def expansions(self):
        return lib().azo_pool_expansions(self.h)

    @property
    def sims_done(self):
        return lib().azo_pool_sims(self.h)

    @property
    def games_played(self):
        return lib().azo_pool_games_played(self.h)


def fake_eval(seed, slot, sim, A, nv):
    p = np.zeros(A, np.float32); v = np.zeros(nv, np.float32)
    lib().azo_fake_eval(seed, slot, sim, A, nv, p, v)
    return p, v
