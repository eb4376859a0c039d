"""DeviceEngine: the Python face of one azg_engine (include/azg.h) -- B game slots and their search trees resident in
HBM on one GPU.  Tensors are torch CUDA(=HIP) tensors; every call is ordered on torch's current stream.

Reference counterparts: the state this object owns is what alphazero/SelfPlayAgent.pyx keeps in self.games /
self.mcts / self.histories (:31-59); select/backup/advance are generateBatch/processBatch/playMoves (:103-202).
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from .utils import default_temp_scaling, temp_table


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class DeviceEngine:
    def __init__(self, game, num_slots, *, arena=False, cpuct=1.25, fpu_reduction=0.2, root_noise_frac=0.1,
                 root_policy_temp=1.1, min_discount=1.0, add_root_noise=False, add_root_temp=False,
                 symmetric_samples=True, mcts_reset_threshold=0, games_per_iteration=1 << 30, start_temp=1.0,
                 arena_temp=0.25, temp_fn=default_temp_scaling, seed=0, slot_base=0, device=None,
                 nodes_per_tree=0, example_capacity=0, result_capacity=0, sims_hint=100, temp_table_override=None):
        self.L = _abi.lib()
        if not torch.cuda.is_available():
            raise RuntimeError('alphazero_general_amd needs a HIP device (no CPU fallback)')
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        self.game, self.gi = game, _abi.game_info(game)
        gi = self.gi
        self.B, self.arena = int(num_slots), bool(arena)
        self.A, self.NV, self.P = gi.action_size, gi.num_players + 1, gi.num_players
        self.obs_shape = (gi.obs_c, gi.obs_h, gi.obs_w)
        self.O = gi.obs_c * gi.obs_h * gi.obs_w
        self._tt = (np.ascontiguousarray(temp_table_override, np.float32) if temp_table_override is not None
                    else temp_table(temp_fn, start_temp, gi.max_turns))
        cfg = _abi.Config()
        # nodes_per_tree = 0: the library sizes the two node semi-spaces of a tree from sims_per_move (every simulation expands
        # at most one node = max_children stubs; dead siblings are reclaimed by compaction after a move)
        cfg.sims_per_move = max(int(sims_hint), 1)
        cfg.abi_version, cfg.game, cfg.device, cfg.num_slots = _abi.ABI_VERSION, game, self.device.index, self.B
        cfg.arena, cfg.nodes_per_tree = int(arena), int(nodes_per_tree)
        cfg.example_capacity, cfg.result_capacity = int(example_capacity), int(result_capacity)
        cfg.cpuct, cfg.fpu_reduction, cfg.root_noise_frac = cpuct, fpu_reduction, root_noise_frac
        cfg.root_policy_temp, cfg.min_discount = root_policy_temp, min_discount
        cfg.add_root_noise, cfg.add_root_temp = int(add_root_noise), int(add_root_temp)
        cfg.symmetric_samples, cfg.mcts_reset_threshold = int(symmetric_samples), int(mcts_reset_threshold or 0)
        cfg.games_per_iteration = int(games_per_iteration)
        cfg.start_temp, cfg.arena_temp = start_temp, arena_temp
        cfg.temp_table_len = len(self._tt)
        cfg.temp_table = self._tt.ctypes.data_as(C.POINTER(C.c_float))
        cfg.tape_seed, cfg.slot_base = int(seed), int(slot_base)
        self.seed, self.slot_base = int(seed), int(slot_base)
        self.example_capacity = int(example_capacity)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _abi.check(self.L.azg_engine_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self._row_of_slot = None
        info = (C.c_int32 * 8)()                                  # the sizes in effect (the library resolves the defaults)
        _abi.check(self.L.azg_engine_info(self.h, info))
        self._nodes_cap, self.compact_reserve = int(info[0]), int(info[1])

    def close(self):
        if getattr(self, 'h', None):
            self.L.azg_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ------------------------------------------------------------------------------------------
    def reset(self):
        _abi.check(self.L.azg_engine_reset(self.h, _stream()))

    def set_states(self, states, first=0, reset_trees=True):
        """states: list of (cells int8 array, player, turns)"""
        arr = _abi.states_array(len(states))
        for s, st in zip(arr, states):
            cells, player, turns = st[0], st[1], st[2]
            c = np.asarray(cells, np.int8).reshape(-1)
            for i, v in enumerate(c):
                s.cells[i] = int(v)
            s.player, s.turns = int(player), int(turns)
            s.aux[0] = int(st[3]) if len(st) > 3 else 0
        _abi.check(self.L.azg_set_states(self.h, _stream(), first, len(states), arr, int(reset_trees)))

    def _get(self, fn, first, count, full=False):
        count = self.B - first if count is None else count
        arr = _abi.states_array(count)
        _abi.check(fn(self.h, _stream(), first, count, arr))
        if full:
            return [(_abi.state_to_np(s, self.gi.cells), s.player, s.turns, s.aux[0]) for s in arr]
        return [(_abi.state_to_np(s, self.gi.cells), s.player, s.turns) for s in arr]

    def get_states(self, first=0, count=None):
        return self._get(self.L.azg_get_states, first, count)

    def get_leaf_states(self, first=0, count=None, full=False):
        """full=True adds the game-specific word aux[0] (brandubh: Board._king_captured) as a 4th element."""
        return self._get(self.L.azg_get_leaf_states, first, count, full)

    def get_states_full(self, first=0, count=None):
        return self._get(self.L.azg_get_states, first, count, True)

    def tape_counters(self):
        out = (C.c_uint64 * self.B)()
        _abi.check(self.L.azg_get_tape_counters(self.h, _stream(), 0, self.B, out))
        return np.array(out[:], np.uint64)

    def set_shuffle_tape(self, ranks):
        """replay RECORDED child shuffles instead of the counter-based tape (azg_set_shuffle_tape): ranks int16 [B, L], the rank of
        child i of the expansion that starts at tape counter c of slot s = ranks[s, c + i]; None: back to the counter-based tape."""
        if ranks is None:
            _abi.check(self.L.azg_set_shuffle_tape(self.h, _stream(), None, 0))
            return
        r = np.ascontiguousarray(ranks, np.int16)
        assert r.ndim == 2 and r.shape[0] == self.B
        _abi.check(self.L.azg_set_shuffle_tape(self.h, _stream(), r.ctypes.data_as(C.c_void_p), int(r.shape[1])))

    def set_random_tape(self, ranks, u=None, noise_off=None, noise_pool=None):
        """replay ALL recorded draws of a self-play game (azg_set_random_tape): ranks int16 [B, L] as set_shuffle_tape; u float64 [B, L] -- the
        uniform np.random.choice drew for the move made at tape counter c; noise_off int32 [B, L] + noise_pool float32 [n] -- the
        np.random.dirichlet vector mixed into the root priors at counter c starts at noise_pool[noise_off[s, c]] (-1: none)."""
        r = np.ascontiguousarray(ranks, np.int16)
        assert r.ndim == 2 and r.shape[0] == self.B
        uu = None if u is None else np.ascontiguousarray(u, np.float64)
        no = None if noise_off is None else np.ascontiguousarray(noise_off, np.int32)
        npool = None if noise_pool is None else np.ascontiguousarray(noise_pool, np.float32)
        assert uu is None or uu.shape == r.shape
        assert no is None or (no.shape == r.shape and npool is not None)
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        _abi.check(self.L.azg_set_random_tape(self.h, _stream(), vp(r), vp(uu), vp(no), vp(npool), 0 if npool is None else int(npool.size), int(r.shape[1])))

    def set_tape_counters(self, ctr, first=0):
        arr = (C.c_uint64 * len(ctr))(*[int(c) for c in ctr])
        _abi.check(self.L.azg_set_tape_counters(self.h, _stream(), first, len(ctr), arr))

    # ---- one simulation step ----------------------------------------------------------------------------
    def new_obs(self, dtype=torch.float32, rows=None):
        return torch.zeros((rows or self.B,) + self.obs_shape, dtype=dtype, device=self.device)

    def select(self, obs, row_of_slot=None):
        """find_leaf on every slot; leaf observations into obs[rows, C, H, W] (float32 or float16)."""
        dt = 0
        if obs is not None:
            assert obs.is_cuda and obs.is_contiguous()
            if obs.dim() == 3:                                   # [rows, H*W, 8] fp16: input format of the MFMA stem conv
                assert obs.dtype == torch.float16 and obs.shape[1:] == (self.gi.obs_h * self.gi.obs_w, 8)
                dt = 2
            else:
                assert obs.shape[1:] == self.obs_shape
                dt = {torch.float32: 0, torch.float16: 1}[obs.dtype]
        _abi.check(self.L.azg_select(self.h, _stream(), _ptr(obs), dt, _ptr(row_of_slot)))

    def arena_rows(self, player_to_index):
        if self._row_of_slot is None:
            self._row_of_slot = torch.zeros(self.B, dtype=torch.int32, device=self.device)
            self._rows_per_model = torch.zeros(self.P, dtype=torch.int32, device=self.device)
        p2i = (C.c_int32 * self.P)(*[int(x) for x in player_to_index])
        _abi.check(self.L.azg_arena_rows(self.h, _stream(), p2i, _ptr(self._row_of_slot), _ptr(self._rows_per_model)))
        return self._row_of_slot, self._rows_per_model

    def arena_rows_seats(self, seat_of_slot):
        """arena_rows with one seating per slot: seat_of_slot = int32 device tensor [B], 4 bits per player (model of player p)."""
        if self._row_of_slot is None:
            self._row_of_slot = torch.zeros(self.B, dtype=torch.int32, device=self.device)
            self._rows_per_model = torch.zeros(self.P, dtype=torch.int32, device=self.device)
        assert seat_of_slot.is_cuda and seat_of_slot.dtype == torch.int32 and seat_of_slot.numel() == self.B
        _abi.check(self.L.azg_arena_rows_seats(self.h, _stream(), _ptr(seat_of_slot), _ptr(self._row_of_slot), _ptr(self._rows_per_model)))
        return self._row_of_slot, self._rows_per_model

    def backup(self, policy, value, row_of_slot=None, add_root_noise=None, add_root_temp=None):
        assert policy.is_cuda and policy.dtype == torch.float32 and policy.is_contiguous() and policy.shape[1] == self.A
        assert value.is_cuda and value.dtype == torch.float32 and value.is_contiguous() and value.shape[1] == self.NV
        flags = -1 if add_root_noise is None and add_root_temp is None else (int(bool(add_root_noise)) | 2 * int(bool(add_root_temp)))
        _abi.check(self.L.azg_backup(self.h, _stream(), _ptr(policy), _ptr(value), _ptr(row_of_slot), flags))

    def backup_select(self, policy, value, obs, row_of_slot=None, add_root_noise=None, add_root_temp=None):
        """backup(policy, value) of this simulation + select(obs) of the next one in one launch."""
        assert policy.is_cuda and policy.dtype == torch.float32 and policy.is_contiguous() and policy.shape[1] == self.A
        assert value.is_cuda and value.dtype == torch.float32 and value.is_contiguous() and value.shape[1] == self.NV
        flags = -1 if add_root_noise is None and add_root_temp is None else (int(bool(add_root_noise)) | 2 * int(bool(add_root_temp)))
        dt = 0
        if obs is not None:
            assert obs.is_cuda and obs.is_contiguous()
            dt = 2 if obs.dim() == 3 else {torch.float32: 0, torch.float16: 1}[obs.dtype]
        _abi.check(self.L.azg_backup_select(self.h, _stream(), _ptr(policy), _ptr(value), _ptr(row_of_slot), flags, _ptr(obs), dt))

    def backup_select_logits(self, logits, obs=None, row_of_slot=None, select=True, add_root_noise=None, add_root_temp=None):
        """backup fed with LOGITS rows [rows, ld >= A + P + 1] (the softmaxes run inside the launch), optionally followed by the
        next simulation's select(obs) in the same launch."""
        assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous() and logits.shape[1] >= self.A + self.NV
        flags = -1 if add_root_noise is None and add_root_temp is None else (int(bool(add_root_noise)) | 2 * int(bool(add_root_temp)))
        dt = 0
        if obs is not None:
            assert obs.is_cuda and obs.is_contiguous()
            dt = 2 if obs.dim() == 3 else {torch.float32: 0, torch.float16: 1}[obs.dtype]
        _abi.check(self.L.azg_backup_select_logits(self.h, _stream(), _ptr(logits), int(logits.shape[1]), _ptr(row_of_slot), flags,
                                                   _ptr(obs), dt, int(bool(select))))

    def backup_select_features(self, feat, head_rows, head_b, obs=None, row_of_slot=None, select=True, add_root_noise=None,
                               add_root_temp=None):
        """backup fed with the head FEATURES of a factorised-heads network (rows [rows, 2 * feat_k] fp16): the launch computes the
        value logits and the policy logits of every leaf's valid actions itself (head_rows fp16 [A + P + 1, feat_k], head_b f32),
        optionally followed by the next simulation's select(obs)."""
        assert feat.is_cuda and feat.dtype == torch.float16 and feat.is_contiguous() and feat.shape[1] % 2 == 0
        fk = feat.shape[1] // 2
        assert head_rows.is_cuda and head_rows.dtype == torch.float16 and head_rows.is_contiguous() and tuple(head_rows.shape) == (self.A + self.NV, fk)
        assert head_b.is_cuda and head_b.dtype == torch.float32 and head_b.numel() >= self.A + self.NV
        flags = -1 if add_root_noise is None and add_root_temp is None else (int(bool(add_root_noise)) | 2 * int(bool(add_root_temp)))
        dt = 0
        if obs is not None:
            assert obs.is_cuda and obs.is_contiguous()
            dt = 2 if obs.dim() == 3 else {torch.float32: 0, torch.float16: 1}[obs.dtype]
        _abi.check(self.L.azg_backup_select_features(self.h, _stream(), _ptr(feat), int(fk), _ptr(head_rows), _ptr(head_b), _ptr(row_of_slot),
                                                     flags, _ptr(obs), dt, int(bool(select))))

    def leaf_heads_sparse(self, feat, head_rows, head_b, row_of_slot=None, out=None):
        """The sparse heads as their own launch: logits [rows, ld] f32 of every slot's last leaf -- A policy logits (-inf off the
        leaf's valid actions) then P + 1 value logits -- exactly what backup_select_features computes internally."""
        assert feat.is_cuda and feat.dtype == torch.float16 and feat.is_contiguous() and feat.shape[1] % 2 == 0
        fk = feat.shape[1] // 2
        assert head_rows.is_cuda and head_rows.dtype == torch.float16 and head_rows.is_contiguous() and tuple(head_rows.shape) == (self.A + self.NV, fk)
        ld = (self.A + self.NV + 15) // 16 * 16
        if out is None:
            out = torch.zeros((feat.shape[0], ld), dtype=torch.float32, device=self.device)
        _abi.check(self.L.azg_leaf_heads_sparse_f16(self.h, _stream(), _ptr(feat), int(fk), _ptr(head_rows), _ptr(head_b), _ptr(row_of_slot),
                                                    _ptr(out), int(out.shape[1])))
        return out

    def heads_softmax(self, logits):
        """(policy [rows, A], value [rows, P + 1]) probabilities of logits rows (azg_heads_softmax: the network's own softmax launch)."""
        pol = torch.empty((logits.shape[0], self.A), dtype=torch.float32, device=self.device)
        val = torch.empty((logits.shape[0], self.NV), dtype=torch.float32, device=self.device)
        _abi.check(self.L.azg_heads_softmax(_stream(), _ptr(logits), int(logits.shape[0]), int(logits.shape[1]), self.A, self.NV, _ptr(pol), _ptr(val)))
        return pol, val

    def advance(self, record_history=True):
        _abi.check(self.L.azg_advance(self.h, _stream(), int(bool(record_history))))

    def advance_begin(self, record_history=True):
        """playMoves up to the win test; returns fin[B] (winstate bits, 0 = still running).  Finished slots keep their
        final state until advance_commit."""
        fin = (C.c_int32 * self.B)()
        _abi.check(self.L.azg_advance_begin(self.h, _stream(), int(bool(record_history)), fin))
        return np.array(fin[:], np.int32)

    def advance_commit(self, counted):
        arr = (C.c_int32 * self.B)(*[int(bool(c)) for c in counted])
        _abi.check(self.L.azg_advance_commit(self.h, _stream(), arr))

    # ---- root statistics --------------------------------------------------------------------------------
    def root_counts(self):
        out = torch.zeros((self.B, self.A), dtype=torch.int32, device=self.device)
        _abi.check(self.L.azg_root_counts(self.h, _stream(), _ptr(out)))
        return out

    def root_probs(self, temp=1.0):
        out = torch.zeros((self.B, self.A), dtype=torch.float32, device=self.device)
        _abi.check(self.L.azg_root_probs(self.h, _stream(), float(temp), _ptr(out)))
        return out

    def root_value(self, average=False):
        out = torch.zeros(self.B, dtype=torch.float32, device=self.device)
        _abi.check(self.L.azg_root_value(self.h, _stream(), int(average), _ptr(out)))
        return out

    def update_root(self, slot, action):
        _abi.check(self.L.azg_update_root(self.h, _stream(), int(slot), int(action)))

    def set_search_flags(self, add_root_noise, add_root_temp):
        """the engine's default root flags from now on (used by backup(..., flags default) and by the persistent search launches)"""
        _abi.check(self.L.azg_set_root_flags(self.h, int(bool(add_root_noise)) | 2 * int(bool(add_root_temp))))

    def export_slot(self, slot=0):
        """bytes: snapshot of one slot's search state (azg_slot_export: trees, path, root / leaf state, tape counter)."""
        n = _abi.check(self.L.azg_slot_export(self.h, _stream(), int(slot), None, 0))
        buf = (C.c_char * n)()
        _abi.check(self.L.azg_slot_export(self.h, _stream(), int(slot), buf, n))
        return bytes(buf)

    def import_slot(self, blob, slot=0):
        _abi.check(self.L.azg_slot_import(self.h, _stream(), int(slot), C.c_char_p(blob), len(blob)))

    def compact(self, slot=-1, force=True):
        """reclaim the nodes outside the subtree under the root (slot -1: every slot).  Never between select and backup."""
        _abi.check(self.L.azg_compact(self.h, _stream(), int(slot), int(bool(force))))

    @property
    def nodes_per_tree(self):
        return int(self._nodes_cap)

    def root_children(self, slot, tree=0):
        K = max(self.gi.max_children, 1)
        a = (C.c_int32 * K)(); n = (C.c_int32 * K)(); q = (C.c_float * K)(); p = (C.c_float * K)(); v = (C.c_float * K)()
        k = _abi.check(self.L.azg_root_children(self.h, _stream(), slot, tree, K, a, n, q, p, v))
        return dict(a=np.array(a[:k], np.int32), n=np.array(n[:k], np.int32), q=np.array(q[:k], np.float32),
                    p=np.array(p[:k], np.float32), v=np.array(v[:k], np.float32))

    def node_children(self, slot, node=-1, tree=0):
        K = max(self.gi.max_children, 1)
        idx = (C.c_int32 * K)(); a = (C.c_int32 * K)(); n = (C.c_int32 * K)()
        q = (C.c_float * K)(); p = (C.c_float * K)(); v = (C.c_float * K)(); pl = (C.c_int32 * K)(); eb = (C.c_int32 * K)()
        k = _abi.check(self.L.azg_node_children(self.h, _stream(), slot, tree, int(node), K, idx, a, n, q, p, v, pl, eb))
        return [dict(idx=idx[i], a=a[i], n=n[i], q=q[i], p=p[i], v=v[i], player=pl[i], e=eb[i]) for i in range(k)]

    def reset_max_depth(self):
        _abi.check(self.L.azg_reset_max_depth(self.h, _stream()))

    def tree_info(self, slot, tree=0):
        o = (C.c_int32 * 8)()
        _abi.check(self.L.azg_tree_info(self.h, _stream(), slot, tree, o))
        f = np.array(o[:], np.int32)
        return dict(n=int(f[0]), q=float(f[1:2].view(np.float32)[0]), v=float(f[2:3].view(np.float32)[0]), player=int(f[3]),
                    e=int(f[4]), depth=int(f[5]), max_depth=int(f[6]), nodes_used=int(f[7]))

    def last_path(self, slot, tree=0):
        L = self.gi.max_turns + 2
        o = (C.c_int32 * L)()
        d = _abi.check(self.L.azg_last_path(self.h, _stream(), slot, tree, L, o))
        return np.array(o[:d], np.int32)

    # ---- outputs ----------------------------------------------------------------------------------------
    def counters(self):
        c = _abi.Counters()
        _abi.check(self.L.azg_read_counters(self.h, _stream(), C.byref(c)))
        if c.error == _abi.E_FLOATING_POINT:                              # what numpy raises in the reference (MCTS.pyx:23,320)
            raise FloatingPointError('invalid value encountered in divide: playMoves at a root without a visited child (numMCTSSims < 2) '
                                     'or a policy whose valid entries sum to 0')
        if c.error:
            raise _abi.AzgError(c.error, 'raised on device (tree arena or example buffer overflow / invalid action)')
        return dict(sims=c.sims, expansions=c.expansions, games_played=c.games_played, num_results=c.num_results,
                    num_examples=c.num_examples, max_nodes_used=c.max_nodes_used, max_nodes_kept=c.max_nodes_kept)

    def examples(self, first=0, count=None):
        """(obs [n,C,H,W], pi [n,A], z [n,P+1]) float32 device tensors, reference output_queue order."""
        if count is None:
            count = self.counters()['num_examples'] - first
        obs = torch.empty((count,) + self.obs_shape, dtype=torch.float32, device=self.device)
        pi = torch.empty((count, self.A), dtype=torch.float32, device=self.device)
        z = torch.empty((count, self.NV), dtype=torch.float32, device=self.device)
        _abi.check(self.L.azg_copy_examples(self.h, _stream(), first, count, _ptr(obs), _ptr(pi), _ptr(z)))
        return obs, pi, z

    def results(self, first=0, count=None):
        if count is None:
            count = self.counters()['num_results'] - first
        ws = np.zeros((max(count, 1), self.NV), np.uint8); turns = np.zeros(max(count, 1), np.int32); slot = np.zeros(max(count, 1), np.int32)
        _abi.check(self.L.azg_read_results(self.h, _stream(), first, count, ws.ctypes.data_as(C.POINTER(C.c_uint8)),
                                           turns.ctypes.data_as(C.POINTER(C.c_int32)), slot.ctypes.data_as(C.POINTER(C.c_int32))))
        return ws[:count], turns[:count], slot[:count]

    def clear_outputs(self):
        _abi.check(self.L.azg_clear_outputs(self.h, _stream()))

    def last_actions(self):
        p = C.c_void_p()
        _abi.check(self.L.azg_last_actions_dev(self.h, C.byref(p)))
        out = torch.empty(self.B, dtype=torch.int32, device=self.device)
        C.cdll.LoadLibrary  # noqa
        # D2D through torch: wrap via cuda array interface
        src = _DevArray(p.value, (self.B,), '<i4')
        out.copy_(torch.as_tensor(src, device=self.device))
        return out

    def bounds_site(self):
        """(site, checked): the first index check of a bounds-checked build that failed (0: none) and whether the loaded library carries
        the checks at all (build.py --variant debug)"""
        site, chk = C.c_int32(0), C.c_int32(0)
        _abi.check(self.L.azg_debug_bounds_site(self.h, _stream(), C.byref(site), C.byref(chk)))
        return int(site.value), bool(chk.value)

    def profile(self, on=True):
        _abi.check(self.L.azg_profile_enable(self.h, int(on)))
        self.profiling = bool(on)

    def profile_read(self):
        ms = (C.c_double * 3)(); n = (C.c_int64 * 3)()
        _abi.check(self.L.azg_profile_read(self.h, ms, n))
        return dict(select_ms=ms[0], backup_ms=ms[1], advance_ms=ms[2], select_n=n[0], backup_n=n[1], advance_n=n[2])


class _DevArray:
    """Minimal __cuda_array_interface__ view of an engine-owned device buffer."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)
