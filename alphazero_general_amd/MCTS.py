"""`MCTS` with the public surface of the reference's Cython class (alphazero/MCTS.pyx:119-344): same constructor
argument (the args dotdict), same method names / argument meaning / errors, backed by a one-slot device engine
(include/azg.h).  Used by GenericPlayers.MCTSPlayer / RawMCTSPlayer (GenericPlayers.py:100-200) and anything else
that drives one tree at a time; the batched path (SelfPlayAgent) talks to a B-slot engine directly.

find_leaf / process_results are one kernel launch on a single wavefront plus a host sync each; `search(gs, nn, ...)` with this
package's NNetWrapper runs all its simulations in ONE persistent launch (azg_search_f16 / azg_search_wide_f16) where the network has
one.  The throughput path is `alphazero_general_amd.selfplay.SelfPlayRunner` / `SelfPlayAgent`.  Objects pickle (MCTS.pyx:8):
parameters, tape seed and a snapshot of the tree (azg_slot_export), so an MCTSPlayer can cross a process boundary.
"""
import os

import numpy as np
import torch

from . import _abi
from .engine import DeviceEngine
from .Game import azg_game_id, has_device_rules

NOISE_ALPHA_RATIO = 10.83          # MCTS.pyx:20
_DRAW_VALUE = 0.5                  # MCTS.pyx:21


def encode_state(gs):
    """GameState object -> (cells int8, player, turns) of include/azg.h azg_state."""
    if hasattr(gs, 'to_azg_state'):
        return gs.to_azg_state()
    gid = azg_game_id(gs)
    if gid == 0:                   # the reference's own connect4 Game (envs/connect4/connect4.pyx:20-40)
        return np.asarray(gs._board.pieces, np.int8).reshape(-1), gs.player, gs.turns
    if gid == 1:                   # the reference's own brandubh Game (envs/brandubh/fastafl.pyx:121-131; fastafl/cengine.pyx:24-32,59)
        return np.asarray(gs._board._state, np.int8).reshape(-1), gs.player, gs.turns, int(gs._board._king_captured)
    raise NotImplementedError('cannot encode %r for the device engine' % type(gs))


def decode_state(template, cells, player, turns, aux0=0):
    """Build a GameState of template's class from an azg_state."""
    cls = type(template)
    if hasattr(cls, 'from_azg_state'):
        try:
            return cls.from_azg_state(cells, player, turns, aux0)
        except TypeError:
            return cls.from_azg_state(cells, player, turns)
    g = template.clone()
    if hasattr(g._board, '_state'):                            # the reference's brandubh Game: fastafl Board (boardgame/board.pxd:31-40)
        shape = np.asarray(template._board._state).shape
        g._board._state = np.asarray(cells, np.uint8).reshape(shape).copy()
        g._board.num_turns = int(turns)
        g._board._king_captured = bool(aux0)
        g._board._king_escaped = False
        g._board._king_escaped = bool(g._board.king_escaped())      # (the flag Board.move leaves behind, cengine.pyx:144-148,271)
    else:
        g._board.pieces = np.asarray(cells, np.intc).reshape(np.asarray(template._board.pieces).shape).copy()
    g._player, g._turns = int(player), int(turns)
    return g


class Node:
    """Read-only view of one tree node with the reference's attribute names (MCTS.pyx:49-57): _children, a, q, n, v, p and -- as the
    reference's Node -- e (uint8 vector of length args._num_players: the node's win state, zeros until the node is expanded, :64-66,223-226)
    and player (the player to move there, 0 until expanded)."""

    def __init__(self, mcts, idx, a=-1, n=0, q=0.0, p=0.0, v=0.0, player=0, e_bits=0):
        self._mcts, self._idx = mcts, idx
        self.a, self.n, self.q, self.p, self.v = a, n, q, p, v
        self.player = int(player)
        self.e = np.array([(int(e_bits) >> j) & 1 for j in range(int(getattr(mcts, '_num_players', 0) or 0))], np.uint8)

    @property
    def _children(self):
        e = self._mcts._engine
        if e is None:
            return []
        return [Node(self._mcts, c['idx'], c['a'], c['n'], c['q'], c['p'], c['v'], c['player'], c['e'])
                for c in e.node_children(0, self._idx)]

    def __repr__(self):
        return 'Node(a={}, q={}, v={}, n={}, p={})'.format(self.a, self.q, self.v, self.n, self.p)


NODE_STORE_BUDGET = 256 << 20      # default ceiling of ONE MCTS object's node store (two semi-spaces of 32-byte nodes), bytes


def _rebuild_mcts(params, game, blob, depth, max_depth, nodes_used):
    """unpickle (MCTS.__reduce__): same parameters and tape seed; the tree, if there was one, is restored into a fresh engine"""
    from .utils import dotdict
    m = MCTS(dotdict(params))
    m.depth, m._max_depth = depth, max_depth
    if blob is not None:
        m._ensure_game(game)
        m._engine.import_slot(blob)
        m._nodes_used = nodes_used
    return m


def _rebuild_ref_mcts(args, blob):
    import pickle
    from . import reference_module
    from .utils import dotdict
    with reference_module('MCTS'):
        ref = pickle.loads(blob)
    m = MCTS(dotdict(args))
    m._ref, m.depth, m._max_depth = ref, ref.depth, ref.max_depth
    return m


class MCTS:
    def __init__(self, args):
        self.root_noise_frac = args.root_noise_frac            # MCTS.pyx:134-139
        self.root_temp = args.root_policy_temp
        self.min_discount = args.min_discount
        self.fpu_reduction = args.fpu_reduction
        self.cpuct = args.cpuct
        self._num_players = args._num_players
        get = args.get if hasattr(args, 'get') else (lambda k, d=None: getattr(args, k, d))
        self._seed = int(get('_azg_seed', None) if get('_azg_seed', None) is not None else int.from_bytes(os.urandom(7), 'little'))
        self._sims_hint = int(get('numMCTSSims', 100) or 100)
        self._nodes_per_tree = int(get('_azg_nodes_per_tree', 0) or 0)       # 0: a whole game's worth, at most NODE_STORE_BUDGET
        self._args = args
        self._ref = None                                       # the reference's own MCTS, for a game without device rules (_fallback)
        self._engine = None
        self._game = None
        self._leaf_template = None
        self._nodes_used = 0
        self._compact_futile = False
        self.depth = 0
        self._max_depth = 0

    # max_depth is a public, WRITABLE attribute of the reference's class (MCTS.pyx:130: `cdef public int max_depth`); Evaluator.py:343 resets
    # it from outside (`self._mcts.max_depth = 0`) before a search of its own made of find_leaf / process_results calls.  Semantics here:
    # assigning 0 resets the device-side maximum as well (the one use the reference's callers make of the setter); any other value is
    # kept on this object only until the next find_leaf / search, whose result -- the device's running maximum -- replaces it, exactly
    # as the reference's find_leaf would overwrite an assigned value that a deeper path exceeds
    @property
    def max_depth(self):
        return self._max_depth

    @max_depth.setter
    def max_depth(self, v):
        self._max_depth = int(v)
        if self._ref is not None:
            self._ref.max_depth = int(v)
        elif self._engine is not None and int(v) == 0:
            self._engine.reset_max_depth()

    # ---- pickling (MCTS.pyx:8 auto_pickle=True: the reference pickles _root with its whole Node tree, _curnode, _path, depth, max_depth)
    def __reduce__(self):
        if self._ref is not None:                              # (the reference's object travels as its own pickle, made where its
            import pickle                                      #  module is the one the names resolve to: reference_module)
            from . import reference_module
            with reference_module('MCTS'):
                blob = pickle.dumps(self._ref, pickle.HIGHEST_PROTOCOL)
            return _rebuild_ref_mcts, (dict(self._args), blob)
        params = dict(root_noise_frac=self.root_noise_frac, root_policy_temp=self.root_temp, min_discount=self.min_discount,
                      fpu_reduction=self.fpu_reduction, cpuct=self.cpuct, _num_players=self._num_players, _azg_seed=self._seed,
                      numMCTSSims=self._sims_hint, _azg_nodes_per_tree=self._nodes_per_tree)
        blob = self._engine.export_slot(0) if self._engine is not None else None
        return _rebuild_mcts, (params, self._game, blob, self.depth, self.max_depth, self._nodes_used)

    # ---- games without device rule kernels: the reference's own class does the search (reference side) ----
    def _fallback(self, gs=None):
        if self._ref is not None or gs is None or self._engine is not None or has_device_rules(gs):
            return self._ref
        from . import reference_class
        cls = reference_class('MCTS')
        if cls is not None:                                    # (else: azg_game_id raises the NotImplementedError below)
            self._ref = cls(self._args)
        return self._ref

    def _via_ref(self, ref, name, *a):
        out = getattr(ref, name)(*a)
        self.depth, self._max_depth = ref.depth, ref.max_depth
        return out

    # ---- engine plumbing ----
    def _ensure(self, gs):
        return self._ensure_game(azg_game_id(gs))

    def _ensure_game(self, gid):
        if self._engine is None:
            self._game = gid
            # ONE tree: its node store holds a whole game's worth of expansions (max_turns moves x the simulations per move x
            # max_children) -- more than a search that drops nothing could fill -- but at most NODE_STORE_BUDGET bytes (several MCTS
            # objects live side by side: one per arena player, one per agent): connect4 2.4 MB; brandubh at 200 simulations would be
            # 123 MB, at 1600 simulations it is clamped to 256 MB (13 moves' worth).  args._azg_nodes_per_tree overrides.  A caller
            # that keeps searching at ONE root is served by the forced compaction in find_leaf / search until the LIVE subtree
            # itself exceeds the store (AZG_E_TREE_FULL; the reference's limit there is host memory)
            gi = _abi.game_info(gid)
            sims = max(self._sims_hint, 200)
            cap = self._nodes_per_tree or min(max(gi.max_turns, 16) * sims * gi.max_children + 64, NODE_STORE_BUDGET // 64)
            cap = min(cap, (1 << 28) - 1)
            self._engine = DeviceEngine(gid, 1, cpuct=self.cpuct, fpu_reduction=self.fpu_reduction,
                                        root_noise_frac=self.root_noise_frac, root_policy_temp=self.root_temp,
                                        min_discount=self.min_discount, seed=self._seed, sims_hint=sims, nodes_per_tree=cap)
            self._max_children = gi.max_children
        elif gid != self._game:
            raise ValueError('this MCTS object was created for another game')
        return self._engine

    def _make_room(self, need):
        """forced compaction before a find_leaf / search that may not fit: drop what the root no longer reaches.  When the last
        forced compaction reclaimed next to nothing the LIVE subtree fills the store: do not thrash, let AZG_E_TREE_FULL surface."""
        e = self._engine
        if self._nodes_used + need <= e.nodes_per_tree or self._compact_futile:
            return
        before = self._nodes_used
        e.compact(0, force=True)
        self._nodes_used = e.tree_info(0)['nodes_used']
        self._compact_futile = before - self._nodes_used < 2 * self._max_children

    def _sync_root_state(self, gs):
        self._engine.set_states([encode_state(gs)], reset_trees=False)

    def reset(self):                                           # MCTS.pyx:154-160
        if self._ref is not None:
            return self._via_ref(self._ref, 'reset')
        if self._engine is not None:
            self._engine.reset()
        self.depth = self._max_depth = 0
        self._nodes_used, self._compact_futile = 0, False

    def __repr__(self):
        return 'MCTS(root_noise_frac={}, root_temp={}, min_discount={}, fpu_reduction={}, cpuct={}, _num_players={}, depth={}, max_depth={})' \
            .format(self.root_noise_frac, self.root_temp, self.min_discount, self.fpu_reduction, self.cpuct,
                    self._num_players, self.depth, self.max_depth)

    # ---- public API ----
    def search(self, gs, nn, sims, add_root_noise, add_root_temp):     # MCTS.pyx:165-173
        ref = self._fallback(gs)
        if ref is not None:
            return self._via_ref(ref, 'search', gs, nn, sims, add_root_noise, add_root_temp)
        e = self._ensure(gs)
        e.reset_max_depth()
        hip = self._persistent_net(nn, e)
        if hip is not None and sims > 0:
            # `nn` is this package's NNetWrapper and a persistent search launch exists for (game, network): all `sims` simulations
            # -- find_leaf, the network on MFMA, process_results -- in ONE launch instead of 3 launches + a host sync each
            # (GenericPlayers.py:133-134 calls this once per move).  Same trees as the loop below, bit for bit: a board's evaluation does
            # not depend on the launch form (wide-head networks: the exact launch, all A + P+1 logits inside it -- include/azg.h)
            self._sync_root_state(gs)
            e.set_search_flags(add_root_noise, add_root_temp)      # (per call; the engine's own defaults are restored below)
            try:
                left = int(sims)
                while left > 0:
                    # a launch cannot compact in the middle: it gets as many simulations as are sure to fit the store (each adds at most
                    # max_children nodes), the dropped siblings are reclaimed between launches -- the same tree, launch by launch
                    self._make_room(left * self._max_children)
                    fit = (e.nodes_per_tree - self._nodes_used) // self._max_children
                    n = min(left, fit) if fit >= 1 else left       # (not one expansion fits even after a compaction: let AZG_E_TREE_FULL surface)
                    hip.search(e, n, exact=True)
                    left -= n
                    info = e.tree_info(0)
                    self._nodes_used = info['nodes_used']
            finally:
                e.set_search_flags(False, False)                   # DeviceEngine's defaults for this class's one-slot engine (_ensure_game)
            self.depth, self._max_depth = info['depth'], info['max_depth']
            return
        for _ in range(sims):
            leaf = self.find_leaf(gs)
            p, v = nn(leaf.observation())
            self.process_results(leaf, v, p, add_root_noise, add_root_temp)

    @staticmethod
    def _persistent_net(nn, e):
        """the HipResNet behind `nn` if nn is an NNetWrapper (or its bound predict / __call__) whose network has a persistent search
        launch for this engine's game on this engine's device, else None"""
        from .nnet import NNetWrapper
        w = nn if isinstance(nn, NNetWrapper) else getattr(nn, '__self__', None)
        if not isinstance(w, NNetWrapper) or not w.fast or w.device.type != 'cuda' or w.device.index not in (None, e.device.index):
            return None
        if w._infer is None:
            w.refresh()
        hip = w._hip
        return hip if (hip is not None and hip.can_search and hip.game == e.game) else None

    def raw_search(self, gs, sims, add_root_noise, add_root_temp):     # MCTS.pyx:175-183
        ref = self._fallback(gs)
        if ref is not None:
            return self._via_ref(ref, 'raw_search', gs, sims, add_root_noise, add_root_temp)
        e = self._ensure(gs)
        e.reset_max_depth()
        v = np.zeros(gs.num_players() + 1, dtype=np.float32)
        p = np.full(gs.action_size(), 1, dtype=np.float32)
        for _ in range(sims):
            leaf = self.find_leaf(gs)
            self.process_results(leaf, v, p, add_root_noise, add_root_temp)

    def update_root(self, gs, a):                                      # MCTS.pyx:185-195 (raises ValueError)
        ref = self._fallback(gs)
        if ref is not None:
            return self._via_ref(ref, 'update_root', gs, a)
        e = self._ensure(gs)
        self._sync_root_state(gs)
        e.update_root(0, int(a))
        self._compact_futile = False                                   # (the played move's siblings are garbage now)

    def find_leaf(self, gs):                                           # MCTS.pyx:208-228
        ref = self._fallback(gs)
        if ref is not None:
            return self._via_ref(ref, 'find_leaf', gs)
        e = self._ensure(gs)
        self._sync_root_state(gs)
        self._make_room(2 * self._max_children)                       # the store is nearly full: drop what the root no longer reaches
        e.select(None)
        st = e.get_leaf_states(0, 1, full=True)[0]
        info = e.tree_info(0)
        self._nodes_used = info['nodes_used']
        self.depth, self._max_depth = info['depth'], info['max_depth']
        return decode_state(gs, *st)

    def process_results(self, gs, value, pi, add_root_noise, add_root_temp):   # MCTS.pyx:230-289
        if self._ref is not None:
            return self._via_ref(self._ref, 'process_results', gs, value, pi, add_root_noise, add_root_temp)
        e = self._engine
        nv = e.NV
        v = np.zeros(nv, np.float32)
        vv = np.asarray(value, np.float32).reshape(-1)
        v[:min(nv, len(vv))] = vv[:nv]
        pol = torch.from_numpy(np.ascontiguousarray(np.asarray(pi, np.float32).reshape(1, -1))).to(e.device)
        val = torch.from_numpy(v.reshape(1, -1)).to(e.device)
        e.backup(pol, val, add_root_noise=bool(add_root_noise), add_root_temp=bool(add_root_temp))

    def counts(self, gs):                                              # MCTS.pyx:297-303
        ref = self._fallback(gs)
        if ref is not None:
            return ref.counts(gs)
        return self._ensure(gs).root_counts()[0].cpu().numpy()

    def best_action(self, gs):                                         # MCTS.pyx:305-306
        if self._ref is not None:
            return self._ref.best_action(gs)
        return int(np.argmax(self.counts(gs)))

    def probs(self, gs, temp=1.0):                                     # MCTS.pyx:308-329
        ref = self._fallback(gs)
        if ref is not None:
            return ref.probs(gs, temp)
        p = self._ensure(gs).root_probs(float(np.float32(temp)))[0].cpu().numpy()
        if np.isnan(p).any():                                          # no visited child: counts / 0 under np.seterr(all='raise') (:23)
            raise FloatingPointError('invalid value encountered in divide')
        return p

    def value(self, average=False):                                    # MCTS.pyx:331-344
        if self._ref is not None:
            return self._ref.value(average)
        if self._engine is None:
            return 0.0
        return float(self._engine.root_value(bool(average))[0].item())

    @property
    def _root(self):
        if self._ref is not None:
            return self._ref._root
        if self._engine is None:
            return Node(self, -1)
        i = self._engine.tree_info(0)
        return Node(self, -1, -1, i['n'], i['q'], 0.0, i['v'], i['player'], i['e'])
