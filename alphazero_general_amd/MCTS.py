"""`MCTS` with the public surface of the reference's Cython class (alphazero/MCTS.pyx:119-344): same constructor
argument (the args dotdict), same method names / argument meaning / errors, backed by a one-slot device engine
(include/azg.h).  Used by GenericPlayers.MCTSPlayer / RawMCTSPlayer (GenericPlayers.py:100-200) and anything else
that drives one tree at a time; the batched path (SelfPlayAgent) talks to a B-slot engine directly.

Every call is a kernel launch on a single wavefront plus a host sync, so this class is functional, not fast: the
throughput path is `alphazero_general_amd.selfplay.SelfPlayRunner` / `SelfPlayAgent`.
"""
import os

import numpy as np
import torch

from . import _abi
from .engine import DeviceEngine
from .Game import azg_game_id

NOISE_ALPHA_RATIO = 10.83          # MCTS.pyx:20
_DRAW_VALUE = 0.5                  # MCTS.pyx:21


def encode_state(gs):
    """GameState object -> (cells int8, player, turns) of include/azg.h azg_state."""
    if hasattr(gs, 'to_azg_state'):
        return gs.to_azg_state()
    gid = azg_game_id(gs)
    if gid == 0:                   # the reference's own connect4 Game (envs/connect4/connect4.pyx:20-40)
        return np.asarray(gs._board.pieces, np.int8).reshape(-1), gs.player, gs.turns
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
    g._board.pieces = np.asarray(cells, np.intc).reshape(np.asarray(template._board.pieces).shape).copy()
    g._player, g._turns = int(player), int(turns)
    return g


class Node:
    """Read-only view of one tree node with the reference's attribute names (MCTS.pyx:49-57)."""

    def __init__(self, mcts, idx, a=-1, n=0, q=0.0, p=0.0, v=0.0):
        self._mcts, self._idx = mcts, idx
        self.a, self.n, self.q, self.p, self.v = a, n, q, p, v

    @property
    def _children(self):
        e = self._mcts._engine
        if e is None:
            return []
        return [Node(self._mcts, c['idx'], c['a'], c['n'], c['q'], c['p'], c['v'])
                for c in e.node_children(0, self._idx)]

    def __repr__(self):
        return 'Node(a={}, q={}, v={}, n={}, p={})'.format(self.a, self.q, self.v, self.n, self.p)


class MCTS:
    def __init__(self, args):
        self.root_noise_frac = args.root_noise_frac            # MCTS.pyx:134-139
        self.root_temp = args.root_policy_temp
        self.min_discount = args.min_discount
        self.fpu_reduction = args.fpu_reduction
        self.cpuct = args.cpuct
        self._num_players = args._num_players
        self._seed = int(args.get('_azg_seed', int.from_bytes(os.urandom(7), 'little'))) if hasattr(args, 'get') else 0
        self._sims_hint = int(args.get('numMCTSSims', 100) or 100) if hasattr(args, 'get') else 100
        self._engine = None
        self._game = None
        self._leaf_template = None
        self.depth = 0
        self.max_depth = 0

    # ---- engine plumbing ----
    def _ensure(self, gs):
        gid = azg_game_id(gs)
        if self._engine is None:
            self._game = gid
            # ONE tree: memory is no concern, so its node store holds a whole game's worth of expansions (max_turns moves x the
            # simulations per move x max_children: connect4 2.4 MB, brandubh 123 MB at 200 simulations) -- more than a search
            # that drops nothing could fill; a caller that keeps searching at ONE root (pondering, sims far above
            # args.numMCTSSims) is served by the forced compaction in find_leaf until the LIVE subtree itself exceeds the store
            # (AZG_E_TREE_FULL; the reference's limit there is host memory)
            from . import _abi
            gi = _abi.game_info(gid)
            sims = max(self._sims_hint, 200)
            cap = min(max(gi.max_turns, 16) * sims * gi.max_children + 64, (1 << 28) - 1)
            self._engine = DeviceEngine(gid, 1, cpuct=self.cpuct, fpu_reduction=self.fpu_reduction,
                                        root_noise_frac=self.root_noise_frac, root_policy_temp=self.root_temp,
                                        min_discount=self.min_discount, seed=self._seed, sims_hint=sims, nodes_per_tree=cap)
            self._max_children = gi.max_children
        elif gid != self._game:
            raise ValueError('this MCTS object was created for another game')
        return self._engine

    def _sync_root_state(self, gs):
        self._engine.set_states([encode_state(gs)], reset_trees=False)

    def reset(self):                                           # MCTS.pyx:154-160
        if self._engine is not None:
            self._engine.reset()
        self.depth = self.max_depth = 0

    def __repr__(self):
        return 'MCTS(root_noise_frac={}, root_temp={}, min_discount={}, fpu_reduction={}, cpuct={}, _num_players={}, depth={}, max_depth={})' \
            .format(self.root_noise_frac, self.root_temp, self.min_discount, self.fpu_reduction, self.cpuct,
                    self._num_players, self.depth, self.max_depth)

    # ---- public API ----
    def search(self, gs, nn, sims, add_root_noise, add_root_temp):     # MCTS.pyx:165-173
        e = self._ensure(gs)
        e.reset_max_depth()
        for _ in range(sims):
            leaf = self.find_leaf(gs)
            p, v = nn(leaf.observation())
            self.process_results(leaf, v, p, add_root_noise, add_root_temp)

    def raw_search(self, gs, sims, add_root_noise, add_root_temp):     # MCTS.pyx:175-183
        e = self._ensure(gs)
        e.reset_max_depth()
        v = np.zeros(gs.num_players() + 1, dtype=np.float32)
        p = np.full(gs.action_size(), 1, dtype=np.float32)
        for _ in range(sims):
            leaf = self.find_leaf(gs)
            self.process_results(leaf, v, p, add_root_noise, add_root_temp)

    def update_root(self, gs, a):                                      # MCTS.pyx:185-195 (raises ValueError)
        e = self._ensure(gs)
        self._sync_root_state(gs)
        e.update_root(0, int(a))

    def find_leaf(self, gs):                                           # MCTS.pyx:208-228
        e = self._ensure(gs)
        self._sync_root_state(gs)
        if getattr(self, '_nodes_used', 0) + 2 * self._max_children > e.nodes_per_tree:
            e.compact(0, force=True)                                   # the store is nearly full: drop what the root no longer reaches
        e.select(None)
        st = e.get_leaf_states(0, 1, full=True)[0]
        info = e.tree_info(0)
        self._nodes_used = info['nodes_used']
        self.depth, self.max_depth = info['depth'], info['max_depth']
        return decode_state(gs, *st)

    def process_results(self, gs, value, pi, add_root_noise, add_root_temp):   # MCTS.pyx:230-289
        e = self._engine
        nv = e.NV
        v = np.zeros(nv, np.float32)
        vv = np.asarray(value, np.float32).reshape(-1)
        v[:min(nv, len(vv))] = vv[:nv]
        pol = torch.from_numpy(np.ascontiguousarray(np.asarray(pi, np.float32).reshape(1, -1))).to(e.device)
        val = torch.from_numpy(v.reshape(1, -1)).to(e.device)
        e.backup(pol, val, add_root_noise=bool(add_root_noise), add_root_temp=bool(add_root_temp))

    def counts(self, gs):                                              # MCTS.pyx:297-303
        return self._ensure(gs).root_counts()[0].cpu().numpy()

    def best_action(self, gs):                                         # MCTS.pyx:305-306
        return int(np.argmax(self.counts(gs)))

    def probs(self, gs, temp=1.0):                                     # MCTS.pyx:308-329
        p = self._ensure(gs).root_probs(float(np.float32(temp)))[0].cpu().numpy()
        if np.isnan(p).any():                                          # no visited child: counts / 0 under np.seterr(all='raise') (:23)
            raise FloatingPointError('invalid value encountered in divide')
        return p

    def value(self, average=False):                                    # MCTS.pyx:331-344
        if self._engine is None:
            return 0.0
        return float(self._engine.root_value(bool(average))[0].item())

    @property
    def _root(self):
        if self._engine is None:
            return Node(self, -1)
        i = self._engine.tree_info(0)
        nd = Node(self, -1, -1, i['n'], i['q'], 0.0, i['v'])
        nd.player, nd.e = i['player'], np.array([(i['e'] >> j) & 1 for j in range(self._num_players)], np.uint8)
        return nd
