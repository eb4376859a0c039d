"""The single-tree `MCTS` class (the reference's public surface, MCTS.pyx:119-344) beyond test_gpu_parity.py::test_mcts_class_api_vs_oracle:
pickling with a live tree (MCTS.pyx:8 auto_pickle), `search` on the persistent launch when `nn` is this package's NNetWrapper
(MCTS.pyx:165-173 called once per move by GenericPlayers.py:133-134), and the node-store budget."""
import pickle

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _args(**kw):
    from alphazero_general_amd.utils import dotdict
    a = dotdict(root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1, fpu_reduction=0.2, cpuct=1.25, _num_players=3, numMCTSSims=50,
                _azg_seed=4242)
    a.update(kw)
    return a


def test_mcts_pickle_round_trip_continues_bit_for_bit():
    """search, pickle, unpickle (a fresh engine), and both objects keep searching: same counts, same tree, same random tape; a leaf
    found before pickling can still be backed up after it (_curnode / _path travel too)."""
    from alphazero_general_amd.MCTS import MCTS
    from alphazero_general_amd.envs.connect4 import Game
    seed = 4242
    m = MCTS(_args())
    g = Game()
    for a in (3, 2):
        g.play_action(a)
    step = [0]

    def nn(obs):
        p, v = ol.fake_eval(seed, 0, step[0], 7, 3)
        step[0] += 1
        return p, v
    m.search(g, nn, 40, True, True)
    leaf = m.find_leaf(g)                                            # a pending find_leaf crosses the pickle
    m2 = pickle.loads(pickle.dumps(m))
    assert type(m2) is MCTS and m2._engine is not m._engine
    assert (m2.counts(g) == m.counts(g)).all() and (m2.depth, m2.max_depth) == (m.depth, m.max_depth)
    assert m2._root.n == m._root.n and m2.value() == m.value()
    p, v = ol.fake_eval(seed, 0, 999, 7, 3)
    for x in (m, m2):
        x.process_results(leaf, v, p, True, True)
    for x in (m, m2):
        step[0] = 100
        x.search(g, nn, 30, True, True)
    assert (m2.counts(g) == m.counts(g)).all() and (m2.probs(g, 1.0) == m.probs(g, 1.0)).all()
    assert (m._engine.tape_counters() == m2._engine.tape_counters()).all()
    ca = [(c.a, c.n, c.q, c.p, c.v) for c in m._root._children]
    cb = [(c.a, c.n, c.q, c.p, c.v) for c in m2._root._children]
    assert ca == cb and len(ca) == 7
    a = m.best_action(g)
    m.update_root(g, a); m2.update_root(g, a)
    g.play_action(a)
    m3 = pickle.loads(pickle.dumps(m2))                              # after a re-root (and its compaction)
    for x in (m, m3):
        step[0] = 200
        x.search(g, nn, 30, False, False)
    assert (m3.counts(g) == m.counts(g)).all()


@pytest.mark.parametrize('game', ['connect4', 'brandubh', 'trimok', 'connect4:32', 'connect4:64'])
def test_mcts_search_on_the_persistent_launch(game):
    """MCTS.search(gs, nn, sims, noise, temp) with nn = this package's NNetWrapper: ONE launch per call, the same tree as the
    find_leaf / nn(obs) / process_results loop bit for bit -- connect4's fused heads and the wide-head networks' exact launch
    (azg_search_wide_exact_f16: all A + P+1 logits inside the launch) alike --, noise and temperature flags honoured per call."""
    import importlib
    import torch
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.MCTS import MCTS
    name, _, width = game.partition(':')                              # 'connect4:32': the reference's default net (Coach.py:108-116) on connect4
    Game = importlib.import_module('alphazero_general_amd.envs.' + name).Game
    na = {'connect4': N.CONNECT4_NET_ARGS, 'brandubh': N.BRANDUBH_NET_ARGS, 'trimok': N.DEFAULT_NET_ARGS}[name]
    if width:
        na = N.dotdict(dict(N.DEFAULT_NET_ARGS, num_channels=int(width)))
    torch.manual_seed(3)
    net = N.NNetWrapper(Game, na, device='cuda:0', dtype=torch.float16)
    args = _args(_num_players=Game.num_players() + 1, numMCTSSims=64)
    fast, slow = MCTS(args), MCTS(args)
    calls = [0]

    def plain(obs):                                                  # not an NNetWrapper: the per-simulation loop
        calls[0] += 1
        return net.predict(obs)
    g = Game()
    for mv in range(4):
        noise = temp = (mv % 2 == 0)
        fast.search(g, net, 64, noise, temp)
        n0 = calls[0]
        slow.search(g, plain, 64, noise, temp)
        assert calls[0] - n0 == 64
        cf, cs = fast.counts(g), slow.counts(g)
        assert cf.sum() == cs.sum() == fast._root.n - 1
        assert (cf == cs).all(), (mv, cf, cs)
        assert (fast.probs(g, 1.0) == slow.probs(g, 1.0)).all() and fast.value() == slow.value()
        assert (fast.depth, fast.max_depth) == (slow.depth, slow.max_depth)
        assert (fast._engine.tape_counters() == slow._engine.tape_counters()).all()
        a = slow.best_action(g)
        fast.update_root(g, a); slow.update_root(g, a)
        g.play_action(a)
    assert fast._persistent_net(net, fast._engine) is net._hip and fast._persistent_net(plain, fast._engine) is None


def test_mcts_search_splits_the_launch_when_the_store_is_small():
    """a persistent launch cannot compact in the middle: MCTS.search hands it as many simulations as are sure to fit the node store and
    reclaims the dropped siblings between launches.  An object whose store holds less than one call's worth of expansions (brandubh:
    96 nodes per expansion at most) must grow the same tree, move after move, as one with the default store -- and the engine's
    default root flags are what they were before the call."""
    import torch
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.MCTS import MCTS
    from alphazero_general_amd.envs.brandubh import Game
    torch.manual_seed(5)
    net = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS, device='cuda:0', dtype=torch.float16)
    small, big = MCTS(_args(numMCTSSims=120, _azg_nodes_per_tree=96 * 90)), MCTS(_args(numMCTSSims=120))   # (a call may add 120 x 96 nodes; it adds ~5 000)
    g = Game()
    launches = [0]
    hip = (net.refresh() or True) and net._hip
    orig = hip.search

    def counting(e, sims, exact=False):
        launches[0] += e is small._engine
        return orig(e, sims, exact=exact)
    hip.search = counting
    try:
        for mv in range(5):
            small.search(g, net, 120, mv == 0, mv == 0)
            big.search(g, net, 120, mv == 0, mv == 0)
            assert (small.counts(g) == big.counts(g)).all(), mv
            assert (small.probs(g, 1.0) == big.probs(g, 1.0)).all()
            assert (small._engine.tape_counters() == big._engine.tape_counters()).all()
            a = big.best_action(g)
            small.update_root(g, a); big.update_root(g, a)
            g.play_action(a)
    finally:
        hip.search = orig
    assert launches[0] > 5                                       # (the small store needed more than one launch per call)


def test_mcts_node_store_budget():
    """the default node store of one MCTS object stays under NODE_STORE_BUDGET whatever numMCTSSims promises (brandubh at 1600
    simulations would take ~1 GB); args._azg_nodes_per_tree overrides; the engine reports the capacity in effect."""
    from alphazero_general_amd import MCTS as M
    from alphazero_general_amd.envs.brandubh import Game
    m = M.MCTS(_args(numMCTSSims=1600))
    e = m._ensure(Game())
    assert e.nodes_per_tree * 64 <= M.NODE_STORE_BUDGET and e.nodes_per_tree >= 1600 * 96
    m2 = M.MCTS(_args(numMCTSSims=1600, _azg_nodes_per_tree=200000))
    assert m2._ensure(Game()).nodes_per_tree == 200000
    from alphazero_general_amd.engine import DeviceEngine
    d = DeviceEngine(0, 4, sims_hint=100)
    assert d.nodes_per_tree == 16 * 100 * 7 + 64 and d.compact_reserve == 100 * 7                 # the library's defaults, read back


def test_slot_snapshot_is_checked_on_import():
    """azg_slot_import refuses a snapshot of another game, a truncated one, and one that holds more nodes than the receiving engine's
    node store (AZG_E_TREE_FULL) -- loudly, never a partial restore that would search a broken tree."""
    import torch
    from alphazero_general_amd import _abi
    from alphazero_general_amd.engine import DeviceEngine
    a = DeviceEngine(0, 2, seed=3, sims_hint=30)
    pol = torch.full((2, 7), 1 / 7, device=a.device); val = torch.full((2, 3), 1 / 3, device=a.device)
    for _ in range(30):
        a.select(None); a.backup(pol, val)
    blob = a.export_slot(1)
    used = a.tree_info(1)['nodes_used']
    assert used > 100 and len(blob) > used * 32
    b = DeviceEngine(0, 1, seed=9, sims_hint=30)
    b.import_slot(blob, 0)                                        # slot 1 of one engine -> slot 0 of another
    assert b.tree_info(0) == a.tree_info(1) and (b.root_counts()[0] == a.root_counts()[1]).all()
    for x in (a, b):
        x.select(None); x.backup(pol, val)
    assert (b.root_counts()[0] == a.root_counts()[1]).all() and b.tape_counters()[0] == a.tape_counters()[1]
    small = DeviceEngine(0, 1, seed=9, sims_hint=30, nodes_per_tree=64)
    with pytest.raises(_abi.AzgError) as ei:
        small.import_slot(blob, 0)
    assert ei.value.code == _abi.E_TREE_FULL
    other = DeviceEngine(2, 1, seed=9, sims_hint=30)
    with pytest.raises(_abi.AzgError):
        other.import_slot(blob, 0)
    with pytest.raises(_abi.AzgError):
        b.import_slot(blob[:200], 0)
    # a snapshot whose indices point outside its own nodes (corrupted, hand-made, or written by a library with another record layout)
    # must be refused before anything is written -- not imported and searched into an out-of-bounds device access
    import struct
    head = 8 + 4 * 4 + 80 * 2 + 8                                  # SnapHead: magic, game / T / maxd / layout, root and leaf state, tape counter
    bad = bytearray(blob)
    struct.pack_into('<i', bad, head + 32 + 8, 10 ** 6)             # TreeHdr.depth far beyond max_turns + 2
    with pytest.raises(_abi.AzgError):
        b.import_slot(bytes(bad), 0)
    bad = bytearray(blob)
    struct.pack_into('<i', bad, head + 16, used + 5)                # the root's first_child + nchild past the live nodes
    with pytest.raises(_abi.AzgError):
        b.import_slot(bytes(bad), 0)
    bad = bytearray(blob)
    node0 = head + 64 + 16 * 44                                     # TreeHdr, then PathEnt[max_turns + 2 = 44], then the nodes
    struct.pack_into('<i', bad, node0 + 32 * 3 + 16, used)          # node 3's child block starts past the end
    struct.pack_into('<H', bad, node0 + 32 * 3 + 22, 7)
    with pytest.raises(_abi.AzgError):
        b.import_slot(bytes(bad), 0)
    bad = bytearray(blob)
    struct.pack_into('<i', bad, 8 + 12, 0)                          # layout word of another library version
    with pytest.raises(_abi.AzgError):
        b.import_slot(bytes(bad), 0)
    b.import_slot(blob, 0)                                          # (the refusals left the engine usable)
    assert b.tree_info(0)['nodes_used'] == used


def test_max_depth_is_writable_like_the_reference():
    """Evaluator.py:343 drives a search of its own -- `mcts.max_depth = 0`, then find_leaf / nn / process_results per simulation (:351-353) --
    and reads mcts.max_depth afterwards (:357,396): the attribute is public and writable in the reference (MCTS.pyx:130)."""
    from alphazero_general_amd.MCTS import MCTS
    from alphazero_general_amd.envs.connect4 import Game
    m, g = MCTS(_args()), Game()
    step = [0]

    def run(n):
        for _ in range(n):
            leaf = m.find_leaf(g)
            p, v = ol.fake_eval(7, 0, step[0], 7, 3); step[0] += 1
            m.process_results(leaf, v, p, False, False)
    run(60)
    deep = m.max_depth
    assert deep >= 2
    a = m.best_action(g)
    m.update_root(g, a); g.play_action(a)
    m.max_depth = 0                                               # Evaluator.run resets it before its loop
    assert m.max_depth == 0
    run(3)
    assert 0 < m.max_depth <= deep and m.max_depth == m._engine.tree_info(0)['max_depth']


def test_node_view_has_e_and_player_like_the_reference():
    """Node.e / Node.player (MCTS.pyx:52,57; SURVEY.md 8b lists `_root` with `_children, a, q, n, v, p, e, player`): zeros on a node that has
    not been expanded (Node.__init__ :59-67), the win state of the position and the player to move there once it has (find_leaf :223-226)."""
    import numpy as np
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.MCTS import MCTS
    args = _args(_num_players=3, numMCTSSims=60)
    m = MCTS(args)
    g = Game()
    for a in (3, 0, 3, 0, 3):                                        # player 0 holds three in column 3: wins two plies on unless blocked
        g.play_action(a)
    m.raw_search(g, 300, False, False)
    root = m._root
    assert root.player == g.player and root.e.dtype == np.uint8 and root.e.shape == (3,) and not root.e.any()
    seen_terminal = False

    def walk(node, state, depth):
        nonlocal seen_terminal
        for c in node._children:
            assert c.e.dtype == np.uint8 and c.e.shape == (3,)
            if c.n == 0:
                assert c.player == 0 and not c.e.any()              # never reached: as constructed
                continue
            s = state.clone(); s.play_action(c.a)
            assert c.player == s.player, (depth, c.a)
            assert (c.e == np.asarray(s.win_state(), np.uint8)).all(), (depth, c.a)
            seen_terminal |= bool(c.e.any())
            if depth < 3 and not c.e.any():
                walk(c, s, depth + 1)
    walk(root, g, 0)
    assert seen_terminal
