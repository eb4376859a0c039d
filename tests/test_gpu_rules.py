"""Device rule kernels (csrc/azg_games.h: struct C4, struct BR) against the rule tables the REFERENCE itself produced
(tests/golden/c4_rules.npz: 10 362 random-playout positions of alphazero/envs/connect4 + the data of the reference's own
envs/connect4/test_connect4.py:31-39,58-64,99-151; tests/golden/br_rules.npz: 10 773 positions of fastafl/cengine.pyx:109-272
through envs/brandubh/fastafl.pyx) -- every position, on the GPU, through the C ABI:

  * expanding a root exposes valid_moves (the child action set), win_state (Node.e) and observation (the leaf row);
  * a second simulation steered onto the playout's next move exposes play_action: the leaf state must be the table's next
    position (captures, surrounds, king flags, draw-by-turns), and so must its observation.

Connect4Logic.pyx:40-110, connect4.pyx:54-91; fastafl/cengine.pyx:109-272, envs/brandubh/fastafl.pyx:48-121,196-211."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
C4, BR = 0, 1


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def _engine(game, B):
    from alphazero_general_amd.engine import DeviceEngine
    return DeviceEngine(game, B, seed=3, sims_hint=4, cpuct=1.25, fpu_reduction=0.2)


def _ebits(ws):
    return int(ws[0]) + 2 * int(ws[1]) + 4 * int(ws[2])


def _check_table(torch, game, cells, lens, kc, valids, ws, obs_crc, moves):
    """cells [n, CELLS] int8, lens [n] plies played, valids [n, A] 0/1, ws [n, 3], obs_crc [n], moves [n, maxlen] (the playout
    prefix of every position; positions of one playout are consecutive, so row i + 1 is row i + one move when lens grows by 1)"""
    n, A = len(lens), valids.shape[1]
    eng = _engine(game, n)
    states = []
    for i in range(n):
        L = int(lens[i])
        states.append((cells[i], L % 2, L) + ((int(kc[i]),) if kc is not None else ()))
    eng.set_states(states)
    obs = eng.new_obs()
    eng.select(obs)                                           # find_leaf at a fresh root: win_state, valid_moves, add_children, observation
    o = obs.cpu().numpy()
    bad = [i for i in range(n) if crc(o[i]) != obs_crc[i]]
    assert not bad, ('observation', bad[:5])
    for i in range(n):
        ch = eng.root_children(i)
        assert (np.sort(ch['a']) == np.flatnonzero(valids[i])).all(), ('valid_moves', i)
        assert eng.tree_info(i)['e'] == _ebits(ws[i]), ('win_state', i)
    # one backup with a policy peaked on the playout's next move (PUCT at root.n == 1 with no visited child picks the largest
    # prior), then the second simulation descends exactly that ply
    has_next = np.zeros(n, bool)
    nxt = np.zeros(n, np.int64)
    for i in range(n - 1):
        if lens[i + 1] == lens[i] + 1 and not ws[i].any():
            assert (moves[i + 1][:lens[i]] == moves[i][:lens[i]]).all()
            has_next[i] = True; nxt[i] = int(moves[i + 1][lens[i]])
    pol = np.full((n, A), 1e-4, np.float32)
    pol[np.arange(n), nxt] = 0.9
    val = np.full((n, 3), 1.0 / 3, np.float32)
    eng.backup(torch.from_numpy(pol).to(eng.device), torch.from_numpy(val).to(eng.device))
    eng.select(obs)
    o = obs.cpu().numpy()
    leaves = eng.get_leaf_states(full=True)
    for i in range(n):
        lc, lp, lt, lk = leaves[i]
        if ws[i].any():                                       # terminal root: find_leaf stops at it (MCTS.pyx:213)
            assert len(eng.last_path(i)) == 0 and (lc == cells[i]).all(), ('terminal', i)
            continue
        if not has_next[i]:
            continue
        assert list(eng.last_path(i)) == [nxt[i]], ('descent', i)
        assert (lc == cells[i + 1]).all(), ('play_action: board', i)
        assert lp == (lens[i] + 1) % 2 and lt == lens[i] + 1, ('play_action: player / turns', i)
        if kc is not None:
            assert lk == kc[i + 1], ('play_action: king flag', i)
        assert crc(o[i]) == obs_crc[i + 1], ('observation after play_action', i)
    eng.counters()                                            # no sticky device error
    eng.close()
    return int(has_next.sum())


def test_c4_rules_vs_reference_tables():
    import torch
    d = dict(np.load(os.path.join(G, 'c4_rules.npz')))
    n = _check_table(torch, C4, d['cells'], d['lens'], None, d['valids'], d['ws'], d['obs_crc'], d['moves'])
    assert len(d['lens']) >= 10000 and n > 9000


def test_c4_reference_test_data_on_device():
    """the reference's own test tables (envs/connect4/test_connect4.py:31-39 move list -> board, :58-64 valid-move table, :99-151
    ten end-state boards) through the device rules."""
    import torch
    d = dict(np.load(os.path.join(G, 'c4_rules.npz')))
    boards = d['end_boards']
    eng = _engine(C4, len(boards))
    eng.set_states([(b.reshape(-1), int(np.count_nonzero(b)) % 2, int(np.count_nonzero(b))) for b in boards])
    eng.select(None)
    for i, (b, ws, winner) in enumerate(zip(boards, d['end_ws'], d['end_winner'])):
        e = eng.tree_info(i)['e']
        assert e == _ebits(ws), i
        assert bool(e & 1) == (winner == 1) and bool(e & 2) == (winner == -1), i
        assert (np.sort(eng.root_children(i)['a']) == np.flatnonzero(b[0] == 0)).all(), i     # columns whose top cell is free
    eng.close()
    # :31-39 the move list [4, 5, 4, 3, 0, 6] played on the device (every prefix set as a root, the tree steered onto the next move, its
    # successor read back as the leaf) must give the reference's board
    eng = _engine(C4, 1)
    cells = np.zeros(42, np.int8)
    for t, a in enumerate([4, 5, 4, 3, 0, 6]):
        eng.set_states([(cells, t % 2, t)])
        eng.select(None)
        pol = np.full((1, 7), 1e-4, np.float32); pol[0, a] = 0.9
        eng.backup(torch.from_numpy(pol).to(eng.device), torch.full((1, 3), 1 / 3, device=eng.device))
        eng.select(None)
        assert list(eng.last_path(0)) == [a]
        cells = eng.get_leaf_states()[0][0].copy()
    assert (cells.reshape(6, 7) == d['moves_board']).all()
    eng.close()
    # :58-64 valid-move table.  The reference's test keeps dropping stones after a four-in-a-row (its Board does not stop), which a
    # search never does, so these boards are built on the host (a stone falls to the lowest free cell of its column, +1 / -1
    # alternating) and the device answers valid_moves for them
    lists = [[int(x) for x in mv[mv >= 0]] for mv in d['vm_moves']]
    boards = []
    for mv in lists:
        b = np.zeros((6, 7), np.int8)
        for t, a in enumerate(mv):
            b[np.flatnonzero(b[:, a] == 0).max(), a] = 1 if t % 2 == 0 else -1
        boards.append(b)
    eng = _engine(C4, len(boards))
    eng.set_states([(b.reshape(-1), len(mv) % 2, len(mv)) for b, mv in zip(boards, lists)])
    eng.select(None)
    for i, ev in enumerate(d['vm_expected']):
        v = np.zeros(7, np.uint8); v[eng.root_children(i)['a']] = 1
        assert (v == ev).all(), i
    eng.close()


def test_br_rules_vs_reference_tables():
    import torch
    d = dict(np.load(os.path.join(G, 'br_rules.npz')))
    valids = np.unpackbits(d['valid_bits'], axis=1)[:, :588]
    n = _check_table(torch, BR, d['cells'], d['lens'], d['kc'], valids, d['ws'], d['obs_crc'], d['moves'])
    assert len(d['lens']) >= 10000 and n > 9000
    assert int(valids.sum(1).max()) == int(d['max_k'])
