"""Host-side GameState plugins (alphazero_general_amd/envs) against the oracle rules, which are pinned to the reference by
the golden vectors: same boards, valid moves, win states, observations and symmetries on random playouts."""
import numpy as np

import oracle_lib as ol


def _playouts(game_cls, gid, n, seed, sym_every=0):
    rng = np.random.RandomState(seed)
    for it in range(n):
        g, o = game_cls(), ol.OGame(gid)
        t = 0
        while True:
            st = g.to_azg_state()
            assert (o.cells() == st[0]).all() and o.player == st[1] == g.player and o.turns == st[2] == g.turns
            if len(st) > 3:
                assert o.s.aux[0] == st[3]
            v = g.valid_moves()
            assert (o.valid_moves() == v).all()
            assert (o.win_state() == g.win_state()).all()
            assert (o.observation() == g.observation()).all()
            if sym_every and t % sym_every == 1:
                pi = (rng.rand(len(v)).astype(np.float32) * v)
                for k, (gs, pk) in enumerate(g.symmetries(pi)):
                    os_, opk = o.symmetry(pi, k)
                    assert (os_.cells() == gs.to_azg_state()[0]).all() and (opk == pk).all()
            if g.win_state().any():
                break
            a = int(rng.choice(np.flatnonzero(v)))
            g.play_action(a); o.play(a); t += 1
        g2 = game_cls.from_azg_state(*g.to_azg_state())
        assert g2 == g and (g2.valid_moves() == g.valid_moves()).all()


def test_connect4_host_env():
    from alphazero_general_amd.envs.connect4 import Game
    _playouts(Game, ol.GAME_CONNECT4, 150, 1, sym_every=5)


def test_brandubh_host_env():
    from alphazero_general_amd.envs.brandubh import Game
    _playouts(Game, ol.GAME_BRANDUBH, 25, 2, sym_every=23)


def test_trimok_host_env():
    from alphazero_general_amd.envs.trimok import Game
    _playouts(Game, ol.GAME_TRIMOK, 150, 3, sym_every=4)
