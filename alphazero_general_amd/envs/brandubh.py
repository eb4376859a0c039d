"""brandubh (7x7 tafl) GameState -- host-side plugin with the API of alphazero/envs/brandubh/fastafl.pyx:123-268 (which at
the reference snapshot lacks has_draw()/max_turns(), SURVEY.md Q19: both are provided here).  Rules follow
fastafl/cengine.pyx with variants.brandubh_args (two-sided king capture, moves over the empty throne, king cannot
re-enter the throne).  Device rule kernels for the search: csrc/azg_games.h struct BR; this class is the Python
object callers hold (Arena.play_game, GenericPlayers, GUI-style code)."""
from typing import List, Tuple

import numpy as np

from ..Game import GameState

W = H = 7
NUM_PLAYERS, NUM_CHANNELS, DRAW_MOVE_COUNT = 2, 5, 100
ACTION_SIZE = W * H * (W + H - 2)
ATT, DEF, KING, THRONE, ESCAPE, KING_THRONE, KING_ESCAPE = 1, 2, 3, 4, 5, 7, 8
_KINGS = (KING, KING_THRONE, KING_ESCAPE)
_ATTACKERS = (ATT,) + _KINGS
_DIRS = ((0, 1), (1, 0), (0, -1), (-1, 0))
_START = np.array([[int(c) for c in row] for row in
                   ('5002005', '0002000', '0001000', '2217122', '0001000', '0002000', '5002005')], dtype=np.int8)


def get_move(action):
    """action -> ((x, y), (new_x, new_y))   (fastafl.pyx:48-63)"""
    mt, a = action % 12, action // 12
    x, y = a % W, a // W
    if mt < H - 1:
        return (x, y), (x, mt + (1 if mt >= y else 0))
    nx = mt - H + 1
    return (x, y), (nx + (1 if nx >= x else 0), y)


def get_action(src, dst):
    """fastafl.pyx:66-79"""
    (x, y), (nx, ny) = src, dst
    mt = (ny if ny < y else ny - 1) if x == nx else (H + nx - 1 - (1 if nx >= x else 0))
    return 12 * (x + y * W) + mt


class Board:
    def __init__(self):
        self._state = _START.copy()
        self.num_turns = 0
        self._king_captured = False

    def copy(self):
        b = Board.__new__(Board)
        b._state, b.num_turns, b._king_captured = self._state.copy(), self.num_turns, self._king_captured
        return b

    def to_play(self):
        return 2 - self.num_turns % 2

    def _ok(self, x, y):
        return 0 <= x < W and 0 <= y < H

    def _valid(self, x, y, king):
        if not self._ok(x, y):
            return False
        v = self._state[y, x]
        return v == 0 or (v == ESCAPE and king)

    def legal_moves(self, team):
        s, out = self._state, []
        for y in range(H):
            for x in range(W):
                v = s[y, x]
                if not (v in _ATTACKERS if team == ATT else v == DEF):
                    continue
                king = v in _KINGS
                for dx, dy in _DIRS:
                    cx, cy = x + dx, y + dy
                    thr = self._ok(cx, cy) and s[cy, cx] == THRONE
                    while thr or self._valid(cx, cy, king):
                        if not thr:
                            out.append(((x, y), (cx, cy)))
                        cx, cy = cx + dx, cy + dy
                        thr = self._ok(cx, cy) and s[cy, cx] == THRONE
        return out

    def _group_blocked(self, start, enemy):
        s, group, stack = self._state, {start}, [start]
        while stack:
            x, y = stack.pop()
            for dx, dy in _DIRS:
                nx, ny = x + dx, y + dy
                if not self._ok(nx, ny):
                    continue
                v = s[ny, nx]
                if v == 0:
                    return None
                if v in enemy and (nx, ny) not in group:
                    group.add((nx, ny)); stack.append((nx, ny))
        return group

    def move(self, src, dst):
        s = self._state
        (x, y), (nx, ny) = src, dst
        v = s[y, x]
        piece = KING if v in (KING_THRONE, KING_ESCAPE) else v
        s[y, x] = THRONE if v == KING_THRONE else ESCAPE if v == KING_ESCAPE else 0
        d = s[ny, nx]
        s[ny, nx] = piece + d if d in (THRONE, ESCAPE) else piece
        pv = s[ny, nx]
        friendly = _ATTACKERS if pv in _ATTACKERS else (pv,)
        enemy = 3 - pv if pv != KING else DEF
        for dx, dy in _DIRS:                                  # custodian capture (cengine.pyx:172-197)
            ex, ey = nx + dx, ny + dy
            if not self._ok(ex, ey):
                continue
            ev = s[ey, ex]
            kingcap = ev == KING
            if ev == enemy or kingcap:
                fx, fy = ex + dx, ey + dy
                if self._ok(fx, fy) and (s[fy, fx] in friendly or s[fy, fx] in (THRONE, ESCAPE)):
                    if kingcap:
                        self._king_captured = True
                    else:
                        s[ey, ex] = 0
        enemy_set = _ATTACKERS if s[ny, nx] == DEF else (DEF,)   # group surround (cengine.pyx:204-247)
        seen = set()
        for dx, dy in _DIRS:
            ex, ey = nx + dx, ny + dy
            if not self._ok(ex, ey) or s[ey, ex] not in enemy_set or (ex, ey) in seen:
                continue
            grp = self._group_blocked((ex, ey), enemy_set)
            if grp is None:
                continue
            seen |= grp
            for gx, gy in grp:
                if s[gy, gx] in _KINGS:
                    self._king_captured = True
                else:
                    s[gy, gx] = 0
        self.num_turns += 1

    def _has_legals(self, team):
        s = self._state
        for y in range(H):
            for x in range(W):
                v = s[y, x]
                if v in _ATTACKERS if team == ATT else v == DEF:
                    if any(self._valid(x + dx, y + dy, v in _KINGS) for dx, dy in _DIRS):
                        return True
        return False

    def get_winner(self):
        if (self._state == KING_ESCAPE).any() or not self._has_legals(DEF):
            return ATT
        if self._king_captured or not self._has_legals(ATT):
            return DEF
        return 0

    def __str__(self):
        return '\n'.join(' '.join(str(v) for v in row) for row in self._state)


class Game(GameState):
    AZG_GAME_ID = 1

    def __init__(self, _board=None):
        super().__init__(_board or Board())
        self.last_action = -1

    def __eq__(self, other):
        return (self._board._state == other._board._state).all() and self._player == other._player and self._turns == other._turns

    def clone(self):
        g = Game(self._board.copy())
        g._player, g._turns, g.last_action = self._player, self._turns, self.last_action
        return g

    @staticmethod
    def num_players():
        return NUM_PLAYERS

    @staticmethod
    def action_size():
        return ACTION_SIZE

    @staticmethod
    def observation_size() -> Tuple[int, int, int]:
        return NUM_CHANNELS, W, H

    @staticmethod
    def has_draw():
        return True

    @staticmethod
    def max_turns():
        return DRAW_MOVE_COUNT

    def valid_moves(self):
        v = np.zeros(ACTION_SIZE, dtype=np.uint8)
        for src, dst in self._board.legal_moves(self._board.to_play()):
            v[get_action(src, dst)] = 1
        return v

    def play_action(self, action: int) -> None:
        self.last_action = action
        src, dst = get_move(int(action))
        self._board.move(src, dst)
        self._update_turn()

    def win_state(self) -> np.ndarray:
        r = np.zeros(NUM_PLAYERS + 1, dtype=np.uint8)
        if self.turns >= DRAW_MOVE_COUNT:
            r[NUM_PLAYERS] = 1
        else:
            w = self._board.get_winner()
            if w:
                r[2 - w] = 1
        return r

    def observation(self):
        s = self._board._state
        return np.array([s == DEF, s == ATT, np.isin(s, _KINGS), np.full(s.shape, self._board.num_turns % 2),
                         np.full(s.shape, self._board.num_turns // DRAW_MOVE_COUNT)], dtype=np.float32)

    def symmetries(self, pi) -> List[Tuple['Game', np.ndarray]]:
        syms = []
        for i in range(1, 5):
            for flip in (False, True):
                st = np.rot90(self._board._state, i)
                if flip:
                    st = np.fliplr(st)
                new_pi = np.zeros(ACTION_SIZE, dtype=np.float32)
                for a in range(ACTION_SIZE):
                    (x, y), (nx, ny) = get_move(a)
                    for _ in range(i):
                        x, nx, y, ny = W - 1 - y, W - 1 - ny, x, nx
                    if flip:
                        x, nx = W - 1 - x, W - 1 - nx
                    new_pi[get_action((x, y), (nx, ny))] = pi[a]
                g = self.clone()
                g._board._state = np.ascontiguousarray(st)
                syms.append((g, new_pi))
        return syms

    def to_azg_state(self):
        return self._board._state.reshape(-1).astype(np.int8), self._player, self._turns, int(self._board._king_captured)

    @classmethod
    def from_azg_state(cls, cells, player, turns, king_captured=0):
        g = cls()
        g._board._state = np.asarray(cells, np.int8).reshape(H, W).copy()
        g._board.num_turns = int(turns)
        g._board._king_captured = bool(king_captured)
        g._player, g._turns = int(player), int(turns)
        return g
