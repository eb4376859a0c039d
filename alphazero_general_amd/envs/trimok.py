"""trimok -- the build-defined 3-player GameState used for the N-player path (BASELINE config 5).  The reference ships no
env with more than two players (every env has NUM_PLAYERS = 2), while its MCTS and SelfPlayAgent are generic in the
player count (alphazero/MCTS.pyx:291-295, alphazero/Game.py:73-79).  Rules: 5x5 board, players 0,1,2 place one stone
per turn on any empty cell (action = y*5+x); three own stones in a row (horizontal, vertical, diagonal) win; a full board
is a draw.  Device rule kernels: csrc/azg_games.h struct TM; oracle: oracle/azg_trimok_ref.c."""
from typing import List, Tuple

import numpy as np

from ..Game import GameState

N, NUM_PLAYERS, MAX_TURNS, NUM_CHANNELS = 5, 3, 25, 5
_LINES = [[(x + t * dx, y + t * dy) for t in range(3)] for y in range(N) for x in range(N)
          for dx, dy in ((1, 0), (0, 1), (1, 1), (1, -1))]
_LINES = [l for l in _LINES if all(0 <= x < N and 0 <= y < N for x, y in l)]


class Game(GameState):
    AZG_GAME_ID = 2

    def __init__(self):
        super().__init__(np.zeros((N, N), dtype=np.int8))

    def __eq__(self, other):
        return (self._board == other._board).all() and self._player == other._player and self._turns == other._turns

    def clone(self):
        g = Game()
        g._board = self._board.copy()
        g._player, g._turns, g.last_action = self._player, self._turns, self.last_action
        return g

    @staticmethod
    def action_size():
        return N * N

    @staticmethod
    def observation_size() -> Tuple[int, int, int]:
        return NUM_CHANNELS, N, N

    @staticmethod
    def num_players():
        return NUM_PLAYERS

    @staticmethod
    def max_turns():
        return MAX_TURNS

    @staticmethod
    def has_draw():
        return True

    def valid_moves(self):
        return (self._board.reshape(-1) == 0).astype(np.uint8)

    def play_action(self, action: int) -> None:
        super().play_action(action)
        y, x = divmod(int(action), N)
        if self._board[y, x] != 0:
            raise ValueError("Can't play cell %d" % action)
        self._board[y, x] = self._player + 1
        self._update_turn()

    def win_state(self) -> np.ndarray:
        ws = np.zeros(NUM_PLAYERS + 1, dtype=np.uint8)
        for p in range(NUM_PLAYERS):
            if any(all(self._board[y, x] == p + 1 for x, y in l) for l in _LINES):
                ws[p] = 1
                return ws
        if not (self._board == 0).any():
            ws[NUM_PLAYERS] = 1
        return ws

    def observation(self):
        b = self._board
        return np.array([b == 1, b == 2, b == 3, np.full(b.shape, self._player), np.full(b.shape, self._turns / MAX_TURNS, dtype=np.float32)],
                        dtype=np.float32)

    def symmetries(self, pi) -> List[Tuple['Game', np.ndarray]]:
        return [(self.clone(), pi)]

    def to_azg_state(self):
        return self._board.reshape(-1).astype(np.int8), self._player, self._turns

    @classmethod
    def from_azg_state(cls, cells, player, turns):
        g = cls()
        g._board = np.asarray(cells, np.int8).reshape(N, N).copy()
        g._player, g._turns = int(player), int(turns)
        return g
