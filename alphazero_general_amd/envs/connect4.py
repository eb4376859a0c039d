"""connect4 GameState (host-side plugin; API of alphazero/envs/connect4/connect4.pyx:20-99 + Connect4Logic.pyx).
Device rules for the search live in csrc/azg_games.h (struct C4); this class is the Python object callers hold."""
from typing import List, Tuple

import numpy as np

from ..Game import GameState

HEIGHT, WIDTH, WIN_LENGTH, NUM_PLAYERS, MAX_TURNS, NUM_CHANNELS = 6, 7, 4, 2, 42, 4


class Board:
    """`pieces` int32[6,7] with 1 / -1 / 0, as Connect4Logic.pyx:20-37."""

    def __init__(self):
        self.pieces = np.zeros((HEIGHT, WIDTH), dtype=np.intc)

    def add_stone(self, column, player):                      # Connect4Logic.pyx:40-47
        col = self.pieces[:, column]
        empty = np.flatnonzero(col == 0)
        if len(empty) == 0:
            raise ValueError("Can't play column %s on board %s" % (column, self))
        self.pieces[empty[-1], column] = player

    def get_valid_moves(self):                                # :49-57
        return (self.pieces[0] == 0).astype(np.intc)

    def get_win_state(self):                                  # :59-110
        p = self.pieces
        for player in (1, -1):
            m = (p == player)
            h = m[:, :-3] & m[:, 1:-2] & m[:, 2:-1] & m[:, 3:]
            v = m[:-3] & m[1:-2] & m[2:-1] & m[3:]
            d1 = m[:-3, :-3] & m[1:-2, 1:-2] & m[2:-1, 2:-1] & m[3:, 3:]
            d2 = m[:-3, 3:] & m[1:-2, 2:-1] & m[2:-1, 1:-2] & m[3:, :-3]
            if h.any() or v.any() or d1.any() or d2.any():
                return True, player
        if not (p[0] == 0).any():
            return True, 0
        return False, 0

    def __str__(self):
        return str(self.pieces)


class Game(GameState):
    AZG_GAME_ID = 0

    def __init__(self):
        super().__init__(Board())

    def __hash__(self):
        return hash(self._board.pieces.tobytes() + bytes([self.turns]) + bytes([self._player]))

    def __eq__(self, other):
        return (self._board.pieces == other._board.pieces).all() and self._player == other._player and self.turns == other.turns

    def clone(self):
        g = Game()
        g._board.pieces = np.copy(self._board.pieces)
        g._player, g._turns, g.last_action = self._player, self._turns, self.last_action
        return g

    @staticmethod
    def max_turns():
        return MAX_TURNS

    @staticmethod
    def has_draw():
        return True

    @staticmethod
    def num_players():
        return NUM_PLAYERS

    @staticmethod
    def action_size():
        return WIDTH

    @staticmethod
    def observation_size() -> Tuple[int, int, int]:
        return NUM_CHANNELS, HEIGHT, WIDTH

    def valid_moves(self):
        return np.asarray(self._board.get_valid_moves())

    def play_action(self, action: int) -> None:
        super().play_action(action)
        self._board.add_stone(action, (1, -1)[self.player])
        self._update_turn()

    def win_state(self) -> np.ndarray:
        result = [False] * 3
        over, player = self._board.get_win_state()
        if over:
            result[{1: 0, -1: 1}.get(player, -1)] = True
        return np.array(result, dtype=np.uint8)

    def observation(self):
        pieces = self._board.pieces
        return np.array([pieces == 1, pieces == -1, np.full_like(pieces, self.player),
                         np.full(pieces.shape, self.turns / MAX_TURNS, dtype=np.float32)], dtype=np.float32)

    def symmetries(self, pi) -> List[Tuple['Game', np.ndarray]]:
        m = self.clone()
        m._board.pieces = self._board.pieces[:, ::-1].copy()
        return [(self.clone(), pi), (m, pi[::-1])]

    # ---- device-engine conversion (include/azg.h azg_state) ----
    def to_azg_state(self):
        return np.asarray(self._board.pieces, np.int8).reshape(-1), self._player, self._turns

    @classmethod
    def from_azg_state(cls, cells, player, turns):
        g = cls()
        g._board.pieces = np.asarray(cells, np.intc).reshape(HEIGHT, WIDTH).copy()
        g._player, g._turns = int(player), int(turns)
        return g
