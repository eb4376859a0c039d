"""GameState plugin API -- same names, argument meaning and error behaviour as alphazero/Game.py:7-113 of the
reference, so env plugins and callers (GenericPlayers, Arena, Coach) are interchangeable.  A game that the device
engine can search additionally exposes `AZG_GAME_ID` (its rule kernels are registered in csrc/azg_games.h) and
`to_azg_state()` / `from_azg_state()`; any other GameState raises NotImplementedError in the engine -- there is no
CPU search fallback."""
from abc import ABC, abstractmethod
from typing import List, Optional, Tuple

import numpy as np


class GameState(ABC):
    AZG_GAME_ID = None

    def __init__(self, board):
        self._board = board
        self._player = 0
        self._turns = 0
        self.last_action = None

    def __str__(self) -> str:
        return f'Player:\t{self._player}\n{self._board}\n'

    @abstractmethod
    def __eq__(self, other) -> bool:
        pass

    @abstractmethod
    def clone(self) -> 'GameState':
        pass

    @staticmethod
    @abstractmethod
    def action_size() -> int:
        pass

    @staticmethod
    @abstractmethod
    def observation_size() -> Tuple[int, int, int]:
        pass

    @abstractmethod
    def valid_moves(self) -> np.ndarray:
        pass

    @staticmethod
    @abstractmethod
    def num_players() -> int:
        pass

    @staticmethod
    def max_turns() -> Optional[int]:
        return None

    @staticmethod
    def has_draw() -> bool:
        return True

    @property
    def player(self) -> int:
        return self._player

    @property
    def turns(self) -> int:
        return self._turns

    def _next_player(self, player, turns=1) -> int:
        return (player + turns) % self.num_players()

    def _update_turn(self) -> None:
        self._player = self._next_player(self._player)
        self._turns += 1

    @abstractmethod
    def play_action(self, action: int) -> None:
        self.last_action = action

    @abstractmethod
    def win_state(self) -> np.ndarray:
        pass

    @abstractmethod
    def observation(self) -> np.ndarray:
        pass

    def symmetries(self, pi) -> List[Tuple['GameState', np.ndarray]]:
        raise NotImplementedError(
            'Symmetries not implemented for this environment. Set symmetricSamples to False in args.')


def azg_game_id(game_cls_or_state):
    """Device game id of a GameState class/instance.  The reference's own env classes are recognised by module name so
    that alphazero.envs.connect4.connect4.Game objects can be searched unchanged."""
    gid = getattr(game_cls_or_state, 'AZG_GAME_ID', None)
    if gid is not None:
        return gid
    cls = game_cls_or_state if isinstance(game_cls_or_state, type) else type(game_cls_or_state)
    mod = getattr(cls, '__module__', '')
    if mod.endswith('envs.connect4.connect4'):
        return 0
    if mod.endswith('envs.brandubh.fastafl'):
        return 1
    raise NotImplementedError('%s.%s has no device rule kernels registered (csrc/azg_games.h); the MI355X engine has '
                              'no CPU search fallback' % (mod, cls.__name__))
