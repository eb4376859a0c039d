"""The game plugin protocol of the engine.

A game is a class whose instances are positions.  Callers written for the reference (alphazero/Game.py:7-113 --
Coach, Arena, GenericPlayers, the env modules) rely on exactly these members, so the names and meanings are kept:

  class level    action_size() observation_size() num_players() max_turns() has_draw()
  instance       clone() valid_moves() play_action(a) win_state() observation() symmetries(pi)
                 .player .turns .last_action  ( _player / _turns / _board are the storage the envs use )

What this build adds is the device side: a game the MI355X engine can search also carries
  AZG_GAME_ID                          index of its rule kernels in csrc/azg_games.h
  to_azg_state() / from_azg_state()    conversion to / from include/azg.h azg_state
A GameState without them is rejected by `azg_game_id` with NotImplementedError -- the engine has no CPU search path; behind
`install()` the MCTS / SelfPlayAgent classes hand such a game to the reference's own classes instead (reference side,
alphazero_general_amd.reference_class).
"""
import numpy as np

_CLASS_API = ('action_size', 'observation_size', 'num_players')
_INSTANCE_API = ('clone', 'valid_moves', 'play_action', 'win_state', 'observation', '__eq__')


class GameState:
    """Base class of every env.  Subclasses must provide the members listed in the module docstring; the check is
    made once per subclass (at class creation) instead of per instantiation."""

    AZG_GAME_ID = None          # set by games that have device rule kernels
    _abstract = True

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        cls._abstract = False
        missing = [n for n in _CLASS_API + _INSTANCE_API if getattr(cls, n, None) is getattr(GameState, n, None)]
        if missing:
            cls._abstract = True
            cls._missing = tuple(missing)

    def __new__(cls, *a, **k):
        if cls._abstract:
            raise TypeError("Can't instantiate %s: missing %s" % (cls.__name__, ', '.join(getattr(cls, '_missing', ('everything',)))))
        return super().__new__(cls)

    # ---- storage shared by the envs -----------------------------------------------------------------------
    def __init__(self, board):
        self._board, self._player, self._turns = board, 0, 0
        self.last_action = None

    player = property(lambda self: self._player, doc='index of the player to move')
    turns = property(lambda self: self._turns, doc='moves played so far')

    def _next_player(self, player, turns=1):
        return (player + turns) % self.num_players()

    def _update_turn(self):
        """called by play_action after the move is on the board"""
        self._turns += 1
        self._player = self._next_player(self._player)

    def __str__(self):
        return 'Player:\t%s\n%s\n' % (self._player, self._board)

    # ---- class-level description (defaults for the optional ones) ------------------------------------------
    @staticmethod
    def action_size():
        raise NotImplementedError

    @staticmethod
    def observation_size():
        """(channels, height, width) of observation()"""
        raise NotImplementedError

    @staticmethod
    def num_players():
        raise NotImplementedError

    @staticmethod
    def max_turns():
        """turn count at which the game is declared drawn, None if unbounded"""
        return None

    @staticmethod
    def has_draw():
        return True

    # ---- position interface ---------------------------------------------------------------------------------
    def __eq__(self, other):
        raise NotImplementedError

    __hash__ = None

    def clone(self):
        raise NotImplementedError

    def valid_moves(self):
        """uint8/0-1 array of length action_size()"""
        raise NotImplementedError

    def play_action(self, action):
        """subclasses call super().play_action(action) first, then move and _update_turn()"""
        self.last_action = action

    def win_state(self):
        """bool/uint8 array [player 0 won, ..., player P-1 won, draw]"""
        raise NotImplementedError

    def observation(self):
        """float32 array of shape observation_size()"""
        raise NotImplementedError

    def symmetries(self, pi):
        """[(state, pi)] equivalent positions for sample augmentation; optional (args.symmetricSamples)"""
        raise NotImplementedError('Symmetries not implemented for this environment. Set symmetricSamples to False in args.')


_REFERENCE_ENVS = {'envs.connect4.connect4': 0, 'envs.brandubh.fastafl': 1}


def azg_game_id(game_cls_or_state):
    """Device game id of a GameState class or instance.  The reference's own env classes are recognised by module
    name, so alphazero.envs.connect4.connect4.Game objects can be searched unchanged."""
    gid = getattr(game_cls_or_state, 'AZG_GAME_ID', None)
    if gid is not None:
        return gid
    cls = game_cls_or_state if isinstance(game_cls_or_state, type) else type(game_cls_or_state)
    mod = getattr(cls, '__module__', '')
    for suffix, g in _REFERENCE_ENVS.items():
        if mod.endswith(suffix):
            return g
    msg = '%s.%s has no device rule kernels registered (csrc/azg_games.h); the MI355X engine has no CPU search fallback' % (mod, cls.__name__)
    # (such games are handed to the reference's own classes on the reference side -- alphazero_general_amd.reference_class; if that import
    #  failed, say why: a missing checkout, pyximport, a Cython build error)
    from . import reference_import_error
    for name in ('MCTS', 'SelfPlayAgent'):
        ex = reference_import_error(name)
        if ex is not None:
            err = NotImplementedError('%s, and the hand-over to the reference\'s own alphazero.%s failed: %s: %s' % (msg, name, type(ex).__name__, ex))
            raise err from ex
    raise NotImplementedError(msg)


def has_device_rules(game_cls_or_state):
    try:
        azg_game_id(game_cls_or_state)
        return True
    except NotImplementedError:
        return False
