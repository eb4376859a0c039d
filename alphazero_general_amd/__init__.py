"""alphazero_general_amd -- MI355X-native batched self-play / MCTS engine behind the alphazero-general API.

The product path is HIP only (csrc/ -> lib/libazg_hip.so through the C ABI of include/azg.h); importing the
package does not load the library, the first engine call does and fails loudly if it is missing.
"""
__version__ = '0.1.0'

_installed = {}        # 'MCTS' / 'SelfPlayAgent' -> this package's module registered under the reference's name by install()
_reference = {}        # 'MCTS' / 'SelfPlayAgent' -> the reference's own class, loaded on first need (reference_class)
_reference_mod = {}    # ... and the module it lives in
_reference_error = {}  # ... or the exception its import raised (reference_import_error)
import threading as _threading
_reference_lock = _threading.RLock()


def install():
    """Make the reference's callers use this engine unchanged: register this package's MCTS / SelfPlayAgent modules
    under the names `alphazero.MCTS` and `alphazero.SelfPlayAgent`, so that `from alphazero.MCTS import MCTS`
    (GenericPlayers.py:1, SelfPlayAgent.pyx:10, Evaluator.py) and `from alphazero.SelfPlayAgent import SelfPlayAgent`
    (Coach.py, Arena.pyx) resolve here.  Call before importing alphazero.Coach / alphazero.Arena.  See INTEGRATION.md.

    Dispatch is per GAME (SURVEY.md 8b "Game plugin"): a game with device rule kernels (Game.azg_game_id) is searched by this
    engine; for any other env of the reference (tictactoe, othello, gobang, ...) the classes registered here hand over to the
    REFERENCE'S OWN alphazero.MCTS.MCTS / alphazero.SelfPlayAgent.SelfPlayAgent (reference_class), so those envs keep working
    exactly as before -- this package contains no CPU search."""
    import importlib
    import sys
    mcts = importlib.import_module(__name__ + '.MCTS')
    sys.modules['alphazero.MCTS'] = mcts
    _installed['MCTS'] = mcts
    try:
        agent = importlib.import_module(__name__ + '.SelfPlayAgent')
        sys.modules['alphazero.SelfPlayAgent'] = agent
        _installed['SelfPlayAgent'] = agent
    except ImportError:
        pass
    return mcts


def reference_class(name):
    """The reference's own `alphazero.<name>.<name>` class (name = 'MCTS' or 'SelfPlayAgent'), or None when the reference package
    is not importable.  Loaded on first need, reference side: the module registered by install() steps aside for the import and
    is put back, so `from alphazero.MCTS import MCTS` keeps resolving to this package."""
    if name in _reference:
        return _reference[name]
    import importlib
    import sys
    full = 'alphazero.' + name
    with _reference_lock:                                      # (the swap below must not interleave with another thread's import of the name)
        if name in _reference:
            return _reference[name]
        ours = sys.modules.get(full)
        if ours is not None and not (getattr(ours, '__name__', '') or '').startswith(__name__):
            _reference[name], _reference_mod[name] = getattr(ours, name, None), ours   # install() was never called: the name IS the reference's module
            return _reference[name]
        cls = None
        try:
            sys.modules.pop(full, None)
            mod = importlib.import_module(full)
            cls = getattr(mod, name, None)
            _reference_mod[name] = mod
        except Exception as ex:                                # noqa: BLE001 (no reference checkout, no pyximport, build failure)
            cls = None
            _reference_error[name] = ex                        # kept: the caller's NotImplementedError names the real cause
        finally:
            if ours is not None:
                sys.modules[full] = ours
                pkg = sys.modules.get('alphazero')
                if pkg is not None:
                    setattr(pkg, name, ours)
            # (install() was never called and the name was not imported yet: the reference's module simply stays imported)
        _reference[name] = cls
        return cls


def reference_import_error(name):
    """why reference_class(name) returned None (the exception its import raised), or None"""
    return _reference_error.get(name)


class reference_module:
    """Context manager: while active, `alphazero.<name>` in sys.modules IS the reference's module.  Pickling an object of a Cython
    `auto_pickle` class resolves its class and its __pyx_unpickle_* helper by NAME in that module (MCTS.pyx:8), so the hand-over
    objects are pickled / unpickled inside this context."""

    def __init__(self, name):
        self.name, self.full = name, 'alphazero.' + name

    def __enter__(self):
        import sys
        if reference_class(self.name) is None:
            raise ImportError('the reference package (alphazero.%s) is not importable' % self.name)
        self.saved = sys.modules.get(self.full)
        sys.modules[self.full] = _reference_mod[self.name]
        return _reference_mod[self.name]

    def __exit__(self, *exc):
        import sys
        if self.saved is not None:
            sys.modules[self.full] = self.saved
        return False
