"""alphazero_general_amd -- MI355X-native batched self-play / MCTS engine behind the alphazero-general API.

The product path is HIP only (csrc/ -> lib/libazg_hip.so through the C ABI of include/azg.h); importing the
package does not load the library, the first engine call does and fails loudly if it is missing.
"""
__version__ = '0.1.0'


def install():
    """Make the reference's callers use this engine unchanged: register this package's MCTS / SelfPlayAgent modules
    under the names `alphazero.MCTS` and `alphazero.SelfPlayAgent`, so that `from alphazero.MCTS import MCTS`
    (GenericPlayers.py:1, SelfPlayAgent.pyx:10, Evaluator.py) and `from alphazero.SelfPlayAgent import SelfPlayAgent`
    (Coach.py, Arena.pyx) resolve here.  Call before importing alphazero.Coach / alphazero.Arena.  See INTEGRATION.md."""
    import importlib
    import sys
    mcts = importlib.import_module(__name__ + '.MCTS')
    sys.modules['alphazero.MCTS'] = mcts
    try:
        agent = importlib.import_module(__name__ + '.SelfPlayAgent')
        sys.modules['alphazero.SelfPlayAgent'] = agent
    except ImportError:
        pass
    return mcts
