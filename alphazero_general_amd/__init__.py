"""alphazero_general_amd -- MI355X-native batched self-play / MCTS engine behind the alphazero-general API.

The product path is HIP only (csrc/ -> lib/libazg_hip.so through the C ABI of include/azg.h); importing the
package does not load the library, the first engine call does and fails loudly if it is missing.
"""
__version__ = '0.1.0'
