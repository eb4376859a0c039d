"""Where the walker wavefront of the persistent wide-head search launch spends a tree phase (s_memtime stamps of every slot's LAST
simulation of a move; needs the tree-timing build: tools/build_timing.sh tree, AZG_LIB_PATH=.../libazg_timing.so).
Stamps: 8 phase start, 1 value logits + softmax done, 3 path backed up, 4 descent starts, [9 -> 0 waited for the previous leaf's
priors], 5 descent done, [2 -> 15 waited for the shuffle masks], 6 expansion done, 7 leaf stored.   usage: [brandubh|trimok] [games] [exact]
(tiles of several games per workgroup: ONE wavefront does a game's whole tree phase -- its softmax + priors sit between stamps 8 and 1)"""
import ctypes as C
import importlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from alphazero_general_amd import _abi, nnet as N
from alphazero_general_amd.engine import DeviceEngine
game = sys.argv[1] if len(sys.argv) > 1 else 'brandubh'
B, sims, netargs = {'brandubh': (512, 200, N.BRANDUBH_NET_ARGS), 'trimok': (256, 50, N.DEFAULT_NET_ARGS)}[game]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
exact = len(sys.argv) > 3 and sys.argv[3] == 'exact'
Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
torch.manual_seed(0)
net = N.NNetWrapper(Game, netargs, device='cuda:0', dtype=torch.float16); net.refresh()
e = DeviceEngine(Game.AZG_GAME_ID, B, cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=0, sims_hint=sims, example_capacity=B * 808 * 2)
L = _abi.lib()
L.azg_debug_tree_timing.argtypes = [C.c_void_p, C.c_void_p]
seg = [('value logits + softmax', 8, 1), ('path backup', 1, 3), ('to descent start', 3, 4), ('descent (incl. waits)', 4, 5), ('expansion (incl. waits)', 5, 6), ('leaf store', 6, 7), ('TOTAL', 8, 7)]
acc = {k: 0.0 for k, _, _ in seg}; n = 0; wp = wm = 0.0; np_ = nm = 0
for mv in range(10):
    for s in (sims - 6, 2, 2, 2):                    # the move in pieces: the last backup + select of each piece leaves its stamps
        net._hip.search(e, s, exact=exact)
        buf = np.zeros((B, 16), np.uint64)
        _abi.check(L.azg_debug_tree_timing(e.h, buf.ctypes.data_as(C.c_void_p)))
        b = buf.astype(np.int64)
        ok = (b[:, 7] > b[:, 8]) & (b[:, 7] - b[:, 8] < 200000) & (b[:, 4] >= b[:, 3]) & (b[:, 5] >= b[:, 4]) & (b[:, 6] >= b[:, 5])
        if mv < 2 or s > 2:
            continue
        for k, i0, i1 in seg:
            acc[k] += (b[ok, i1] - b[ok, i0]).sum()
        n += ok.sum()
        w = ok & (b[:, 9] > b[:, 8]) & (b[:, 0] >= b[:, 9]) & (b[:, 0] <= b[:, 7])
        wp += (b[w, 0] - b[w, 9]).sum(); np_ += w.sum()
        w = ok & (b[:, 2] > b[:, 8]) & (b[:, 15] >= b[:, 2]) & (b[:, 15] <= b[:, 7])
        wm += (b[w, 15] - b[w, 2]).sum(); nm += w.sum()
    e.advance(True)
print('%s, walker wavefront, %d samples, shader cycles:' % (game, n))
for k, _, _ in seg:
    print('  %-28s %9.1f' % (k, acc[k] / max(n, 1)))
print('  waited for priors   in %4.1f %% of the simulations, %8.1f cycles each' % (100.0 * np_ / max(n, 1), wp / max(np_, 1)))
print('  waited for masks    in %4.1f %% of the simulations, %8.1f cycles each' % (100.0 * nm / max(n, 1), wm / max(nm, 1)))
