import ctypes as C, importlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from alphazero_general_amd import _abi, nnet as N
from alphazero_general_amd.engine import DeviceEngine
game = sys.argv[1] if len(sys.argv) > 1 else 'brandubh'
B, sims, netargs = {'brandubh': (512, 200, N.BRANDUBH_NET_ARGS), 'trimok': (256, 50, N.DEFAULT_NET_ARGS)}[game]
Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
torch.manual_seed(0)
net = N.NNetWrapper(Game, netargs, device='cuda:0', dtype=torch.float16); net.refresh()
e = DeviceEngine(Game.AZG_GAME_ID, B, cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=0, sims_hint=sims, example_capacity=B * 808 * 2)
L = _abi.lib()
L.azg_debug_tree_timing.argtypes = [C.c_void_p, C.c_void_p]
rows = []
for mv in range(10):
    for s in (sims - 6, 2, 2, 2):
        net._hip.search(e, s)
        buf = np.zeros((B, 16), np.uint64)
        _abi.check(L.azg_debug_tree_timing(e.h, buf.ctypes.data_as(C.c_void_p)))
        b = buf.astype(np.int64)
        ok = (b[:, 7] > b[:, 8]) & (b[:, 7] - b[:, 8] < 200000) & (b[:, 4] >= b[:, 3]) & (b[:, 5] >= b[:, 4]) & (b[:, 6] >= b[:, 5])
        if mv < 2 or s > 2: continue
        rows.append(b[ok])
    e.advance(True)
b = np.concatenate(rows)
depth = b[:, 12] >> 48; b[:, 12] &= (1 << 48) - 1
n = len(b)
print(game, n, 'samples, mean depth %.2f' % depth.mean())
print(' descent total      %9.1f' % (b[:, 5] - b[:, 4]).mean())
print('   block load waits %9.1f' % b[:, 10].mean())
print('   best_child math  %9.1f' % b[:, 11].mean())
print('   publish          %9.1f' % b[:, 12].mean())
exp = b[:, 13] > b[:, 5]
print(' expansion total    %9.1f' % (b[:, 6] - b[:, 5]).mean())
print('   wait for rules   %9.1f (%d)' % ((b[exp, 13] - b[exp, 5]).mean(), exp.sum()))
print('   add_children     %9.1f' % (b[exp, 14] - b[exp, 13]).mean())
print('   header store     %9.1f' % (b[exp, 6] - b[exp, 14]).mean())
print(' TOTAL              %9.1f' % (b[:, 7] - b[:, 8]).mean())
