"""Phase breakdown of the brandubh tree launch (azg_backup_select_logits) from s_memtime stamps.

    hipcc ... -DAZG_TREE_TIMING -o gpurun_out/libazg_timing.so      (tools/build_timing.sh)
    AZG_LIB_PATH=gpurun_out/libazg_timing.so python tools/time_tree.py [brandubh|trimok]

Stamps (shader cycles, last simulation of every slot).  Walk wave: 8 entry, 1 header / path / value row landed, 3 path stores
issued, 4 descent starts, 5 descent done, 6 expansion done, 7 leaf stored.  Prior wave: 9 = softmax + leaf policy done, 0 = shuffle masks ready."""
import ctypes as C
import importlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from alphazero_general_amd import _abi, nnet as N
from alphazero_general_amd.selfplay import SelfPlayRunner
from alphazero_general_amd.utils import dotdict, default_temp_scaling

game = sys.argv[1] if len(sys.argv) > 1 else 'brandubh'
B, sims, netargs = {'brandubh': (512, 200, N.BRANDUBH_NET_ARGS), 'trimok': (256, 50, N.DEFAULT_NET_ARGS)}[game]
Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
torch.manual_seed(0)
net = N.NNetWrapper(Game, netargs, device='cuda:0', dtype=torch.float16)
args = dotdict(cpuct=1.25, fpu_reduction=0.2, root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1.0, numMCTSSims=sims, numFastSims=20,
               numWarmupSims=5, probFastSim=0.0, gamesPerIteration=1 << 30, add_root_noise=True, add_root_temp=True, symmetricSamples=True,
               mctsResetThreshold=None, startTemp=1.0, arenaTemp=0.25, temp_scaling_fn=default_temp_scaling)
r = SelfPlayRunner(Game, net, args, num_slots=B, seed=0, example_capacity=B * 101 * 8 * 2)
r.prepare()
L = _abi.lib()
L.azg_debug_tree_timing.argtypes = [C.c_void_p, C.c_void_p]
names = ['loads', 'path stores', 'fence', 'descent', 'expansion', 'leaf store']
order = [8, 1, 3, 4, 5, 6, 7]
acc = np.zeros(len(names)); tot = 0.0; dep = 0.0; n = 0; w1 = 0.0; w0 = 0.0; extra = {}; nx = 0
ln, e = r.lanes[0], r.engine
for rnd in range(8):
    r.play_round()                                               # (advance the games: realistic trees)
    e.select(ln.obs)                                             # then one move of eager launches, every one backup + select
    for s_ in range(sims - 1):
        e.backup_select_logits(ln.net.run_logits(), ln.obs, select=True)
        if s_ % 16 != 15 or rnd < 2:
            continue
        buf = np.zeros((B, 16), np.uint64)
        _abi.check(L.azg_debug_tree_timing(e.h, buf.ctypes.data_as(C.c_void_p)))
        t = buf[:, order].astype(np.int64)
        d = np.diff(t, axis=1)
        ok = (d >= 0).all(axis=1) & (t[:, 0] > 0) & (t[:, -1] - t[:, 0] < 200000)
        b = buf.astype(np.int64)
        okx = ok & (b[:, 15] >= 1) & (b[:, 10] < b[:, 11]) & (b[:, 11] < b[:, 12]) & (b[:, 12] <= b[:, 5]) & (b[:, 13] > b[:, 5]) & (b[:, 14] > b[:, 13])
        for nm, (i0, i1) in {'last level: block load': (4, 10), 'last level: best_child': (10, 11), 'last level: play': (11, 12),
                             'expansion: win + valid list': (5, 13), 'expansion: add_children': (13, 14), 'expansion: header stores': (14, 6)}.items():
            if nm == 'last level: block load':
                sel = okx & (b[:, 15] == 1)
                extra[nm + ' (depth-1 slots)'] = extra.get(nm + ' (depth-1 slots)', 0.0) + (b[sel, i1] - b[sel, i0]).sum() * (okx.sum() / max(sel.sum(), 1))
            else:
                extra[nm] = extra.get(nm, 0.0) + (b[okx, i1] - b[okx, i0]).sum()
        nx += okx.sum()
        w0 += (buf[ok, 0].astype(np.int64) - t[ok, 0]).sum()
        w1 += (buf[ok, 9].astype(np.int64) - t[ok, 0]).sum(); acc += d[ok].sum(0); tot += (t[ok, -1] - t[ok, 0]).sum(); dep += buf[ok, 15].astype(np.float64).sum(); n += ok.sum()
    e.backup_select_logits(ln.net.run_logits(), None, select=False)
    e.advance(True)
print('%s: %d samples, mean depth %.2f, shader cycles per phase:' % (game, n, dep / n))
for nm, v in zip(names, acc / n):
    print('  %-14s %8.1f' % (nm, v))
print('  %-14s %8.1f' % ('total (walk)', tot / n))
print('  %-14s %8.1f' % ('priors ready', w1 / n))
print('  %-14s %8.1f' % ('shuffle ready', w0 / n))
for k_, v_ in sorted(extra.items()):
    print('  %-28s %8.1f' % (k_, v_ / max(nx, 1)))
