"""Cycles per simulation and phase (tree, tower, head convolutions, logits GEMM) of the persistent wide-head search launch --
needs the measurement build (hipcc ... -DAZG_TOWER_TIMING -o alphazero_general_amd/lib/libazg_timing.so; AZG_LIB_PATH=<that>).
The library prints the phase means to stderr after every launch.  usage: wide_search_phases.py [games] [brandubh|trimok] [exact]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from alphazero_general_amd import nnet as N
from alphazero_general_amd.engine import DeviceEngine
import importlib
game = sys.argv[2] if len(sys.argv) > 2 else 'brandubh'
Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
torch.manual_seed(0)
net = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS if game == 'brandubh' else N.DEFAULT_NET_ARGS, device='cuda:0'); net.refresh()
B = int(sys.argv[1]) if len(sys.argv) > 1 else (512 if game == 'brandubh' else 256)
sims = 200 if game == 'brandubh' else 50
e = DeviceEngine(Game.AZG_GAME_ID, B, cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=0, sims_hint=sims, example_capacity=B * 808 * 2)
exact = len(sys.argv) > 3 and sys.argv[3] == 'exact'
net._hip.search(e, 0, exact=exact)                           # one-time set-up: the tile is measured here
print('TILE %s' % (net._hip.search_tile(e, exact=exact) or {}).get('games_per_workgroup'), flush=True)
for mv in range(6):
    net._hip.search(e, sims, exact=exact); e.advance(True)
torch.cuda.synchronize()
