"""Cycles per simulation and phase (tree, tower, head convolutions, logits GEMM) of the persistent wide-head search launch --
needs the measurement build (hipcc ... -DAZG_TOWER_TIMING -o alphazero_general_amd/lib/libazg_timing.so; AZG_LIB_PATH=<that>).
The library prints the phase means to stderr after every launch.  usage: wide_search_phases.py [games]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from alphazero_general_amd import nnet as N
from alphazero_general_amd.engine import DeviceEngine
from alphazero_general_amd.envs.brandubh import Game
torch.manual_seed(0)
net = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS, device='cuda:0'); net.refresh()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
e = DeviceEngine(1, B, cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=0, sims_hint=200, example_capacity=B * 808 * 2)
for mv in range(6):
    net._hip.search(e, 200); e.advance(True)
torch.cuda.synchronize()
