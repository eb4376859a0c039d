"""Per-layer s_memtime stamps of the tower kernel (workgroup 0) -- needs the measurement build:
    hipcc ... -DAZG_TOWER_TIMING -o alphazero_general_amd/lib/libazg_timing.so ;  AZG_LIB_PATH=<that> python tools/tower_stamps.py brandubh 512
The library prints `layer L wave W: main / wait / epi / bar` cycles to stderr on the 8th launch."""
import importlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_general_amd import nnet as N
game, B = (sys.argv[1] if len(sys.argv) > 1 else 'brandubh'), int(sys.argv[2]) if len(sys.argv) > 2 else 512
Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
args = {'brandubh': N.BRANDUBH_NET_ARGS, 'connect4': N.CONNECT4_NET_ARGS, 'trimok': N.DEFAULT_NET_ARGS}[game]
torch.manual_seed(0)
net = N.NNetWrapper(Game, args, device='cuda:0'); net.refresh()
hw = Game.observation_size()[1] * Game.observation_size()[2]
x = torch.zeros((B, hw, 8), dtype=torch.float16, device='cuda:0'); x[:, :, :3] = (torch.rand(B, hw, 3, device='cuda:0') > 0.7).half()
for _ in range(12):
    net._hip.forward_nhwc8(x)
torch.cuda.synchronize()
