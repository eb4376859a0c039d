import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from alphazero_general_amd import nnet as N
from alphazero_general_amd.engine import DeviceEngine
from alphazero_general_amd.envs.brandubh import Game
torch.manual_seed(0)
net = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS, device='cuda:0'); net.refresh()
e = DeviceEngine(1, 512, cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=0, sims_hint=200, example_capacity=512 * 808 * 2)
for mv in range(6):
    net._hip.search(e, 200); e.advance(True)
torch.cuda.synchronize()
