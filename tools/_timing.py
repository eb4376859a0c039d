import os, sys, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_general_amd import _abi
_abi.LIB_PATH = _abi.LIB_PATH.replace('libazg_hip.so', 'libazg_hip_timing.so')
import torch
from alphazero_general_amd.envs.connect4 import Game
from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper
torch.manual_seed(0)
net = NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0'); net.refresh()
x = (torch.rand(2048, 42, 8, device='cuda:0') > 0.5).half()
for _ in range(5):
    net._hip.forward_nhwc8(x)
torch.cuda.synchronize()
os.environ['AZG_TOWER_DUMP'] = '1'
net._hip.forward_nhwc8(x)
torch.cuda.synchronize()
