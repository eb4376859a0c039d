"""Run only the network forward (MFMA tower + heads) a few times -- target for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_general_amd.envs.connect4 import Game
from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
net = NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0')
net.refresh()
x = (torch.rand(B, 42, 8, device='cuda:0') > 0.5).half()
for _ in range(n):
    p, v = net._hip.forward_nhwc8(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    p, v = net._hip.forward_nhwc8(x)
e1.record(); torch.cuda.synchronize()
print('forward ms', e0.elapsed_time(e1) / n)
