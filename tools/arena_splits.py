import os, sys
sys.path.insert(0, os.getcwd())
import torch
from alphazero_general_amd import nnet as N
from alphazero_general_amd.envs.connect4 import Game
nets = []
for sd in (0, 1):
    torch.manual_seed(sd)
    n = N.NNetWrapper(Game, N.CONNECT4_NET_ARGS, device='cuda:0'); n.refresh(); nets.append(n._hip)
print('weight buffers at', [hex(n.tower_w.data_ptr()) for n in nets])
for split in ((0, 256), (64, 192), (96, 160), (112, 144), (120, 136), (127, 129), (128, 128), (129, 127), (136, 120), (160, 96), (256, 0), (64, 64), (128, 0), (100, 100)):
    B = sum(split)
    x = torch.zeros((B, 42, 8), dtype=torch.float16, device='cuda:0'); x[:, :, :3] = (torch.rand(B, 42, 3, device='cuda:0') > 0.6).half()
    pol = torch.zeros((B, 7), device='cuda:0'); val = torch.zeros((B, 3), device='cuda:0')
    rpm = torch.tensor(split, dtype=torch.int32, device='cuda:0')
    for _ in range(5): N.HipResNet.forward_models(nets, x, pol, val, rpm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): N.HipResNet.forward_models(nets, x, pol, val, rpm)
    e1.record(); torch.cuda.synchronize()
    print('%3d+%3d rows: %.1f us' % (split[0], split[1], e0.elapsed_time(e1) * 5))
