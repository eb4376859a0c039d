"""Launch time of the fused tower + heads as a function of the network's depth at a fixed batch: slope = one residual block, intercept =
prologue + stem + heads + launch overhead.   usage: tower_intercept.py [game] [boards]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alphazero_general_amd import nnet as N
from alphazero_general_amd.utils import dotdict
game, B = (sys.argv[1] if len(sys.argv) > 1 else 'connect4'), int(sys.argv[2]) if len(sys.argv) > 2 else 256
Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
base = {'brandubh': N.BRANDUBH_NET_ARGS, 'connect4': N.CONNECT4_NET_ARGS, 'trimok': N.DEFAULT_NET_ARGS}[game]
hw = Game.observation_size()[1] * Game.observation_size()[2]
x = torch.zeros((B, hw, 8), dtype=torch.float16, device='cuda:0'); x[:, :, :3] = (torch.rand(B, hw, 3, device='cuda:0') > 0.7).half()
res = []
for depth in (0, 2, 4, 8):
    args = dotdict(dict(base)); args.depth = depth
    torch.manual_seed(0)
    net = N.NNetWrapper(Game, args, device='cuda:0'); net.refresh()
    for _ in range(20): net._hip.forward_nhwc8(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): net._hip.forward_nhwc8(x)
    e1.record(); torch.cuda.synchronize()
    res.append((depth, e0.elapsed_time(e1) / 200 * 1000))
    print('%s %d boards, %d blocks: %.2f us per launch' % (game, B, depth, res[-1][1]))
d = np.array(res)
slope, icpt = np.polyfit(d[:, 0], d[:, 1], 1)
print('per block %.2f us, intercept %.2f us' % (slope, icpt))
