"""Network evaluation time of the narrow towers (brandubh 64ch, 3-player 32ch) at their BASELINE batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_general_amd import nnet as N
for game, args, HW, sizes in (('brandubh', 'BRANDUBH_NET_ARGS', 49, (256, 512, 1024, 2048)), ('trimok', 'DEFAULT_NET_ARGS', 25, (256, 1024))):
    Game = __import__('alphazero_general_amd.envs.' + game, fromlist=['Game']).Game
    net = N.NNetWrapper(Game, getattr(N, args), device='cuda:0'); net.refresh()
    out = []
    for B in sizes:
        x = (torch.rand(B, HW, 8, device='cuda:0') > 0.5).half()
        for _ in range(5): net._hip.forward_nhwc8(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): net._hip.forward_nhwc8(x)
        e1.record(); torch.cuda.synchronize()
        out.append('%d:%.1f' % (B, e0.elapsed_time(e1) / 50 * 1000))
    print(game, 'psplit', os.environ.get('AZG_TOWER_PSPLIT', 'auto'), 'boards', os.environ.get('AZG_TOWER_BOARDS', 'auto'), 'us', ' '.join(out))
