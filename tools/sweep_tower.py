"""Tower time vs batch size and boards-per-tile (run once per AZG_TOWER_BOARDS value; the choice is read at first launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
game = sys.argv[1] if len(sys.argv) > 1 else 'connect4'
if game == 'connect4':
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import CONNECT4_NET_ARGS as NA, NNetWrapper
    HW = 42
else:
    from alphazero_general_amd.envs.brandubh import Game
    from alphazero_general_amd.nnet import BRANDUBH_NET_ARGS as NA, NNetWrapper
    HW = 49
net = NNetWrapper(Game, NA, device='cuda:0'); net.refresh()
out = []
for B in (1, 64, 128, 256, 512, 768, 1024, 1536, 2048, 4096):
    x = (torch.rand(B, HW, 8, device='cuda:0') > 0.5).half()
    for _ in range(5): net._hip.forward_nhwc8(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): net._hip.forward_nhwc8(x)
    e1.record(); torch.cuda.synchronize()
    out.append('%d:%.1f' % (B, e0.elapsed_time(e1) / 30 * 1000))
print(game, 'boards/tile', os.environ.get('AZG_TOWER_BOARDS', 'auto'), 'us', ' '.join(out))
