"""Time the fused search kernel (azg_search_f16) against the three-launch path: connect4, 2048 games x 100 sims."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_general_amd.engine import DeviceEngine
from alphazero_general_amd.envs.connect4 import Game
from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sims, rounds = 100, int(sys.argv[2]) if len(sys.argv) > 2 else 12
torch.manual_seed(0)
net = NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0'); net.refresh()
kw = dict(cpuct=4.0, fpu_reduction=0.4, add_root_noise=True, add_root_temp=True, seed=0, games_per_iteration=1 << 30,
          example_capacity=B * 43 * 2 * 4, sims_hint=sims)
for mode in ('fused', 'three'):
    e = DeviceEngine(0, B, **kw)
    obs = torch.zeros((B, 42, 8), dtype=torch.float16, device=e.device)
    def round_():
        if mode == 'fused':
            net._hip.search(e, sims)
        else:
            for _ in range(sims):
                e.select(obs); p, v = net._hip.forward_nhwc8(obs); e.backup(p, v)
        e.advance(True)
    for _ in range(3): round_()
    torch.cuda.synchronize(); c0 = e.counters(); t0 = time.time()
    for _ in range(rounds): round_()
    torch.cuda.synchronize(); dt = time.time() - t0; c1 = e.counters()
    print(mode, 'ms/round %.2f' % (dt / rounds * 1e3), 'expansions/s %.0f' % ((c1['expansions'] - c0['expansions']) / dt))
