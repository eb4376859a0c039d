"""Debug aid: the state of a brandubh engine after `sims` simulations of one persistent exact launch, dumped to a .npz -- run once per
library (AZG_LIB_PATH) and compare.  usage: ovl_diff.py <out.npz> [games] [sims]"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from alphazero_general_amd import nnet as N
from alphazero_general_amd.engine import DeviceEngine
from alphazero_general_amd.envs.brandubh import Game
out, B, sims = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 512, int(sys.argv[3]) if len(sys.argv) > 3 else 200
torch.manual_seed(0)
net = N.NNetWrapper(Game, N.BRANDUBH_NET_ARGS, device='cuda:0'); net.refresh()
e = DeviceEngine(Game.AZG_GAME_ID, B, cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=3, sims_hint=200, example_capacity=B * 808 * 2)
net._hip.search(e, 0, exact=True)
net._hip.search(e, sims, exact=True)
torch.cuda.synchronize()
c = e.counters()
np.savez(out, counts=e.root_counts().cpu().numpy(), tape=e.tape_counters(), sims=c['sims'], exp=c['expansions'])
print(out, c['sims'], c['expansions'], e.root_counts().sum().item())
