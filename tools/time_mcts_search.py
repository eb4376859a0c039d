"""Wall time of MCTS.search (the reference's single-tree API, MCTS.pyx:165-173: what GenericPlayers.MCTSPlayer.play calls once per move) on
this package's class: nn = NNetWrapper -> ONE persistent launch per call; nn = a plain callable -> find_leaf / nn(obs) /
process_results per simulation (3 launches + host syncs each).  usage: time_mcts_search.py [connect4|brandubh|trimok] [sims]"""
import importlib
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_general_amd import nnet as N
from alphazero_general_amd.MCTS import MCTS
from alphazero_general_amd.utils import dotdict
game = sys.argv[1] if len(sys.argv) > 1 else 'connect4'
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 100
Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
na = {'connect4': N.CONNECT4_NET_ARGS, 'brandubh': N.BRANDUBH_NET_ARGS, 'trimok': N.DEFAULT_NET_ARGS}[game]
torch.manual_seed(0)
net = N.NNetWrapper(Game, na, device='cuda:0', dtype=torch.float16); net.refresh()
args = dotdict(root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1, fpu_reduction=0.2, cpuct=1.25, _num_players=Game.num_players() + 1,
               numMCTSSims=sims, _azg_seed=1)
for label, nn in (('persistent launch (nn = NNetWrapper)', net), ('per-simulation loop (nn = plain callable)', lambda o: net.predict(o))):
    m, g = MCTS(args), Game()
    m.search(g, nn, sims, True, True)                                   # warm
    torch.cuda.synchronize()
    t0, n = time.time(), 0
    while n < 8 and not g.win_state().any():
        m.search(g, nn, sims, True, True)
        a = m.best_action(g); m.update_root(g, a); g.play_action(a); n += 1
    torch.cuda.synchronize()
    print('%s %s, %d simulations per search: %.2f ms per move (search + best_action + update_root), %d moves' % (game, label, sims, (time.time() - t0) * 1e3 / max(n, 1), n), flush=True)
