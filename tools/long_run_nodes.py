"""How full do the tree node stores get in long self-play runs?  (sizing evidence for DESIGN.md section 2)"""
import os, sys, importlib
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from alphazero_general_amd import nnet as N, _abi
from alphazero_general_amd.selfplay import SelfPlayRunner
from alphazero_general_amd.utils import dotdict, default_temp_scaling
for game, netargs, B, sims, cpuct, fpu, rounds in (('brandubh', N.BRANDUBH_NET_ARGS, 512, 200, 1.25, 0.2, 160), ('connect4', N.CONNECT4_NET_ARGS, 2048, 100, 4.0, 0.4, 120), ('trimok', N.DEFAULT_NET_ARGS, 256, 50, 1.25, 0.2, 120)):
    Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
    torch.manual_seed(0)
    net = N.NNetWrapper(Game, netargs, device='cuda:0', dtype=torch.float16)
    args = dotdict(cpuct=cpuct, fpu_reduction=fpu, root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1.0, numMCTSSims=sims, numFastSims=20,
                   numWarmupSims=5, probFastSim=0.0, gamesPerIteration=1 << 30, add_root_noise=True, add_root_temp=True, symmetricSamples=True,
                   mctsResetThreshold=None, startTemp=1.0, arenaTemp=0.25, temp_scaling_fn=default_temp_scaling)
    gi = _abi.game_info(Game.AZG_GAME_ID)
    r = SelfPlayRunner(Game, net, args, num_slots=B, seed=0, example_capacity=B * 6 * (gi.max_turns + 1) * gi.num_symmetries)
    r.prepare()
    peak = 0
    kept = 0
    for i in range(rounds):
        r.play_round()
        if i % 10 == 9:
            c = r.counters(); peak = max(peak, c['max_nodes_used']); kept = max(kept, c['max_nodes_kept'])
    c = r.counters()
    cap = 8 * sims * gi.max_children + 64
    print(game, 'rounds', rounds, 'games', c['games_played'], 'max live nodes', max(peak, c['max_nodes_used']), 'largest subtree kept by a compaction', max(kept, c['max_nodes_kept']), 'of cap', cap, '(one move = %d)' % (sims * gi.max_children), flush=True)
    del r, net
    torch.cuda.empty_cache()
