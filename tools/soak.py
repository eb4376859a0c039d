"""Soak run of the native runners (persistent search launches, whole rounds replayed as hipGraphs): thousands of rounds with the node
stores compacting and the output buffers recycled, checking the sticky device error word and the counters' arithmetic every 50 rounds.
usage: soak.py [scale]   (scale 1.0: connect4 1500 rounds, brandubh 1200, 3-player env 4000, arena 600)"""
import importlib
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphazero_general_amd import nnet as N, _abi
from alphazero_general_amd.selfplay import ArenaRunner, SelfPlayRunner
from alphazero_general_amd.utils import dotdict, default_temp_scaling
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0


def args_for(sims, cpuct, fpu):
    return dotdict(cpuct=cpuct, fpu_reduction=fpu, root_noise_frac=0.1, root_policy_temp=1.1, min_discount=1.0, numMCTSSims=sims, numFastSims=20,
                   numWarmupSims=5, probFastSim=0.0, gamesPerIteration=1 << 30, add_root_noise=True, add_root_temp=True, symmetricSamples=True,
                   mctsResetThreshold=None, startTemp=1.0, arenaTemp=0.25, temp_scaling_fn=default_temp_scaling)


# (round 5: every tile shape of the persistent launches -- 1 / 2 / 3 / 4 games per workgroup, exact heads by default -- and the small connect4 engines)
for game, netargs, B, sims, cpuct, fpu, rounds in (('connect4', N.CONNECT4_NET_ARGS, 2048, 100, 4.0, 0.4, 1500), ('connect4', N.CONNECT4_NET_ARGS, 1024, 100, 4.0, 0.4, 800),
                                                   ('connect4', N.CONNECT4_NET_ARGS, 256, 100, 4.0, 0.4, 1500),
                                                   ('brandubh', N.BRANDUBH_NET_ARGS, 512, 200, 1.25, 0.2, 1200), ('brandubh', N.BRANDUBH_NET_ARGS, 768, 200, 1.25, 0.2, 800),
                                                   ('brandubh', N.BRANDUBH_NET_ARGS, 1024, 200, 1.25, 0.2, 800), ('brandubh', N.BRANDUBH_NET_ARGS, 2048, 200, 1.25, 0.2, 800),
                                                   ('trimok', N.DEFAULT_NET_ARGS, 256, 50, 1.25, 0.2, 4000), ('trimok', N.DEFAULT_NET_ARGS, 1024, 50, 1.25, 0.2, 3000)):
    rounds = max(60, int(rounds * scale))
    Game = importlib.import_module('alphazero_general_amd.envs.' + game).Game
    torch.manual_seed(0)
    net = N.NNetWrapper(Game, netargs, device='cuda:0', dtype=torch.float16)
    gi = _abi.game_info(Game.AZG_GAME_ID)
    r = SelfPlayRunner(Game, net, args_for(sims, cpuct, fpu), num_slots=B, seed=0, example_capacity=B * 3 * (gi.max_turns + 1) * gi.num_symmetries)
    r.prepare()
    assert r.fused_search
    t0, total_games, total_sims, peak = time.time(), 0, 0, 0
    for i in range(rounds):
        r.play_round()
        if i % 50 == 49 or i + 1 == rounds:
            c = r.counters()                                     # raises on a sticky device error (tree store / example buffer / internal wait)
            assert c['num_examples'] <= r.engine.example_capacity
            peak = max(peak, c['max_nodes_used'])
            total_games += c['games_played']
            for ln in r.lanes:
                ln.engine.clear_outputs()
    c = r.counters()
    assert c['sims'] == B * sims * rounds, (c['sims'], B * sims * rounds)
    print('%s x %d (%s): %d rounds, %d games, %d simulations, %d expansions, peak nodes %d of %d, %.1f s' % (game, B, 'exact heads' if r.search_exact else 'sparse', rounds, total_games, c['sims'], c['expansions'], peak, r.engine.nodes_per_tree, time.time() - t0), flush=True)
    del r, net
    torch.cuda.empty_cache()

from alphazero_general_amd.envs.connect4 import Game
nets = []
for sd in (0, 1):
    torch.manual_seed(sd)
    n = N.NNetWrapper(Game, N.CONNECT4_NET_ARGS, device='cuda:0', dtype=torch.float16); n.refresh(); nets.append(n)
rounds = max(60, int(600 * scale))
r = ArenaRunner(Game, nets, args_for(100, 4.0, 0.4), num_slots=256, seed=0, result_capacity=64 * 256)
t0, games = time.time(), 0
for i in range(rounds):
    r.play_round()
    if i % 50 == 49 or i + 1 == rounds:
        c = r.engine.counters(); games += c['games_played']; r.engine.clear_outputs()
c = r.engine.counters()
assert c['sims'] == 256 * 100 * rounds
print('arena x 256 (persistent launch: %s): %d rounds, %d games, %d simulations, %.1f s' % (r.fused_search, rounds, games, c['sims'], time.time() - t0), flush=True)
