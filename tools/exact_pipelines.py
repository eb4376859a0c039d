"""launch-per-phase forms of the wide-head workloads at a shard size, with the slots split into `pipelines` stream lanes (tree launch
of one lane under the tower of another): tools/exact_pipelines.py <workload> <slots> <heads: logits|features> <pipelines...>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
import numpy as np
import torch
import bench
from alphazero_general_amd import nnet as nn_mod
from alphazero_general_amd.nnet import NNetWrapper
from alphazero_general_amd.selfplay import SelfPlayRunner

name, B, heads = sys.argv[1], int(sys.argv[2]), sys.argv[3]
W = dict(bench.WORKLOADS[name]); W['B'] = B
Game = importlib.import_module('alphazero_general_amd.envs.' + W['game']).Game
torch.manual_seed(0)
net = NNetWrapper(Game, getattr(nn_mod, W['net']), device='cuda:0', dtype=torch.float16)
for p in [int(x) for x in sys.argv[4:]]:
    r = SelfPlayRunner(Game, net, bench.selfplay_args(W), num_slots=B, seed=0, device=0, pipelines=p, heads=heads,
                       example_capacity=int(B * 4) * (Game.max_turns() + 1) * 8)
    r.prepare()
    for _ in range(2):
        r.play_round()
    torch.cuda.synchronize(); c0 = r.counters(); t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        r.play_round()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0; c1 = r.counters()
    print('%s %d games heads=%s pipelines=%d: %.1f M expansions/s, %.3f ms/move, %.1f us per simulation step' % (
        name, B, heads, p, (c1['expansions'] - c0['expansions']) / dt / 1e6, dt / n * 1e3, dt / n / W['sims'] * 1e6), flush=True)
    for ln in r.lanes:
        ln.engine.close()
    del r
    torch.cuda.empty_cache()
