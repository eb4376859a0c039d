"""compat mode (unmodified-Coach protocol, bench.compat_run) over agent counts and games per agent: tools/compat_sweep.py [seconds]
prints one JSON line per (workers, games per worker)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from alphazero_general_amd import nnet as nn_mod
from alphazero_general_amd.nnet import NNetWrapper
from alphazero_general_amd.envs.connect4 import Game

if __name__ == '__main__':
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    torch.manual_seed(0)
    net = NNetWrapper(Game, nn_mod.CONNECT4_NET_ARGS, device='cuda:0', dtype=torch.float16)
    net.refresh()
    for workers, gpw in [tuple(int(x) for x in a.split('x')) for a in sys.argv[2:]] or ((2, 1024), (4, 512), (4, 1024), (8, 512), (2, 2048), (4, 2048), (8, 1024)):
        r = bench.compat_run(bench.WORKLOADS['connect4'], net, seconds=secs, workers=workers, games_per_worker=gpw)
        print(json.dumps({k: r.get(k) for k in ('workers', 'games_per_worker', 'value', 'ms_per_batch', 'parent_us_per_batch', 'error')}), flush=True)
