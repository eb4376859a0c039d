"""Rig for tests/test_gpu_iteration.py: started under torch.distributed.run with two ranks that share GPU 0 (AZG_SINGLE_DEVICE, gloo --
RCCL refuses two ranks on one device).  Modes:
  direct  every rank calls iteration.run_iteration / run_arena itself (the collective form)
  coach   rank 0 is where a Coach would live: it hands its LIVE nets to iteration.lead; rank 1 sits in iteration.serve
Rank 0 writes a JSON record (argv[2]) and, for self-play, the three iteration files under argv[3]."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from alphazero_general_amd import distributed as D  # noqa: E402
from alphazero_general_amd import iteration as I  # noqa: E402
from alphazero_general_amd.envs.connect4 import Game  # noqa: E402
from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper  # noqa: E402
from alphazero_general_amd.utils import dotdict, default_temp_scaling  # noqa: E402

B, SIMS, GAMES, ARENA_GAMES, ITER, SEED = 48, 16, 60, 40, 5, 9


def args(**kw):
    a = dotdict(numMCTSSims=SIMS, numFastSims=4, probFastSim=0.0, gamesPerIteration=GAMES, cpuct=4.0, fpu_reduction=0.4, root_noise_frac=0.3,
                root_policy_temp=1.3, min_discount=1.0, add_root_noise=True, add_root_temp=True, symmetricSamples=True, mctsResetThreshold=0,
                startTemp=1.0, arenaTemp=0.25, temp_scaling_fn=default_temp_scaling, use_draws_for_winrate=True, _azg_seed=SEED)
    a.update(kw)
    return a


def net(seed, dev='cuda:0'):
    torch.manual_seed(seed)
    return NNetWrapper(Game, CONNECT4_NET_ARGS, device=dev, dtype=torch.float16)


def digest(samples):
    """order-free digest of a sample set: sha of the sorted per-sample row hashes"""
    o, p, z = [t.cpu().numpy() for t in samples]
    rows = sorted(hashlib.sha1(o[i].tobytes() + p[i].tobytes() + z[i].tobytes()).hexdigest() for i in range(o.shape[0]))
    return hashlib.sha1(''.join(rows).encode()).hexdigest(), len(rows)


def main():
    mode, out_json, folder = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, local_rank, world = D.init_from_env()
    torch.cuda.set_device(local_rank)
    rec = {}
    if mode == 'direct':
        n = net(1)
        r = I.run_iteration(Game, n, args(), ITER, folder, num_slots=B)
        rec['quota'] = {k: r[k] for k in ('wins', 'draws', 'avg_game_length', 'num_results', 'games', 'num_samples', 'ranks', 'sims', 'expansions')}
        rec['quota']['digest'] = digest(r['samples'])
        r2 = I.run_iteration(Game, n, args(gamesPerIteration=1 << 30), ITER + 1, None, num_slots=B, max_rounds=14)
        rec['rounds'] = {k: r2[k] for k in ('games', 'num_samples', 'sims', 'num_results')}
        rec['rounds']['digest'] = digest(r2['samples'])
        rec['arena'] = I.run_arena(Game, [n, net(2)], args(), ARENA_GAMES, num_slots=16, details=True)
    else:
        if rank == 0:
            live = [net(1), net(2)]                                  # the Coach's live nets (train_net / self_play_net)
            r = I.lead('selfplay', Game, [live[0]], args(), iteration=ITER, folder=folder, num_slots=B)
            rec['quota'] = {k: r[k] for k in ('wins', 'draws', 'avg_game_length', 'num_results', 'games', 'num_samples', 'ranks', 'sims', 'expansions')}
            rec['quota']['digest'] = digest(r['samples'])
            rec['arena'] = I.lead('arena', Game, live, args(), num_games=ARENA_GAMES, num_slots=16, details=True)
            rec['warmup'] = {k: v for k, v in I.lead('selfplay', Game, [None], args(gamesPerIteration=20, numWarmupSims=5), iteration=1, folder=None,
                                                     num_slots=16, warmup=True, keep_samples=False).items() if k in ('games', 'num_samples', 'ranks')}
            I.lead('stop', Game, [], args())
        else:
            torch.manual_seed(12345)                                 # (nothing this rank could build by itself equals rank 0's nets)
            rec['served'] = I.serve(Game)
    for k in ('arena',):
        if k in rec:
            rec[k].pop('seconds', None)
    if rank == 0:
        with open(out_json, 'w') as f:
            json.dump(rec, f)
    elif 'served' in rec:
        with open(out_json + '.rank1', 'w') as f:
            json.dump(rec, f)
    D.barrier()
    D.shutdown()


if __name__ == '__main__':
    main()
