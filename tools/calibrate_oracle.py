"""BASELINE.md section 3 calibration: how fast is the C oracle (what bench.py's cpu_baseline runs on the GPU box) relative to the
REFERENCE's own Cython path?  Runs in the BUILD CONTAINER only (it imports /root/reference through tests/golden/refharness.py);
writes profiles/cpu_oracle_vs_reference.json, which bench.py attaches to every cpu_baseline.

    PYTHONDONTWRITEBYTECODE=1 python tools/calibrate_oracle.py

Both sides: one host core, tree only (the reference agent in warm-up mode: uniform policy / value, SelfPlayAgent.pyx:48-52,
111-114; the oracle with the same evaluator), same games x sims, np.random left alone on the reference side (its own legacy
stream: a shuffle / choice there costs what it costs in the reference)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import refharness as rh  # noqa: E402
from refharness import ol  # noqa: E402

CONFIGS = [('connect4', 0, 'alphazero.envs.connect4.connect4', 256, 100, dict(cpuct=4.0, fpu_reduction=0.4)),
           ('brandubh', 1, 'alphazero.envs.brandubh.fastafl', 64, 50, dict(cpuct=1.25, fpu_reduction=0.2))]


def time_reference(modname, gid, B, sims, kw, seconds):
    import importlib
    import torch
    torch.set_num_threads(1)
    Game = importlib.import_module(modname).Game
    if not hasattr(Game, 'max_turns') or gid == 1:                       # the brandubh snapshot lacks these two
        class Game(Game):                                               # noqa: F811
            @staticmethod
            def max_turns():
                return 100

            @staticmethod
            def has_draw():
                return True
    args = rh.ref_args(Game, numMCTSSims=sims, numWarmupSims=sims, gamesPerIteration=1 << 30, add_root_noise=True, add_root_temp=True, **kw)
    tape = rh.Tape(0)                                                   # (not installed: the reference keeps its own np.random)
    ag = rh.make_ref_agent(Game, gid, B, args, tape, is_warmup=True)
    np.random.seed(0)
    done, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        for _ in range(sims):
            ag.generateBatch(); ag.processBatch()
        ag.playMoves()
        done += B * sims
    return done / (time.time() - t0)


def time_oracle(gid, B, sims, kw, seconds):
    pool = ol.OPool(gid, 1, B, sims=sims, games_per_iteration=1 << 30, seed=0, add_root_noise=True, add_root_temp=True, **kw)
    dt = pool.run_tree_only(seconds)
    return pool.sims_done / dt


def main():
    rh.import_reference()
    out = {'where': 'build container, one core for both sides, tree only (uniform evaluator)', 'configs': {}}
    try:
        out['git'] = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT).decode().strip()
        out['cpu'] = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
    except Exception:
        pass
    for name, gid, mod, B, sims, kw in CONFIGS:
        ref = time_reference(mod, gid, B, sims, kw, 8.0)
        ora = time_oracle(gid, B, sims, kw, 4.0)
        out['configs'][name] = {'games': B, 'sims_per_move': sims, 'reference_cython_sims_per_s': round(ref, 1),
                                'c_oracle_sims_per_s': round(ora, 1), 'oracle_over_reference': round(ora / ref, 2)}
        print(name, out['configs'][name], flush=True)
    with open(os.path.join(ROOT, 'profiles', 'cpu_oracle_vs_reference.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
