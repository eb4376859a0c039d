"""Is the tile the persistent wide-head launch picks (measured at set-up / device-derived model) the fastest one?

    python tools/tile_pick_check.py [--game brandubh] [--sizes 512,768,1024,2048] [--depths 4,8] [--sims 40] [--out FILE]

One child process per library: the PRODUCT (lib/libazg_hip.so: picks by itself) and the TUNING build (lib/libazg_tuning.so, built by
alphazero_general_amd.build --variant tuning) with AZG_WIDE_BOARDS = 1 .. 4 forcing every tile shape.  Each child times the exact
persistent launch (min over 7 launches of `sims` simulations after 3 warm ones) for every (depth, engine size) and prints JSON; the parent
prints / writes the table and, per cell, pick_over_best = time(product's pick) / min(time over forced tiles).
(tests/test_gpu_tiles.py asserts pick_over_best <= 1.05 for depths 4 and 8.)"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import importlib
    import torch
    from alphazero_general_amd import nnet as N
    from alphazero_general_amd.engine import DeviceEngine
    from alphazero_general_amd.utils import dotdict
    Game = importlib.import_module('alphazero_general_amd.envs.' + a.game).Game
    base = N.BRANDUBH_NET_ARGS if a.game == 'brandubh' else N.DEFAULT_NET_ARGS
    out = {}
    for depth in a.depths:
        na = dotdict(dict(base)); na['depth'] = depth
        torch.manual_seed(31)
        net = N.NNetWrapper(Game, na, device='cuda:0', dtype=torch.float16)
        net.refresh()
        hip = net._hip
        for B in a.sizes:
            e = DeviceEngine(Game.AZG_GAME_ID, B, cpuct=1.25, fpu_reduction=0.2, add_root_noise=True, add_root_temp=True, seed=6, games_per_iteration=1 << 30,
                             example_capacity=0, sims_hint=a.sims, nodes_per_tree=(2 * a.sims + 2) * e_maxk(Game) + 64)
            rec = {'us': None, 'tile': None}
            try:
                hip.search(e, 0, exact=True)                         # one-time set-up (the product measures its tiles here)
                rec['tile'] = hip.search_tile(e, exact=True)
                best = None
                for i in range(10):
                    e.reset()
                    torch.cuda.synchronize()
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                    hip.search(e, a.sims, exact=True)
                    t1.record(); torch.cuda.synchronize()
                    if i >= 3:
                        us = t0.elapsed_time(t1) * 1e3
                        best = us if best is None else min(best, us)
                e.counters()                                          # (a sticky device error would raise here)
                rec['us'] = round(best, 1)
            except Exception as ex:                                  # noqa: BLE001 (a forced tile this depth does not fit)
                rec['error'] = '%s: %s' % (type(ex).__name__, str(ex)[:120])
            out['%d/%d' % (depth, B)] = rec
            e.close()
    print('TILEJSON ' + json.dumps(out))


def e_maxk(Game):
    from alphazero_general_amd import _abi
    return _abi.game_info(Game.AZG_GAME_ID).max_children


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--game', default='brandubh')
    ap.add_argument('--sizes', default='512,768,1024,2048')
    ap.add_argument('--depths', default='4,8')
    ap.add_argument('--sims', type=int, default=40)
    ap.add_argument('--out', default=None)
    ap.add_argument('--child', action='store_true')
    a = ap.parse_args()
    a.sizes = [int(x) for x in a.sizes.split(',')]
    a.depths = [int(x) for x in a.depths.split(',')]
    if a.child:
        return child(a)
    from alphazero_general_amd import build
    tuning = build.build(variant='tuning')
    tiles = 4 if a.game == 'brandubh' else 2
    runs = {}
    for name, env in [('pick', {})] + [('forced_%d' % t, {'AZG_LIB_PATH': tuning, 'AZG_WIDE_BOARDS': str(t)}) for t in range(1, tiles + 1)]:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--game', a.game, '--sizes', ','.join(map(str, a.sizes)),
                            '--depths', ','.join(map(str, a.depths)), '--sims', str(a.sims)], env=dict(os.environ, **env), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=1200)
        line = [l for l in r.stdout.decode(errors='replace').splitlines() if l.startswith('TILEJSON ')]
        if r.returncode != 0 or not line:
            raise SystemExit('%s failed:\n%s' % (name, r.stdout.decode(errors='replace')[-2000:]))
        runs[name] = json.loads(line[0][9:])
    table = {}
    for key, rec in runs['pick'].items():
        forced = {t: runs['forced_%d' % t][key]['us'] for t in range(1, tiles + 1)}
        ok = [v for v in forced.values() if v]
        table[key] = {'pick': rec['tile'], 'pick_us': rec['us'], 'forced_us': forced, 'pick_over_best': round(rec['us'] / min(ok), 4) if ok and rec['us'] else None}
    res = {'game': a.game, 'sims_per_launch': a.sims, 'cells': table}
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, 'w').write(txt + '\n')
    return 0


if __name__ == '__main__':
    sys.exit(main() or 0)
