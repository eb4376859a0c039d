"""Once per round: the parity suite and a soak on the BOUNDS-CHECKED build of the library (alphazero_general_amd/build.py --variant debug,
-DAZG_DEBUG_BOUNDS: every node / child-block / path index of the tree kernels is checked against the store's capacity, the tree's live
allocation and the path length before it is used; the reference compiles its own checks out, MCTS.pyx:2-6).

    python tools/debug_soak.py [--scale 0.05] [--out gpurun_out/debug_bounds.txt]

1. positive control: a root whose child block points outside the live allocation IS caught (sticky AZG_E_INTERNAL, site 2) and not read;
2. the parity suites against the oracle / the goldens, run with AZG_LIB_PATH = the debug library;
3. tools/soak.py (every tile shape of the persistent launches, node stores compacting) on the debug library;
the log ends with the verdict: zero bounds reports or the first failing site."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTROL = r'''
import ctypes as C, sys, torch
sys.path.insert(0, %r)
from alphazero_general_amd import _abi
from alphazero_general_amd.engine import DeviceEngine
e = DeviceEngine(0, 8, seed=1, sims_hint=8, example_capacity=64)
assert e.bounds_site() == (0, True), e.bounds_site()
obs = e.new_obs(torch.float16)
p = torch.full((8, 7), 1 / 7, device='cuda'); v = torch.full((8, 3), 1 / 3, device='cuda')
for _ in range(4):
    e.select(obs); e.backup(p, v)
e.counters()
L = _abi.lib()
L.azg_debug_poke_root_fc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int32]
assert L.azg_debug_poke_root_fc(e.h, None, 3, 1 << 20) == 0          # slot 3: the root's child block now lies far outside the store
e.select(obs)
try:
    e.counters()
    print('CONTROL FAILED: no error raised')
except _abi.AzgError as ex:
    site, chk = e.bounds_site()
    print('CONTROL OK: %%s, first failing site %%d' %% (ex, site))
    assert site == 2
'''


def run(cmd, env, log, timeout):
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = r.stdout.decode(errors='replace')
    log.write('$ %s\n%s\n[exit %d, %.0f s]\n\n' % (' '.join(cmd), out[-6000:], r.returncode, time.time() - t0))
    log.flush()
    return r.returncode, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scale', type=float, default=0.05)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'debug_bounds.txt'))
    ap.add_argument('--suites', default='tests/test_gpu_parity.py,tests/test_gpu_rules.py,tests/test_gpu_runner_oracle.py,tests/test_gpu_fullsize.py,tests/test_gpu_runners.py')
    a = ap.parse_args()
    from alphazero_general_amd import build
    lib = build.build(variant='debug')
    env = dict(os.environ, AZG_LIB_PATH=lib)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    ok = True
    with open(a.out, 'w') as log:
        log.write('bounds-checked build: %s (source stamp %s)\n\n' % (os.path.relpath(lib, ROOT), build.source_sha('debug')))
        rc, out = run([sys.executable, '-c', CONTROL % ROOT], env, log, 600)
        ok &= rc == 0 and 'CONTROL OK' in out
        rc, out = run([sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '-x', '-k', 'not two_ranks and not bench_'] + a.suites.split(','), env, log, 3000)
        ok &= rc == 0
        rc, out = run([sys.executable, os.path.join(ROOT, 'tools', 'soak.py'), str(a.scale)], env, log, 3000)
        ok &= rc == 0
        log.write('VERDICT: %s\n' % ('positive control caught; parity suites green and soak clean on the bounds-checked build: zero bounds reports' if ok
                                     else 'FAILED -- see above'))
    print(open(a.out).read()[-3000:])
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
