"""Phase budget of the persistent wide-head search launches (brandubh, 3-player env) -- run ON the GPU box:

    sh tools/build_timing.sh tower && python tools/phase_budget.py --git <short hash>

Runs tools/wide_search_phases.py under the measurement build (libazg_timing.so: s_memtime stamps around the tree phase, the tower and
the head convolutions of every simulation) and writes gpurun_out/r06_phase_budget.json -- cycles per simulation and phase, mean over
the workgroups of the last launch -- stamped with the hash of the kernel sources (bench.csrc_sha), which bench.py's `phase_budget`
block reads from profiles/r06_phase_budget.json."""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--git', default='?')
    a = ap.parse_args()
    import bench
    lib = os.path.join(ROOT, 'alphazero_general_amd', 'lib', 'libazg_timing.so')
    assert os.path.exists(lib), 'build the measurement library first: sh tools/build_timing.sh tower'
    out = {'git': a.git, 'csrc_sha': bench.csrc_sha(), 'unit': 'shader cycles per simulation (s_memtime), mean over the workgroups of the last launch',
           'workloads': {}}
    for game, B, key in (('brandubh', 512, 'brandubh'), ('brandubh', 2048, 'brandubh_2048'), ('trimok', 256, 'trimok'), ('trimok', 1024, 'trimok_1024')):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'wide_search_phases.py'), str(B), game, 'exact'], env=dict(os.environ, AZG_LIB_PATH=lib),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        err = r.stderr.decode(errors='replace')
        ms = re.findall(r'wide search, cycles per simulation \(mean over (\d+) workgroups\): tree (\d+) tower (\d+) headconv (\d+) heads (\d+)', err)
        hs = re.findall(r'helper wavefront .*?: header (\d+) masks (\d+) logits (\d+) softmax (\d+) priors (\d+)', err)
        assert ms, err[-2000:]
        n, tree, tower, hc, heads = [int(x) for x in ms[-1]]
        tl = re.findall(r'TILE (\d+)', r.stdout.decode(errors='replace'))
        assert tl, r.stdout.decode(errors='replace')[-500:]
        gpw = int(tl[-1])                                            # the tile the launch picked for this engine size (measured at set-up)
        rec = {'games': B, 'games_per_workgroup': gpw, 'search_heads': 'exact', 'workgroups_sampled': n, 'tree': tree, 'tower': tower, 'headconv': hc, 'heads': heads}
        if hs:
            rec['helper_wavefront'] = dict(zip(('header', 'masks', 'logits', 'softmax', 'priors'), [int(x) for x in hs[-1]]))
        out['workloads'][key] = rec
        print(key, rec)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'r06_phase_budget.json'), 'w') as fh:
        json.dump(out, fh, indent=1)


if __name__ == '__main__':
    main()
