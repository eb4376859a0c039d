"""Collect the rocprofv3 evidence bench.py and DESIGN.md cite -- run ON the GPU box:

    python tools/collect_profiles.py --git <short hash of the commit being profiled> [--workloads connect4 brandubh arena trimok]

Per workload: (1) `rocprofv3 --kernel-trace --stats` of the bench command -> profiles/<round>_<workload>_kernel_stats.csv and the bench
line printed under the profiler; (2) PMC passes of the same command, one counter set per pass as MI355X_MICROARCH.md prescribes
(FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with --stats / trace domains other than the kernel trace):
(profiles/ does not travel back from the GPU box, gpurun_out/ does: re-run with --reuse in the build container to rebuild the
profiles/<round>_* files from the raw output)  per-kernel averages -> profiles/<round>_pmc_summary.csv, and HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; the
guide's gfx950 correction: FETCH_SIZE reports half the bytes of wide loads) -> profiles/<round>_pmc.json, which bench.py reads for its
`traffic` fields.  The connect4 run adds an MFMA / LDS / clock pass for the search launch."""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')
ROUND = 'r06'
SCRATCH = os.path.join(ROOT, 'gpurun_out', ROUND + '_prof')
KEEP = ('k_tower2', 'k_backup_select2', 'k_heads', 'k_select', 'k_backup', 'k_play', 'k_compact', 'k_emit', 'k_finalize', 'k_arena_rows')


REUSE = False


def rocprof(tag, extra, bench_args, timeout=600):
    out = os.path.join(SCRATCH, tag)
    saved = os.path.join(out, 'bench_line.json')
    if REUSE:                                                   # post-process the raw output of an earlier run (no GPU needed)
        line = open(saved).read().strip() if os.path.exists(saved) else None
        return out, line, 0 if os.path.isdir(out) else -1
    shutil.rmtree(out, ignore_errors=True)
    env = dict(os.environ, TMPDIR='/tmp')
    cmd = ['rocprofv3'] + extra + ['--output-format', 'csv', '-d', out, '--', sys.executable, os.path.join(ROOT, 'bench.py')] + bench_args
    r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    line = [l for l in r.stdout.decode(errors='replace').splitlines() if l.startswith('{"metric"')]
    if line and os.path.isdir(out):
        with open(saved, 'w') as fh:
            fh.write(line[0] + '\n')
    return out, (line[0] if line else None), r.returncode


_DEMANGLED = {}


def short(name):
    """a readable kernel label: demangled template name up to the argument list"""
    if name.startswith('_Z'):
        if name not in _DEMANGLED:
            try:
                _DEMANGLED[name] = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], stdout=subprocess.PIPE, timeout=20).stdout.decode().strip() or name
            except Exception:
                _DEMANGLED[name] = name
        name = _DEMANGLED[name]
    n = name.replace('void ', '').replace('azg::', '')
    return n.split('(')[0][:120]


def pmc_averages(outdir):
    acc = {}
    # (gpurun merges a call's output INTO the local directory: files of earlier collections -- their names carry the profiler's process
    #  id -- stay next to the new ones; only the newest run counts)
    files = glob.glob(os.path.join(outdir, '**', '*counter_collection.csv'), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row['Kernel_Name'])
                if not any(s in k for s in KEEP):
                    continue
                key = (k, row['Counter_Name'])
                a = acc.setdefault(key, [0.0, 0])
                a[0] += float(row['Counter_Value']); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--git', default=os.environ.get('AZG_GIT', 'unknown'))
    ap.add_argument('--workloads', nargs='*', default=['connect4', 'brandubh', 'brandubh@2048', 'arena', 'trimok', 'trimok@1024'])
    ap.add_argument('--reuse', action='store_true', help='rebuild profiles/<round>_* from the raw output already under gpurun_out/<round>_prof')
    a = ap.parse_args()
    global REUSE
    REUSE = a.reuse
    os.makedirs(PROF, exist_ok=True); os.makedirs(SCRATCH, exist_ok=True)
    sys.path.insert(0, ROOT)
    import bench
    summary_rows, pmc = [], {'git': a.git, 'csrc_sha': bench.csrc_sha(), 'unit': 'bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024', 'workloads': {}}
    for w in a.workloads:
        # (name or name@slots: a shard size of the workload, e.g. brandubh@2048 -- config 3's 2-GPU shard, four games per workgroup)
        wname, _, wslots = w.partition('@')
        base = ['--workload', wname, '--no-cpu-baseline', '--no-library-gemm', '--no-other-workloads', '--no-sparse-heads'] + (['--slots', wslots] if wslots else [])
        w = w.replace('@', '_')
        out, line, rc = rocprof('kt_' + w, ['--kernel-trace', '--stats'], base + ['--steps', '12', '--warmup', '2'])
        stats = glob.glob(os.path.join(out, '**', '*kernel_stats.csv'), recursive=True)
        if stats:
            shutil.copy(max(stats, key=os.path.getmtime), os.path.join(PROF, ROUND + '_%s_kernel_stats.csv' % w))
        if line:
            with open(os.path.join(PROF, ROUND + '_%s_bench_line_under_rocprof.json' % w), 'w') as fh:
                fh.write(line + '\n')
        print(w, 'kernel trace rc', rc, 'stats' if stats else 'NO STATS', flush=True)
        passes = [['FETCH_SIZE'], ['WRITE_SIZE']]
        if w == 'connect4':
            passes += [['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_ANY',
                        'GRBM_GUI_ACTIVE'], ['TCC_HIT_sum', 'TCC_MISS_sum']]
        else:
            passes += [['SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE']]
        per_kernel = {}
        for cs in passes:
            out, _, rc = rocprof('pmc_%s_%s' % (w, cs[0]), ['--pmc'] + cs, base + ['--steps', '4', '--warmup', '1'])
            av = pmc_averages(out)
            print(w, cs, 'rc', rc, len(av), 'kernel-counter pairs', flush=True)
            for (k, c), (v, n) in sorted(av.items()):
                summary_rows.append([w, k, c, n, '%.3f' % v])
                per_kernel.setdefault(k, {})[c] = (v, n)
        rec = {}
        for k, cs in per_kernel.items():
            if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
                f, wv = cs['FETCH_SIZE'][0], cs['WRITE_SIZE'][0]
                rec[k] = {'fetch_kb': round(f, 2), 'write_kb': round(wv, 2), 'traffic_bytes': int((2 * f + wv) * 1024),
                          'dispatches': int(cs['FETCH_SIZE'][1])}
                if cs.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 0))[0] > 0:   # cycles the MFMA pipes were busy, summed over the chip, per launch
                    rec[k]['mfma_busy_cycles'] = round(cs['SQ_VALU_MFMA_BUSY_CYCLES'][0], 1)
        pmc['workloads'][w] = rec
    with open(os.path.join(PROF, ROUND + '_pmc.json'), 'w') as fh:
        json.dump(pmc, fh, indent=1, sort_keys=True)
    with open(os.path.join(PROF, ROUND + '_pmc_summary.csv'), 'w', newline='') as fh:
        wr = csv.writer(fh)
        wr.writerow(['workload', 'kernel', 'counter', 'dispatches', 'avg_per_dispatch'])
        wr.writerows(summary_rows)
    print('wrote profiles/%s_pmc.json, profiles/%s_pmc_summary.csv' % (ROUND, ROUND))


if __name__ == '__main__':
    main()
