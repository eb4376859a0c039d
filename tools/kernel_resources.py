"""Register / spill / occupancy table of every kernel in the library (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os
import re
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ''
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-Rpass-analysis=kernel-resource-usage',
       '-o', '/tmp/azg_res.so', os.path.join(ROOT, 'alphazero_general_amd', 'csrc', 'azg_engine.hip')] + sys.argv[2:]
err = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode(errors='replace')
cur, rows = None, []
for line in err.splitlines():
    m = re.search(r'remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass', line) or re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        name = t.split(':', 1)[1].strip()
        try:
            name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], stdout=subprocess.PIPE).stdout.decode().strip()
        except Exception:
            pass
        cur = {'name': name.replace('azg::', '').split('(')[0]}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
for r in rows:
    if flt in r['name']:
        print('%-78s VGPR %4s AGPR %3s spill %4s sgpr-spill %3s scratch %5s occ %s LDS %s' % (
            r['name'][:78], r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'), r.get('SGPRs Spill'), r.get('ScratchSize [bytes/lane]'),
            r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
