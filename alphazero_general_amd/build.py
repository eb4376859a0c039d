"""Compile libazg_hip.so (the HIP engine + C ABI) for gfx950 with hipcc, in-tree.

    python -m alphazero_general_amd.build

-ffp-contract=off is REQUIRED: bit-exact visit counts depend on unfused float arithmetic.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'azg_engine.hip')
DEPS = [os.path.join(HERE, 'csrc', f) for f in sorted(os.listdir(os.path.join(HERE, 'csrc')))] + \
       [os.path.join(os.path.dirname(HERE), 'include', 'azg.h')]
OUT = os.path.join(HERE, 'lib', 'libazg_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared']


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def source_sha():
    """content hash of the kernel sources (csrc/*.h, the device code) -- compiled into the library (azg_source_sha) so that a run can say
    which sources the loaded binary was built from; bench.csrc_sha computes the same over the working tree"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, 'csrc', '*.h'))):
        h.update(os.path.basename(f).encode() + b'\0' + open(f, 'rb').read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc()] + FLAGS + ['-DAZG_SRC_SHA="%s"' % source_sha(), '-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
