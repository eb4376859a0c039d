"""Compile libazg_hip.so (the HIP engine + C ABI) for gfx950 with hipcc, in-tree.

    python -m alphazero_general_amd.build [--force] [--variant NAME]

-ffp-contract=off is REQUIRED: bit-exact visit counts depend on unfused float arithmetic.

Variants (never shipped as the product; AZG_LIB_PATH selects one):
    product        lib/libazg_hip.so      what every test, smoke() and bench.py load
    debug          lib/libazg_debug.so    -DAZG_DEBUG_BOUNDS: every node / child-block / path index the tree kernels form is checked against
                                          the store's capacity, the live allocation and the path length before it is used; a violation
                                          raises the sticky AZG_E_INTERNAL (tools/debug_soak.py runs the parity suite + a soak on it)
    tuning         lib/libazg_tuning.so   the product kernels + the AZG_TOWER_BOARDS / AZG_TOWER_PSPLIT / AZG_WIDE_TILE overrides
    timing-tree    lib/libazg_timing.so   s_memtime phase stamps of the tree kernels   (tools/time_tree.py)
    timing-tower   lib/libazg_timing.so   per-layer / per-phase stamps of k_tower2     (tools/tower_stamps.py, tools/wide_search_phases.py)
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
VARIANTS = {                                                         # name -> (output file, extra defines)
    'product': ('libazg_hip.so', []),
    'debug': ('libazg_debug.so', ['-DAZG_DEBUG_BOUNDS']),
    'tuning': ('libazg_tuning.so', ['-DAZG_TUNING']),
    'timing-tree': ('libazg_timing.so', ['-DAZG_TREE_TIMING']),
    'timing-tower': ('libazg_timing.so', ['-DAZG_TOWER_TIMING']),
    'timing-w1': ('libazg_timing_w1.so', ['-DAZG_TOWER_TIMING', '-DAZG_PHASE_TID=64', '-DAZG_TUNING']),          # phase stamps of wavefront 1 (a streaming wavefront in every heads layout)
    'timing-w1-overlap': ('libazg_timing_w1o.so', ['-DAZG_TOWER_TIMING', '-DAZG_PHASE_TID=64', '-DAZG_OVL_NW=2', '-DHEADS_A_RING=1', '-DAZG_TUNING']),
    'ring1': ('libazg_ring1.so', ['-DHEADS_A_RING=1', '-DAZG_TUNING']),                        # A/B of the heads' A-operand ring depth (product: 4); ring 1 = round 5's schedule
    # experiment (not adopted): the one-game exact tile with the walk overlapped with the policy-head stream.  A/B against 'ring1': the
    # 19-subtile unrolled stream of the two streaming wavefronts only compiles sanely with the A-operand ring at depth 1 (depth 4: 22 k spilled SGPRs)
    'overlap': ('libazg_overlap.so', ['-DAZG_OVL_NW=2', '-DHEADS_A_RING=1', '-DAZG_TUNING']),
    'headline1': ('libazg_headline1.so', ['-DAZG_HEADLINE_ONE_WG']),   # experiment: the connect4 search kernel with one workgroup per CU (512 registers, no spills)
}


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def source_sha(variant='product'):
    """Content hash of EVERYTHING the binary is made from: csrc/*.h (the device code), csrc/*.hip (the host side: tile choice, launch
    bounds, LDS sizing, the cost model), include/azg.h, and the compile flags incl. the variant's defines.  Compiled into the library
    (azg_source_sha) so that a run can say which sources the LOADED binary was built from; bench.csrc_sha is this function over the
    working tree, and committed counters (profiles/*_pmc.json, *_phase_budget.json) are quoted only when all three agree."""
    import hashlib
    h = hashlib.sha256()
    for f in DEPS:
        if f.endswith(('.h', '.hip')):
            h.update(os.path.basename(f).encode() + b'\0' + open(f, 'rb').read())
    h.update(' '.join(FLAGS + VARIANTS[variant][1]).encode())
    return h.hexdigest()[:16]


def out_path(variant='product'):
    return os.path.join(HERE, 'lib', VARIANTS[variant][0])


def needs_build(variant='product'):
    out = out_path(variant)
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, variant='product'):
    out = out_path(variant)
    if not force and not needs_build(variant):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc()] + FLAGS + VARIANTS[variant][1] + ['-DAZG_SRC_SHA="%s"' % source_sha(variant), '-o', out, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == '__main__':
    v = sys.argv[sys.argv.index('--variant') + 1] if '--variant' in sys.argv else 'product'
    print(build(force='--force' in sys.argv, verbose=True, variant=v))
