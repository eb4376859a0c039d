#!/bin/sh
# the 1 / 2 / 4-GPU shard sizes of BASELINE configs 3-5 on ONE GPU (config 3: 4096 games total -> 4096 / 2048 / 1024 / 512 per GPU,
# config 5: 1024 -> 1024 / 512 / 256, config 4: 512 -> 512 / 256): tools/shard_sweep.sh <outdir> [extra bench args]
O=${1:-gpurun_out/shards}; shift
mkdir -p $O
for s in 512 1024 2048; do python bench.py --workload brandubh --slots $s --steps 6 --warmup 2 --no-cpu-baseline --no-library-gemm "$@" > $O/brandubh_$s.json 2> $O/brandubh_$s.err; done
python bench.py --workload brandubh --slots 4096 --steps 4 --warmup 1 --no-cpu-baseline --no-library-gemm --no-sparse-heads "$@" > $O/brandubh_4096.json 2> $O/brandubh_4096.err
for s in 256 512 1024; do python bench.py --workload trimok --slots $s --steps 8 --warmup 2 --no-cpu-baseline --no-library-gemm "$@" > $O/trimok_$s.json 2> $O/trimok_$s.err; done
for s in 256 512; do python bench.py --workload arena --slots $s --steps 8 --warmup 2 --no-cpu-baseline --no-library-gemm "$@" > $O/arena_$s.json 2> $O/arena_$s.err; done
for f in $O/*.json; do python - "$f" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print(sys.argv[1], d['config']['games_per_gpu'], 'value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r.get('frac'), 'heads', d['config'].get('search_heads'), 'sparse', (d.get('sparse_heads') or {}).get('value'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
