#!/bin/sh
# tile shapes of the persistent wide-head search launch at a shard size (tuning build: AZG_WIDE_BOARDS): tools/wide_tiles.sh <workload> <slots> <boards...>
W=$1; S=$2; shift; shift
for b in "$@"; do
  AZG_LIB_PATH=$PWD/alphazero_general_amd/lib/libazg_tuning.so AZG_WIDE_BOARDS=$b python bench.py --workload $W --slots $S --steps 6 --warmup 2 --no-cpu-baseline --no-library-gemm --no-sparse-heads --profile-rounds 1 $AZG_BENCH_EXTRA 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W slots $S boards/tile $b:', d['value'], 'exp/s', d['ms_per_step'], 'ms/step frac', d['roofline']['frac'])"
done
