#!/bin/sh
# A/B/n of several builds of the library on the same box, alternating: tools/abn.sh <workload> <rounds> <lib> <lib> ...
W=$1; N=$2; shift; shift
for i in $(seq $N); do
  for L in "$@"; do
    AZG_LIB_PATH=$PWD/$L python bench.py --workload $W --no-sparse-heads --no-other-workloads --no-cpu-baseline $AZG_BENCH_EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], d['roofline'].get('launch_us_min_median_max'))"
  done
done
