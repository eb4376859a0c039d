#!/bin/sh
# A/B of two builds of the library on the same box, alternating: tools/ab.sh <workload> <libA> <libB> [rounds]
W=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for L in $A $B; do
    AZG_LIB_PATH=$PWD/$L python bench.py --workload $W --no-sparse-heads --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'])"
  done
done
