cd /tmp && export TMPDIR=/tmp
for cfg in "brandubh 2048" "brandubh 512"; do set -- $cfg
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/tcc_$2 -- python $GRAFT_REPO_ROOT/bench.py --workload $1 --slots $2 --steps 3 --warmup 1 --no-cpu-baseline --no-library-gemm --no-other-workloads --no-sparse-heads --profile-rounds 1 > /dev/null 2>&1
python - $2 <<'PY'
import csv,glob,sys
acc={}
for f in glob.glob('/tmp/tcc_%s/**/*counter_collection.csv'%sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'SearchWide' not in k and 'k_tower2' not in k: continue
        a=acc.setdefault((k[:60], r['Counter_Name']), [0,0]); a[0]+=float(r['Counter_Value']); a[1]+=1
for (k,c),(v,n) in sorted(acc.items()): print(sys.argv[1], k, c, n, v/n)
PY
done
