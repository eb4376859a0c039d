cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cs in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"; do
  tag=$(echo $cs | cut -d' ' -f1)
  rocprofv3 --pmc $cs --output-format csv -d $R/gpurun_out/lds2048/$tag -- python $R/bench.py --workload brandubh --slots 2048 --no-cpu-baseline --no-library-gemm --no-other-workloads --no-sparse-heads --steps 4 --warmup 1 > $R/gpurun_out/lds2048/$tag.out 2> $R/gpurun_out/lds2048/$tag.err
  echo $tag rc $?
done
