# SQ counters of the verification kernel (one pass; --kernel-trace + --pmc only) on 16,384 pairs of the bench's
# verify workload.  Run on the GPU box:  bash profiles/r01/recipes/pmc_tvg.sh
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r02
OUT=$R/gpurun_out/r02/pmc_tvg_r02.txt
: > $OUT
run() {
  tag=$1; shift
  rm -rf /tmp/pmct_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmct_$tag -- python $R/bench.py --images 40 --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 16384 --no-pipeline --no-dense > /tmp/pmct_$tag.log 2>&1
  echo "rc=$?" >> $OUT
  db=$(find /tmp/pmct_$tag -name "*.db" | head -1)
  echo "=== pass $tag: $@" >> $OUT
  python $R/tools/pmc_summary.py $db tvg_kernel | grep -E "tvg_kernel" >> $OUT
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU
run b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE
cat $OUT
