# round 3: SQ / GRBM counters of the match kernel at the bench's default workload (500 images x 4096, 2 launches per
# step), one counter group per pass (--kernel-trace + --pmc only).  Writes gpurun_out/r03/pmc_match_r03.txt.
#   bash profiles/r03/recipes/pmc_match_r03.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-v1}
mkdir -p $R/gpurun_out/r03
OUT=$R/gpurun_out/r03/pmc_match_r03_$TAG.txt
: > $OUT
run() {  # $1 = tag, $2... = counters; a pass whose rocprofv3 dies (segmentation faults happen on some boxes) is retried
  tag=$1; shift
  for try in 1 2 3; do
    rm -rf /tmp/pmcm_$tag
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmcm_$tag -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense > /tmp/pmcm_$tag.log 2>&1
    rc=$?
    db=$(find /tmp/pmcm_$tag -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && break
  done
  echo "rc=$rc tries=$try" >> $OUT
  echo "=== pass $tag: $@" >> $OUT
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db match_mfma | grep -E "calls=|n=" | grep -v "pmc tables" >> $OUT
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run b SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE
run g GRBM_GUI_ACTIVE
cat $OUT
