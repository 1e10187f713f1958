# round 5: counters of the match scan at the bench's default workload (500 images x 4096, the launches of one step), one
# counter group per pass (--kernel-trace + --pmc only), Writes gpurun_out/r05/pmc_match_r05_<tag>.txt, pmc_hbm_r05_<tag>.{txt,json}, rocprofv3_kernel_stats_*.csv
#   bash tools/pmc_r05.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-v1}
O=$R/gpurun_out/r05
mkdir -p $O
BENCH="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3"
run() {  # $1 = out file, $2 = tag, $3... = counters; a pass whose rocprofv3 dies (it happens on some boxes) is retried
  out=$1; tag=$2; shift; shift
  for try in 1 2 3; do
    rm -rf /tmp/pmc4_$tag
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc4_$tag -- $BENCH > /tmp/pmc4_$tag.log 2>&1
    rc=$?
    db=$(find /tmp/pmc4_$tag -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && break
  done
  echo "rc=$rc tries=$try" >> $out
  echo "=== pass $tag: $@" >> $out
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db match_mfma | grep -E "calls=|n=" | grep -v "pmc tables" >> $out
}
M=$O/pmc_match_r05_$TAG.txt; : > $M
run $M a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run $M b SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE
run $M g GRBM_GUI_ACTIVE
H=$O/pmc_hbm_r05_$TAG.txt; : > $H
run $H f FETCH_SIZE
run $H w TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run $H g GRBM_GUI_ACTIVE
python $R/tools/pmc_hbm_json.py $H $O/pmc_hbm_r05_$TAG.json
cat $M $H
