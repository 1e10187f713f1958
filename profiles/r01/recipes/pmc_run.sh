set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_v3.txt
: > $OUT
run() {  # $1 = tag, $2... = counters
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$tag -- python $R/bench.py --images 160 --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 > /tmp/pmc_$tag.log 2>&1
  db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  echo "=== pass $tag: $@" >> $OUT
  python $R/tools/pmc_summary.py $db match_mfma | grep -E "calls=|n=" | grep -v "pmc tables" >> $OUT
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run b SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC
run c GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE
tail -5 /tmp/pmc_c.log
wc -c $OUT
