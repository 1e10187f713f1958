# round 2: HBM traffic of the match kernel at the bench's default workload (500 images x 4096, 2 launches per
# step).  FETCH_SIZE, TCC_EA0_WRREQ / _64B and GRBM_GUI_ACTIVE in SEPARATE passes (--kernel-trace + --pmc only).
# Writes gpurun_out/r02/pmc_hbm_r02.txt.   bash profiles/r02/recipes/pmc_hbm_r02.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02
OUT=$R/gpurun_out/r02/pmc_hbm_r02.txt
: > $OUT
run() {  # $1 = tag, $2... = counters
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$tag -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense > /tmp/pmc_$tag.log 2>&1
  echo "rc=$?" >> $OUT
  db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  echo "=== pass $tag: $@" >> $OUT
  python $R/tools/pmc_summary.py $db match_mfma | grep -E "calls=|n=" | grep -v "pmc tables" >> $OUT
}
run f FETCH_SIZE
run w TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run g GRBM_GUI_ACTIVE
cat $OUT
