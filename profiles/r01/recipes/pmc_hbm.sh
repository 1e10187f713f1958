# HBM traffic of the match kernel at the bench's default workload (500 images x 4096, 2 launches per
# step), FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots: MI355X_MICROARCH.md, "rocprofv3 PMC
# slots"), --kernel-trace + --pmc only.  Run on the GPU box:  bash profiles/r01/recipes/pmc_hbm.sh
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_hbm_v6.txt
: > $OUT
run() {  # $1 = tag, $2... = counters
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$tag -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 > /tmp/pmc_$tag.log 2>&1
  echo "rc=$?" >> $OUT
  db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  echo "=== pass $tag: $@" >> $OUT
  python $R/tools/pmc_summary.py $db match_mfma | grep -E "calls=|n=" | grep -v "pmc tables" >> $OUT
  tail -2 /tmp/pmc_$tag.log | cut -c1-600 >> $OUT
}
run f FETCH_SIZE
run w WRITE_SIZE
run g GRBM_GUI_ACTIVE
cat $OUT
