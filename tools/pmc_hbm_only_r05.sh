# only the HBM passes of tools/pmc_r05.sh (FETCH_SIZE, TCC_EA0_WRREQ, GRBM), for a retry on another box when rocprofv3 died
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-v1}; O=$R/gpurun_out/r05; mkdir -p $O
BENCH="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3"
H=$O/pmc_hbm_r05_$TAG.txt; : > $H
run() {
  tag=$1; shift
  for try in 1 2; do
    rm -rf /tmp/pmch_$tag
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmch_$tag -- $BENCH > /tmp/pmch_$tag.log 2>&1
    rc=$?
    db=$(find /tmp/pmch_$tag -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && break
  done
  echo "rc=$rc tries=$try" >> $H
  echo "=== pass $tag: $@" >> $H
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db match_mfma | grep -E "calls=|n=" | grep -v "pmc tables" >> $H
}
# (round 5: the FETCH_SIZE pass died three times in a row on one box - the raw read-request counter it is derived from is
# taken as well, pmc_hbm_json.py uses it when FETCH_SIZE is missing)
run f FETCH_SIZE
run r TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run w TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run g GRBM_GUI_ACTIVE
python $R/tools/pmc_hbm_json.py $H $O/pmc_hbm_r05_$TAG.json | head -20
grep "rc=" $H
