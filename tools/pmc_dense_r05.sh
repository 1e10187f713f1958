# round 5: memory-side counters of the cross-check stage's kernels on the dense set (every pair overlapping): what
# "gather-bound" means in bytes.  One counter group per pass.   bash tools/pmc_dense_r05.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-v1}; mkdir -p $R/gpurun_out/r05
OUT=$R/gpurun_out/r05/pmc_dense_r05_$TAG.txt
: > $OUT
run() {
  tag=$1; shift
  for try in 1 2; do
    rm -rf /tmp/pmcd_$tag
    timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmcd_$tag -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-db --no-sift-stats --no-config3 > /tmp/pmcd_$tag.log 2>&1
    rc=$?
    db=$(find /tmp/pmcd_$tag -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && break
  done
  echo "rc=$rc tries=$try" >> $OUT
  echo "=== pass $tag: $@" >> $OUT
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db amc:: | grep -E "resolve_index|match_mfma_kernel<1|finalize_kernel|select_candidates" | sed -E 's/\(amc::[^)]*\)?[^ ]* +/ /' | cut -c1-230 >> $OUT
}
# (FETCH_SIZE = TCC_EA0_RDREQ x 64 B in KB, MI355X_MICROARCH.md HBM section; the derived counter's pass died on round 5's
# boxes, the raw request counters do not: bytes read = RDREQ x 64 x 2 with the guide's gfx950 correction)
run r TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run h TCC_HIT_sum TCC_MISS_sum
cat $OUT
