# round 6: memory-side counters of the cross-check stage's kernels on the dense set (every pair overlapping), ONE counter
# per pass, the kernel filter on the stage's kernels (every grouped TCC pass died with rc 139).
#   bash tools/pmc_dense_r06.sh [tag]   -> gpurun_out/r06/pmc_dense_r06_<tag>.txt (+ .json through tools/pmc_dense_json.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-v1}; O=$R/gpurun_out/r06; mkdir -p $O
OUT=$O/pmc_dense_r06_$TAG.txt
echo "box: boot_id $(cat /proc/sys/kernel/random/boot_id 2>/dev/null)" > $OUT
BENCH="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-db --no-sift-stats --no-config3 --no-config4 --detail-json /tmp/dense_line.json"
run() {
  tag=$1; shift
  for try in 1 2 3; do
    rm -rf /tmp/pmcd_$tag
    timeout 400 rocprofv3 --kernel-trace --kernel-include-regex 'resolve_index|match_mfma_kernel<1|select_candidates|finalize_kernel|match_mfma_kernel<0' --pmc "$@" -d /tmp/pmcd_$tag -- $BENCH > /tmp/pmcd_$tag.log 2>&1
    rc=$?
    db=$(find /tmp/pmcd_$tag -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && break
  done
  echo "rc=$rc tries=$try" >> $OUT
  echo "=== pass $tag: $@" >> $OUT
  [ -n "$db" ] && python $R/tools/pmc_dispatches.py $db resolve_index match_mfma_kernel finalize_kernel select_candidates >> $OUT
}
run f FETCH_SIZE
run r TCC_EA0_RDREQ_sum
run w TCC_EA0_WRREQ_sum
run w64 TCC_EA0_WRREQ_64B_sum
run h TCC_HIT_sum
run m TCC_MISS_sum
run t GRBM_GUI_ACTIVE
run i SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
python -c "
import json; d=json.loads(open('/tmp/dense_line.json').read())['dense']; print('dense leg under the profiler:', {k: d[k] for k in ('ms_per_step','matches_per_pair','scan_kernel_ms','resolve_select_reverse_scan_ms')})" >> $OUT 2>/dev/null
cat $OUT
