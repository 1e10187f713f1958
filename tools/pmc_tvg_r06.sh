# round 6: SQ counters of the two verification kernels (tvg_e_kernel, tvg_fh_kernel) on 16,384 pairs of the bench's
# verify workload, one counter group per pass (--kernel-trace + --pmc only; a pass whose rocprofv3 dies is retried).
#   bash tools/pmc_tvg_r06.sh [tag]      -> gpurun_out/r06/pmc_tvg_r06_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-v1}; mkdir -p $R/gpurun_out/r06
OUT=$R/gpurun_out/r06/pmc_tvg_r06_$TAG.txt
: > $OUT
run() {
  tag=$1; shift
  for try in 1 2 3; do
    rm -rf /tmp/pmct_$tag
    AMC_TVG_SLICES=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmct_$tag -- python $R/bench.py --images 40 --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 16384 --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3 > /tmp/pmct_$tag.log 2>&1
    rc=$?
    db=$(find /tmp/pmct_$tag -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && break
  done
  echo "rc=$rc tries=$try" >> $OUT
  echo "=== pass $tag: $@" >> $OUT
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db tvg_ | grep -E "tvg_(e|fh)_kernel" >> $OUT
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU
run b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE
# round 6: lane utilisation (thread-cycles of vector instructions over 64 x their issue cycles) and the
# bytes the kernels move through the L2's memory side (item 6: verify.roofline.traffic), one TCC counter per pass
run lanes SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES
run fetch FETCH_SIZE
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
cat $OUT
