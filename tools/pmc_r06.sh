# round 6: counters of the match scan at the bench's default workload (500 images x 4096, the launches of one step), ONE
# counter per pass for the memory-side passes (rocprofv3 died on grouped TCC passes in round 5), the kernel filter on the
# forward scan, everything on ONE box (its boot id goes into the JSON).   bash tools/pmc_r06.sh [tag]
# Writes gpurun_out/r06/pmc_match_r06_<tag>.txt, pmc_hbm_r06_<tag>.{txt,json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-v1}
O=$R/gpurun_out/r06
mkdir -p $O
BOX="boot_id $(cat /proc/sys/kernel/random/boot_id 2>/dev/null) host $(hostname) gpu $(rocm-smi --showserial 2>/dev/null | grep -i serial | head -1 | tr -s ' ' | cut -c1-80)"
BENCH="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3 --no-config4"
run() {  # $1 = out file, $2 = tag, $3... = counters; a pass whose rocprofv3 dies (it happens on some boxes) is retried
  out=$1; tag=$2; shift; shift
  for try in 1 2 3; do
    rm -rf /tmp/pmc6_$tag
    timeout 300 rocprofv3 --kernel-trace --kernel-include-regex 'match_mfma_kernel<0' --pmc "$@" -d /tmp/pmc6_$tag -- $BENCH > /tmp/pmc6_$tag.log 2>&1
    rc=$?
    db=$(find /tmp/pmc6_$tag -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && break
  done
  echo "rc=$rc tries=$try" >> $out
  echo "=== pass $tag: $@" >> $out
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db match_mfma | grep -E "calls=|n=" | grep -v "pmc tables" >> $out
}
M=$O/pmc_match_r06_$TAG.txt; echo "box: $BOX" > $M
run $M a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run $M b SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE
H=$O/pmc_hbm_r06_$TAG.txt; echo "box: $BOX" > $H
run $H f FETCH_SIZE
run $H r TCC_EA0_RDREQ_sum
run $H w TCC_EA0_WRREQ_sum
run $H w64 TCC_EA0_WRREQ_64B_sum
run $H g GRBM_GUI_ACTIVE
python $R/tools/pmc_hbm_json.py $H $O/pmc_hbm_r06_$TAG.json | tail -30
grep -c "rc=0" $H
