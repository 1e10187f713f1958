# round 6 probes, one gpurun call: (1) host-phase timeline of the chained pipeline leg (AMC_MATCH_PROFILE / AMC_VERIFY_PROFILE),
# (2) lane utilisation of the two verification kernels (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU), (3) the per-phase cycle
# report of the -DAMC_TVG_LODIAG build (tools/variant_build_tvg.sh lodiag "-DAMC_TVG_LODIAG" first).
#   bash tools/r06_probe.sh <tag>     -> gpurun_out/r06/probe_<tag>.txt
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; OUT=$O/probe_$TAG.txt; : > $OUT
PIPE="--steps 2 --warmup 1 --no-cpu-baseline --no-dense --verify-pairs 0 --no-ragged --no-db --no-sift-stats --no-config3"
echo "=== pipeline leg, host phases" >> $OUT
AMC_MATCH_PROFILE=1 AMC_VERIFY_PROFILE=1 timeout 300 python bench.py $PIPE 2>&1 >/tmp/pipe.json | grep "amc .* profile" >> $OUT
python -c "import json; d=json.loads(open('/tmp/pipe.json').read().strip().splitlines()[-1])['pipeline']; print(d['ms_per_step'], d['stage_ms_per_step'])" >> $OUT
if [ -f pycolmap_amd/csrc/_obj/libamc_lodiag.so ]; then
  echo "=== LODIAG build, verify leg 16384 pairs" >> $OUT
  AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_lodiag.so AMC_TVG_PROFILE=1 timeout 300 python bench.py --images 40 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3 --verify-pairs 16384 2>&1 >/dev/null | grep "amc tvg" | tail -12 >> $OUT
fi
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  tag=$1; shift
  for try in 1 2; do
    rm -rf /tmp/pmct_$tag
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmct_$tag -- python $R/bench.py --images 40 --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 16384 --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3 > /tmp/pmct_$tag.log 2>&1
    rc=$?
    db=$(find /tmp/pmct_$tag -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && break
  done
  echo "rc=$rc tries=$try" >> $OUT
  echo "=== pass $tag: $@" >> $OUT
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db tvg_ | grep -E "tvg_(e|fh)_kernel" >> $OUT
}
run lanes SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES
run fetch FETCH_SIZE
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
cat $OUT
