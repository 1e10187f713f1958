# round 6: the shipped pipeline (host sides interleaved, kernels behind the match) against the stages fully behind each other
# (AMC_PIPELINE_SERIAL=1), the device-interleaved variant (AMC_PIPELINE_INTERLEAVE=1) and the round-5 library, one box.
#   bash tools/r06_ab_final.sh <tag> [reps]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; REPS=${2:-3}; OUT=$O/ab_final_$TAG.txt; : > $OUT
PREV=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_prev.so
PIPE="--steps 6 --warmup 1 --no-cpu-baseline --no-dense --verify-pairs 0 --no-ragged --no-db --no-sift-stats --no-config3 --pipeline-steps 6"
pipe() {
  python bench.py $PIPE 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['pipeline']; s=p['stage_ms_per_step']; print('$1', 'pairs/s', round(p['value']), 'ms_per_step', round(p['ms_per_step'],2), 'match', round(s['match_ms'],2), 'scan', round(s['scan_ms'],2), 'cross', round(s['cross_ms'],2), 'verify_ms', round(s['verify_ms'],2), 'verify_kernels', round(s['verify_kernel_ms'],2), 'non_scan', round(p['ms_per_step']-s['scan_ms'],2), 'headline_ms', round(d['ms_per_step'],2))"
}
for r in $(seq $REPS); do
  echo "--- rep $r" >> $OUT
  [ -f $PREV ] && AMC_LIB_PATH=$PREV pipe round5_library >> $OUT
  AMC_PIPELINE_SERIAL=1 pipe serial >> $OUT
  pipe shipped >> $OUT
  AMC_PIPELINE_INTERLEAVE=1 pipe device_interleaved >> $OUT
done
AMC_MATCH_PROFILE=1 AMC_VERIFY_PROFILE=1 python bench.py $PIPE 2>&1 >/dev/null | grep "amc .* profile" | grep 124750 | tail -6 >> $OUT
cat $OUT
