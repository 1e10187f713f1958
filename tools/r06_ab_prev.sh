# same-box A/B of the current library against pycolmap_amd/csrc/_obj/libamc_prev.so (tools/ab_prev_lib.sh <rev>):
# verify leg (124,750 and 9,585 pairs) and the chained pipeline leg.   bash tools/r06_ab_prev.sh <tag> [reps]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; REPS=${2:-2}; OUT=$O/ab_prev_$TAG.txt; : > $OUT
PREV=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_prev.so
VER="--images 40 --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3"
ver() {
  python bench.py $VER --verify-pairs $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['verify']; print('$1', $2, 'pairs/s', round(d['value']), 'ms', round(d['ms_per_step'],1), 'kernels', round(d['kernel_ms_per_step'],1))"
}
PIPE="--steps 3 --warmup 1 --no-cpu-baseline --no-dense --verify-pairs 0 --no-ragged --no-db --no-sift-stats --no-config3"
pipe() {
  python bench.py $PIPE 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['pipeline']; s=p['stage_ms_per_step']; print('$1', 'pairs/s', round(p['value']), 'ms_per_step', round(p['ms_per_step'],2), 'match', round(s['match_ms'],2), 'scan', round(s['scan_ms'],2), 'verify_ms', round(s['verify_ms'],2), 'verify_kernels', round(s['verify_kernel_ms'],2), 'non_scan', round(p['ms_per_step']-s['scan_ms'],2), 'headline_ms', round(d['ms_per_step'],2))"
}
for r in $(seq $REPS); do
  echo "--- rep $r" >> $OUT
  AMC_LIB_PATH=$PREV ver prev 124750 >> $OUT
  AMC_TVG_SLICES=1 ver cur_slices1 124750 >> $OUT
  AMC_TVG_SLICES=2 ver cur_slices2 124750 >> $OUT
  AMC_LIB_PATH=$PREV ver prev 9585 >> $OUT
  AMC_TVG_SLICES=1 ver cur_slices1 9585 >> $OUT
  AMC_LIB_PATH=$PREV pipe prev >> $OUT
  pipe cur_interleaved >> $OUT
  AMC_PIPELINE_SERIAL=1 AMC_TVG_SLICES=1 pipe cur_serial >> $OUT
done
cat $OUT
