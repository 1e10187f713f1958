# round 6 A/B inside ONE gpurun call: the chained pipeline leg (configs[2]) interleaved vs AMC_PIPELINE_SERIAL=1 (the
# stages behind each other), and the verify leg at 1 / 4 / 8 slices.   bash tools/r06_ab_pipeline.sh <tag> [reps]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; REPS=${2:-2}; OUT=$O/ab_pipeline_$TAG.txt; : > $OUT
PIPE="--steps 3 --warmup 1 --no-cpu-baseline --no-dense --verify-pairs 0 --no-ragged --no-db --no-sift-stats --no-config3"
pipe() {
  python bench.py $PIPE 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['pipeline']; s=p['stage_ms_per_step']; print('$1', 'pairs/s', round(p['value']), 'ms_per_step', round(p['ms_per_step'],2), 'match', round(s['match_ms'],2), 'scan', round(s['scan_ms'],2), 'cross', round(s['cross_ms'],2), 'verify_ms', round(s['verify_ms'],2), 'verify_kernels', round(s['verify_kernel_ms'],2), 'non_scan', round(p['ms_per_step']-s['scan_ms'],2), 'headline_ms', round(d['ms_per_step'],2))"
}
for r in $(seq $REPS); do
  echo "--- rep $r" >> $OUT
  pipe interleaved >> $OUT
  AMC_PIPELINE_SERIAL=1 pipe serial >> $OUT
  AMC_PIPELINE_SERIAL=1 AMC_TVG_SLICES=1 pipe serial_1slice >> $OUT
done
VER="--images 40 --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3"
ver() {
  python bench.py $VER --verify-pairs $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['verify']; print('$1', $2, 'pairs/s', round(d['value']), 'ms', round(d['ms_per_step'],1), 'kernels', round(d['kernel_ms_per_step'],1))"
}
for r in $(seq $REPS); do
  for n in 124750 9585; do
    for sl in 1 2 4 8; do AMC_TVG_SLICES=$sl ver slices$sl $n >> $OUT; done
  done
done
cat $OUT
