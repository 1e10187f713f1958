# round 6: the two shapes of the forward scan on whatever box this is.  bash tools/r06_ab_shape.sh <tag> <reps>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; REPS=${2:-2}; OUT=$O/ab_shape_$TAG.txt; : > $OUT
HEAD="--steps 8 --warmup 2 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-dense --no-db --no-sift-stats --no-config3 --no-config4"
for r in $(seq $REPS); do for sh in 8 4; do
  AMC_MFMA_SHAPE=$sh python bench.py $HEAD 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('shape', $sh, 'headline_ms', round(d['ms_per_step'],2), 'scan_ms', round(r['avg_kernel_ms'],3), 'scan_frac', round(r['frac'],4), 'whole', round(r['whole_step_frac'],4))" >> $OUT
done; done
cat $OUT
