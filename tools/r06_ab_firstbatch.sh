# round 6: a call's first batch small (AMC_MATCH_FIRST_DIV=d) against a full first batch, same box.
#   bash tools/r06_ab_firstbatch.sh <tag> <reps> [d ...]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; REPS=${2:-2}; shift; shift
DIVS="${@:-8}"
OUT=$O/ab_firstbatch_$TAG.txt; : > $OUT
HEAD="--steps 10 --warmup 2 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-dense --no-db --no-sift-stats --no-config3 --no-config4"
line() {
  python bench.py $HEAD 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'headline_ms', round(d['ms_per_step'],2), 'scan_ms_sum', round(r['avg_kernel_ms']*r['launches_per_step'],2), 'launches', r['launches_per_step'], 'non_scan_ms', round(d['ms_per_step']-r['avg_kernel_ms']*r['launches_per_step'],2), 'whole_step_frac', round(r['whole_step_frac'],4))" >> $OUT
}
for r in $(seq $REPS); do
  AMC_MATCH_FIRST_DIV=0 line full
  for n in $DIVS; do AMC_MATCH_FIRST_DIV=$n line first_div$n; done
done
cat $OUT
