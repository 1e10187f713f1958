# round 6: a batch's cross-check chain beside the next batch's forward scan (AMC_MATCH_OVERLAP=1, AMC_CHAIN_CUS=n)
# against the serial order, same box, same library.   bash tools/r06_ab_overlap.sh <tag> <reps> [cus ...]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; REPS=${2:-2}; shift; shift
CUS="${@:-8}"
OUT=$O/ab_overlap_$TAG.txt; : > $OUT
HEAD="--steps 8 --warmup 2 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-db --no-sift-stats --no-config3 --no-config4"
line() {
  python bench.py $HEAD 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; e=d.get('dense') or {}; print('$1', 'headline_ms', round(d['ms_per_step'],2), 'scan_ms', round(r['avg_kernel_ms'],3), 'scan_frac', round(r['frac'],4), 'whole_step_frac', round(r['whole_step_frac'],4), '| dense_ms', e.get('ms_per_step'), 'dense_stage_ms', e.get('resolve_select_reverse_scan_ms'))" >> $OUT
}
# parity first: the match / pipeline / guided tests with the overlap on, small batches included
AMC_MATCH_OVERLAP=1 timeout 900 python -m pytest tests/test_match_gpu.py tests/test_pipeline_gpu.py tests/test_guided_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
AMC_MATCH_OVERLAP=1 timeout 600 python tools/stress_match.py --rounds 6 --seed 11 2>&1 | tail -2 >> $OUT
for r in $(seq $REPS); do
  AMC_MATCH_OVERLAP=0 line serial
  for n in $CUS; do AMC_MATCH_OVERLAP=1 AMC_CHAIN_CUS=$n line overlap_cus$n; done
done
AMC_MATCH_OVERLAP=1 AMC_CHAIN_CUS=8 AMC_MATCH_PROFILE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-dense --no-db --no-sift-stats --no-config3 --no-config4 2>&1 | grep "amc match profile" | tail -2 >> $OUT
cat $OUT
