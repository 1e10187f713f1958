# round 6: a single-batch amc_match_verify_pairs call cut in two so that the first part's verification host work runs
# beside the second part's scan (AMC_HOOK_SPLIT=0: off), bench.py's db leg (four calls of ~33 k pairs).  bash tools/r06_ab_hooksplit.sh <tag> <reps>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; REPS=${2:-3}; OUT=$O/ab_hooksplit_$TAG.txt; : > $OUT
timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_verify_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1 >> $OUT
for r in $(seq $REPS); do for v in 0 1; do
  AMC_HOOK_SPLIT=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-sift-stats --no-config3 --no-config4 --full-line 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['db']; print('hook_split', $v, {k: (round(d[k],4) if isinstance(d.get(k), float) else d.get(k)) for k in ('wall_s','rerun_wall_s','device_s','match_device_ms','verify_device_ms','pairs_verified')}, {k: round(v,1) for k, v in (d.get('stats') or {}).items() if isinstance(v, float)})" >> $OUT
done; done
cat $OUT
