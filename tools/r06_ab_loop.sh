# round 6: loop-detection queries per device call in match_sequential (AMC_LOOP_QUERIES_PER_CALL=1: one call per query,
# as before; unset: ~2^19 pairs per call), same box, same database recipe.   bash tools/r06_ab_loop.sh <tag> <images> <feats>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; N=${2:-3000}; F=${3:-1024}; OUT=$O/ab_loop_$TAG.txt; : > $OUT
timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -1 >> $OUT
for v in ${VARIANTS:-1 batched 1 batched}; do
  if [ $v = batched ]; then unset AMC_LOOP_QUERIES_PER_CALL; else export AMC_LOOP_QUERIES_PER_CALL=$v; fi
  timeout 900 python tools/pipeline_bench.py --images $N --feats $F --loop --loop-features 512 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stats']
print('queries_per_call', '$v', 'wall_s', round(d['wall_s'],3), 'loop_queries', s.get('loop_queries'), 'loop_pairs_scored', s.get('loop_pairs_scored'), 'loop_device_ms', round(s.get('loop_device_ms',0),1), 'matched_pairs', d['matched_pairs'], 'verified_pairs', d['verified_pairs'], 'num_matches', d['num_matches'], 'num_inlier_matches', d['num_inlier_matches'])" >> $OUT
done
cat $OUT
