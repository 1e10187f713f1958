# round 6: pairs per device call of the database pipeline (AMC_GROUP_PAIRS), bench.py's db leg.  bash tools/r06_ab_group.sh <tag> <reps> <n ...>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; REPS=${2:-2}; shift; shift
OUT=$O/ab_group_$TAG.txt; : > $OUT
for r in $(seq $REPS); do for n in "$@"; do
  AMC_GROUP_PAIRS=$n python bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-sift-stats --no-config3 --no-config4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['db']; print('group_pairs', $n, {k: d.get(k) for k in ('wall_s','rerun_wall_s','pairs_with_matches','pairs_verified')})" >> $OUT
done; done
cat $OUT
