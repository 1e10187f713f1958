# round 6: the pose kernel at four waves per SIMD against three (variant library posew3), same box: bench.py's verify leg
# with_relative_pose, and the pose tests on the shipped build.   bash tools/r06_ab_pose.sh <tag> <reps>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; REPS=${2:-2}; OUT=$O/ab_pose_$TAG.txt; : > $OUT
timeout 600 python -m pytest tests/test_pose_gpu.py tests/test_verify_gpu.py -m gpu -x -q 2>&1 | tail -1 >> $OUT
for r in $(seq $REPS); do for v in posew3 base; do
  if [ $v = base ]; then unset AMC_LIB_PATH; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_$v.so; fi
  python bench.py --images 40 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3 --no-config4 --verify-pairs 124750 --full-line 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['verify']; p=d['with_relative_pose']; print('$v', 'verify', round(d['value']), 'with_pose', round(p['value']), 'ms', round(p['ms_per_step'],1), 'pose_kernel_ms', round(p['pose_kernel_ms_per_step'],2))" >> $OUT
done; done
unset AMC_LIB_PATH
timeout 600 python tools/stress_verify.py --rounds 10 --seed 31 2>&1 | tail -1 >> $OUT
cat $OUT
