# verification A/B inside one gpurun call: tests on the shipped library, then the verify leg on base / variants
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_verify_gpu.py tests/test_estimators_gpu.py tests/test_pipeline_gpu.py tests/test_pose_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/stress_verify.py --rounds 4 --pairs 300 2>&1 | tail -1
AMC_VERIFY_PROFILE=1 timeout 300 python bench.py --images 40 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline --no-dense --no-ragged --no-db --verify-pairs 124750 2>&1 | grep -E "amc verify profile" | tail -1
bash tools/verify_run.sh 124750 "$@"
