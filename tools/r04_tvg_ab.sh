# verification A/B inside one gpurun call: tests on the shipped library, then the verify leg on base / variants
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_verify_gpu.py tests/test_estimators_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/stress_verify.py --rounds 6 --pairs 300 2>&1 | tail -2
bash tools/verify_run.sh 124750 "$@"
