cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_v1.log 2>&1; echo "pytest all rc=$?"; tail -5 $O/pytest_gpu_v1.log
AMC_MFMA_SHAPE=4 timeout 900 python -m pytest tests/test_match_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > $O/pytest_match_w4.log 2>&1; echo "pytest w4 rc=$?"; tail -3 $O/pytest_match_w4.log
AMC_SCAN_ACCEPT_TRIVIAL=1 timeout 900 python -m pytest tests/test_match_gpu.py -x -q -m gpu > $O/pytest_match_trivial.log 2>&1; echo "pytest trivial rc=$?"; tail -3 $O/pytest_match_trivial.log
timeout 600 python tools/stress_match.py > $O/stress_match_v1.txt 2>&1; echo "stress rc=$?"; tail -5 $O/stress_match_v1.txt
