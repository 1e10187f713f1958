cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 1200 python -m pytest tests/test_verify_gpu.py -x -q -m gpu -k "more_matches or large_match" > $O/pytest_verify_big.log 2>&1; echo "rc=$?"; tail -15 $O/pytest_verify_big.log
