cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_match_gpu.py -x -q -m gpu 2>&1 | tail -2
AMC_MFMA_SHAPE=4 timeout 600 python -m pytest tests/test_match_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/stress_match.py 2>&1 | tail -1
bash tools/var_run.sh nokey base nokey base
AMC_MFMA_SHAPE=4 bash tools/var_run.sh nokey base
