cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_match_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/var_run.sh notouch base epfirst epfirst_notouch notouch base epfirst epfirst_notouch
