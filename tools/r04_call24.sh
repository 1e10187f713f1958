cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
( time timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_v2.log 2>&1 ) 2>&1 | grep real; tail -4 $O/pytest_gpu_v2.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -4
