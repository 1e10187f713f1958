cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
timeout 900 python -m pytest tests/test_match_gpu.py -x -q -m gpu > $O/pytest_match_w8.log 2>&1; echo "pytest w8 rc=$?"; tail -2 $O/pytest_match_w8.log
for shape in 8 4 8 4; do
  export AMC_MFMA_SHAPE=$shape
  echo "== shape $shape"
  bash tools/diag_run.sh prev base
done
