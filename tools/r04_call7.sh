cd $GRAFT_REPO_ROOT
for shape in 8 4; do
  export AMC_MFMA_SHAPE=$shape
  echo "== shape $shape"
  bash tools/diag_run.sh base 1 8 12 2 base
done
