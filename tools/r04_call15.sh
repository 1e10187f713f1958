cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
# dry run of the reference kit through this repo's module, then the consuming tests against the dry file
timeout 900 python tests/golden/make_reference_golden.py --module pycolmap_amd --out $O/reference_dry_r04.npz --limit 12 > $O/reference_dry_r04.log 2>&1; echo "kit rc=$?"; tail -3 $O/reference_dry_r04.log
AMC_REFERENCE_GOLDEN=$GRAFT_REPO_ROOT/$O/reference_dry_r04.npz timeout 900 python -m pytest tests/test_reference_golden.py -q -x 2>&1 | tail -5
timeout 900 python -m pytest tests/test_verify_gpu.py -q -x -m gpu 2>&1 | tail -3
