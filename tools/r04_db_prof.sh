cd $GRAFT_REPO_ROOT
AMC_DESTROY_PROFILE=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-dense 2>&1 | grep "amc destroy" | tail -24
