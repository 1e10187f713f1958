cd $GRAFT_REPO_ROOT
AMC_VERIFY_PROFILE=1 timeout 300 python bench.py --images 40 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-dense --no-ragged --no-db --verify-pairs 124750 2>&1 | grep -E "amc verify profile|\"verify\"" | cut -c1-400
