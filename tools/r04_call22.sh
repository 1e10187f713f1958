cd $GRAFT_REPO_ROOT
AMC_VERIFY_PROFILE=1 AMC_MATCH_PROFILE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-dense --no-ragged --no-db 2>&1 | grep -E "amc (verify|match) profile|^\{" | tail -12 | cut -c1-400
