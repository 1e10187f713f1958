cd $GRAFT_REPO_ROOT
AMC_MATCH_PROFILE=1 AMC_VERIFY_PROFILE=1 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-dense 2>&1 | grep -E "amc (match|verify) profile|amc match" | tail -30 | cut -c1-330
