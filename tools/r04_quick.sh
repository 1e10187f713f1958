cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_match_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
AMC_MATCH_PROFILE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-db 2>&1 | grep -E "amc match profile" | tail -3 | cut -c1-200
bash tools/var_run.sh prev base prev base
