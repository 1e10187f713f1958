cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -1
for v in a b; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-dense 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['db']; s=d['stats']
print('$v', 'wall', round(d['wall_s'],3), 'rerun', round(d['rerun_wall_s'],3), 'match_total', round(s['match_total_ms']), 'match_dev', round(s['match_device_ms']), 'verify_dev', round(s['verify_device_ms']), 'write', round(s['write_ms']), 'setup', round(s['setup_ms']), 'rest', round(1e3*d['wall_s']-s['setup_ms']-s['match_total_ms']), 'teardown', round(s.get('teardown_ms',-1)), 'call', round(s.get('call_ms',-1)))"
done
