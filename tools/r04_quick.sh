cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_match_gpu.py tests/test_pipeline_gpu.py tests/test_guided_gpu.py -x -q -m gpu 2>&1 | tail -2
for v in memcpy fused memcpy fused; do
  if [ $v = memcpy ]; then export AMC_D2H=memcpy; else unset AMC_D2H; fi
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-db 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['dense']
print('$v', 'headline', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), '| dense', '%.3e'%e['value'], round(e['ms_per_step'],1), {k:round(x,1) for k,x in e['stage_ms_per_step'].items() if k in ('scan_kernel','resolve_select_reverse_scan','device_total_incl_d2h')})"
done
