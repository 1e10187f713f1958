cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_verify_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -1
for v in low high serial low high serial; do
  unset AMC_TVG_SERIAL_CLASSES; export AMC_AUX_PRIORITY=$v
  if [ $v = serial ]; then export AMC_TVG_SERIAL_CLASSES=1; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-ragged --no-dense 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pipeline']['stage_ms_per_step']; s=d['db']['stats']
print('$v', 'pipeline', round(d['pipeline']['ms_per_step'],1), 'verify_ms', round(p['verify_ms'],2), 'kernels', round(p['verify_kernel_ms'],2), '| db wall', round(d['db']['wall_s'],3), 'verify_dev', round(s['verify_device_ms'],1))"
done
