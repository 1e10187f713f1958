cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_match_gpu.py tests/test_pipeline_gpu.py tests/test_guided_gpu.py -x -q -m gpu 2>&1 | tail -1
for v in base prev base; do
  if [ $v = base ]; then unset AMC_LIB_PATH; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_$v.so; fi
  timeout 300 python bench.py --config 4 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('$v', '%.3e'%d['value'], round(d['ms_per_step'],1), 'frac', round(r.get('frac',0),4), 'launches', r.get('launches_per_step'))"
done
unset AMC_LIB_PATH
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-db 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['dense']
print('headline', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), '| dense', '%.3e'%e['value'], round(e['ms_per_step'],1))"
