cd $GRAFT_REPO_ROOT
for v in prev base prev base; do
  if [ $v = base ]; then unset AMC_LIB_PATH; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_prev.so; fi
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense --no-ragged --no-db 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['verify']; p=d['pipeline']
print('$v', 'verify', round(v['value']), round(v['ms_per_step'],1), v.get('stage_ms_per_step'), '| pipeline', round(p['value']), p['stage_ms_per_step'])"
done
