cd $GRAFT_REPO_ROOT
for v in cpu nocpu cpu nocpu; do
  if [ $v = cpu ]; then F=""; else F="--no-cpu-baseline"; fi
  timeout 300 python bench.py $F --no-pipeline --no-dense --no-ragged --no-db 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['verify']
print('$v', 'headline', round(d['ms_per_step'],1), d['roofline']['frac'], 'verify', round(v['value']), round(v['ms_per_step'],1), v.get('kernel_ms_per_step'))"
done
