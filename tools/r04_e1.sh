cd $GRAFT_REPO_ROOT
for v in base e1 base e1; do
  if [ $v = base ]; then unset AMC_LIB_PATH; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_$v.so; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-ragged --no-dense 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['pipeline']['stage_ms_per_step']; s=d['db']['stats']
print('$v', 'pipeline verify_ms', round(p['verify_ms'],2), 'kernels', round(p['verify_kernel_ms'],2), '| db wall', round(d['db']['wall_s'],3), 'verify_dev', round(s['verify_device_ms'],1))"
done
