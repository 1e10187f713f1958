cd $GRAFT_REPO_ROOT
for v in base prev; do
  if [ $v = base ]; then unset AMC_LIB_PATH; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_$v.so; fi
  echo "== $v"
  AMC_VERIFY_PROFILE=1 timeout 300 python bench.py --images 40 --steps 4 --warmup 1 --no-cpu-baseline --no-pipeline --no-dense --no-ragged --no-db --verify-pairs 124750 2>&1 | grep -E "amc verify profile|\"verify\"" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('[amc'): print(l.strip()[22:])
    else:
        d=json.loads(l)['verify']; print('leg', round(d['value']), round(d['ms_per_step'],1), round(d['kernel_ms_per_step'],1))"
done
