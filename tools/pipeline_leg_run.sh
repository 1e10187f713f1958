# usage: bash tools/pipeline_leg_run.sh base <name> ...   bench.py's chained pipeline leg (configs[2]) on libamc.so (base) or on
# pycolmap_amd/csrc/_obj/libamc_<name>.so: verified pairs/s, ms per step, verification ms, verification kernels ms
for v in "$@"; do
  if [ $v = base ]; then unset AMC_LIB_PATH; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_$v.so; fi
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense --verify-pairs 2048 --no-config3 --no-config4 --no-ragged --no-sift-stats --no-db --full-line 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['pipeline']; s=d['stage_ms_per_step']; print('$v', round(d['value']), round(d['ms_per_step'],1), round(s['verify_ms'],2), round(s['verify_kernel_ms'],2))"
done
