# usage: bash tools/verify_run.sh <npairs> base <name> ...   bench.py's verify leg only, on libamc.so (base) or on
# pycolmap_amd/csrc/_obj/libamc_<name>.so (tools/variant_build_tvg.sh, tools/ab_build.sh -> prev)
n=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset AMC_LIB_PATH; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_$v.so; fi
  timeout 250 python bench.py --images 40 --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3 --verify-pairs $n 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['verify']; print('$v', $n, round(d['value']), round(d['ms_per_step'],1), round(d['kernel_ms_per_step'],1))"
done
