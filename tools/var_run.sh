# usage: bash tools/var_run.sh base name1 name2 ...   (tools/var_build.sh)
for v in "$@"; do
  if [ $v = base ]; then unset AMC_LIB_PATH; elif [ $v = prev ]; then export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_prev.so; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_var_$v.so; fi
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-db 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['avg_kernel_ms'],2))"
done
