# usage: bash tools/diag_run.sh base [1 2 3 ...]   (see tools/diag_build.sh)
for v in "$@"; do
  if [ $v = base ]; then unset AMC_LIB_PATH; elif [ $v = prev ]; then export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_prev.so; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_diag$v.so; fi
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-db 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_ms'])"
done
