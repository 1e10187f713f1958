# usage: bash tools/verify_run.sh <npairs> base prev w3 ...   (verify leg only; variants: _obj/libamc_prev.so, libamc_diag<v>.so)
n=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset AMC_LIB_PATH; elif [ $v = prev ]; then export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_prev.so; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_diag$v.so; fi
  timeout 250 python bench.py --images 40 --steps 2 --warmup 1 --no-cpu-baseline --verify-pairs $n 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['verify']; print('$v', $n, round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])"
done
