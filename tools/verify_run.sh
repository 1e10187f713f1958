# usage: bash tools/verify_run.sh base w3 w4 ...   (verify leg only; variants: pycolmap_amd/csrc/_obj/libamc_diag<v>.so)
for v in "$@"; do
  if [ $v = base ]; then unset AMC_LIB_PATH; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_diag$v.so; fi
  timeout 200 python bench.py --images 40 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['verify']; print('$v', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])"
done
