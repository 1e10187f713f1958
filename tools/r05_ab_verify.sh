# Same-box A/B of the verification kernels: bench.py's verify leg (124,750 pairs) and the pipeline leg's verification on
# libamc.so (base) and on variant builds (tools/variant_build_tvg.sh <name> "<-D flags>").
#   bash tools/r05_ab_verify.sh <tag> base e3 e1 ...   -> gpurun_out/r05/ab_verify_<tag>.txt
R=${GRAFT_REPO_ROOT:-.}
TAG=$1; shift
mkdir -p $R/gpurun_out/r05
OUT=$R/gpurun_out/r05/ab_verify_$TAG.txt
: > $OUT
for rep in 1 2; do
  for v in "$@"; do
    if [ $v = base ]; then unset AMC_LIB_PATH; else export AMC_LIB_PATH=$R/pycolmap_amd/csrc/_obj/libamc_$v.so; fi
    timeout 400 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense --no-ragged --no-db --no-sift-stats --no-config3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['verify']; p=d['pipeline']; sm=p['stage_ms_per_step']
print('$v rep $rep: verify %.0f pairs/s, call %.1f ms, kernels %.1f ms | pipeline verify %.2f ms (kernels %.2f), %d pairs' % (v['value'], v['ms_per_step'], v['kernel_ms_per_step'], sm['verify_ms'], sm['verify_kernel_ms'], p['pairs_verified']))" >> $OUT
  done
done
unset AMC_LIB_PATH
cat $OUT
