# Same-box A/B of the cross-check stage (resolve / select / finalize): headline + dense legs with the previous
# library (tools/ab_prev_lib.sh <rev> -> pycolmap_amd/csrc/_obj/libamc_prev.so) and the current one, twice each.
#   bash tools/r05_ab_dense.sh [tag]  -> gpurun_out/r05/ab_dense_<tag>.txt
R=${GRAFT_REPO_ROOT:-.}
TAG=${1:-v1}
mkdir -p $R/gpurun_out/r05
OUT=$R/gpurun_out/r05/ab_dense_$TAG.txt
: > $OUT
ARGS="--steps 5 --warmup 2 --verify-pairs 0 --no-pipeline --no-ragged --no-sift-stats --no-db --no-cpu-baseline --no-config3"
for rep in 1 2; do
  for which in prev cur; do
    if [ $which = prev ]; then export AMC_LIB_PATH=$R/pycolmap_amd/csrc/_obj/libamc_prev.so; else unset AMC_LIB_PATH; fi
    python $R/bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); dn=d['dense']
print('$which rep $rep: headline %.2f ms/step (scan %.2f ms/launch x %d); dense %.2f ms/step: scan %.2f cross %.2f device %.2f' % (d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['launches_per_step'], dn['ms_per_step'], dn['stage_ms_per_step']['scan_kernel'], dn['stage_ms_per_step']['resolve_select_reverse_scan'], dn['stage_ms_per_step']['device_total_incl_d2h']))" >> $OUT
  done
done
unset AMC_LIB_PATH
cat $OUT
