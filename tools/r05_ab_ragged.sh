R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for which in prev cur; do
    if [ $which = prev ]; then export AMC_LIB_PATH=$R/pycolmap_amd/csrc/_obj/libamc_prev.so; else unset AMC_LIB_PATH; fi
    python $R/bench.py --steps 5 --warmup 2 --verify-pairs 0 --no-pipeline --no-sift-stats --no-dense --no-db --no-cpu-baseline --no-config3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['ragged']
print('$which rep $rep: headline %.2f | ragged %.2f ms scan %.2f cross %.2f other %.2f vs_uniform %.3f' % (d['ms_per_step'], r['ms_per_step'], r['scan_kernel_ms'], r['resolve_select_reverse_scan_ms'], r['ms_per_step']-r['scan_kernel_ms']-r['resolve_select_reverse_scan_ms'], r['vs_uniform']))"
  done
done
