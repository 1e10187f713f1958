# round 6: headline + ragged leg, the shipped library against a variant (tools/var_build.sh <name> ...), same box.
#   bash tools/r06_ab_ragged.sh <reps> base <name> ...
cd $GRAFT_REPO_ROOT
REPS=$1; shift
for r in $(seq $REPS); do
for v in "$@"; do
  if [ $v = base ]; then unset AMC_LIB_PATH; else export AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_var_$v.so; fi
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-db --no-sift-stats --no-config3 --no-config4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['ragged']; print('$v', 'headline_ms', round(d['ms_per_step'],2), 'scan_frac', round(d['roofline']['frac'],4), '| ragged_ms', round(r['ms_per_step'],2), 'vs_uniform', round(r['vs_uniform'],4), 'ragged_scan_ms', round(r['scan_kernel_ms'],2), 'ragged_scan_frac', round(r['scan_frac_of_int8_peak'],4))"
done; done
