cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests/test_match_gpu.py -x -q -m gpu > $O/pytest_match_w8.log 2>&1; echo "pytest w8 rc=$?"
tail -3 $O/pytest_match_w8.log
AMC_MFMA_SHAPE=4 timeout 900 python -m pytest tests/test_match_gpu.py -x -q -m gpu > $O/pytest_match_w4.log 2>&1; echo "pytest w4 rc=$?"
tail -3 $O/pytest_match_w4.log
run() {
  name=$1; shift
  env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline > $O/bench_ab_$name.json 2> $O/bench_ab_$name.err
  tail -1 $O/bench_ab_$name.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; g=d.get('ragged',{}); e=d.get('dense',{})
print('$name', 'ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'kern_ms', round(r['avg_kernel_ms'],2), '| ragged', '%.3e'%g.get('value',0), 'vs_uniform', g.get('vs_uniform'), 'scanfrac', g.get('scan_frac_of_int8_peak'), '| dense', '%.3e'%e.get('value',0), e.get('stage_ms_per_step'))
" || tail -5 $O/bench_ab_$name.err
}
run prev AMC_LIB_PATH=$GRAFT_REPO_ROOT/pycolmap_amd/csrc/_obj/libamc_prev.so
run w8 AMC_MFMA_SHAPE=8
run w4 AMC_MFMA_SHAPE=4
run w8b AMC_MFMA_SHAPE=8
run w4b AMC_MFMA_SHAPE=4
