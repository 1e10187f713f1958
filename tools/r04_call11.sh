cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
run() {
  name=$1; shift
  env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged > $O/bench_bt_$name.json 2> $O/bench_bt_$name.err
  tail -1 $O/bench_bt_$name.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; e=d.get('dense',{})
print('$name', 'ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'kern_ms', round(r['avg_kernel_ms'],2), r['launches_per_step'], '| dense', '%.3e'%e.get('value',0), round(e.get('ms_per_step',0),1), e.get('stage_ms_per_step'))
" || tail -5 $O/bench_bt_$name.err
}
run b2
run b4 AMC_MATCH_BATCH_ENTRIES=134217728
run b8 AMC_MATCH_BATCH_ENTRIES=67108864
run b16 AMC_MATCH_BATCH_ENTRIES=33554432
run b2b
