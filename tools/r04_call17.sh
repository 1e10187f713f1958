cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
( time timeout 900 python bench.py --config 4 --steps 2 --warmup 1 > $O/bench_config4_n1_v1.json 2> $O/bench_config4_n1_v1.err ) 2>&1 | grep real
tail -1 $O/bench_config4_n1_v1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config4', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_kernel_ms'), d['config'].get('loop_pairs'))"
( time timeout 900 python bench.py --config 3 --images 250 --steps 2 --warmup 1 > $O/bench_config3_n1_v1.json 2> $O/bench_config3_n1_v1.err ) 2>&1 | grep real
tail -1 $O/bench_config3_n1_v1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config3(250 imgs)', d['value'], d['ms_per_step'], d['roofline']['frac'])"
