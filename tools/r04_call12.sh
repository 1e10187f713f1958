cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests/test_match_gpu.py tests/test_multigpu_gpu.py -x -q -m gpu > $O/pytest_match_v3.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_match_v3.log
( time timeout 900 python bench.py > $O/bench_line_unprofiled_v1.json 2> $O/bench_line_unprofiled_v1.err ) 2>&1 | grep real
tail -1 $O/bench_line_unprofiled_v1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))
for k in ('cpu_baseline','verify','pipeline','ragged','dense','db'):
    v=d.get(k,{})
    print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','wall_s','rerun_wall_s','vs_uniform','stats','error','cores') if kk in v})
"
tail -3 $O/bench_line_unprofiled_v1.err
