# round 3: kernel-trace statistics of the default bench command (all legs), copied to gpurun_out/r03/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
rm -rf /tmp/prof_r03
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03 -o bench -- python $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/r03/bench_under_rocprofv3.log 2>&1
f=$(find /tmp/prof_r03 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/r03/rocprofv3_kernel_stats_bench_steps3.csv
head -16 $R/gpurun_out/r03/rocprofv3_kernel_stats_bench_steps3.csv
tail -c 300 $R/gpurun_out/r03/bench_under_rocprofv3.log
