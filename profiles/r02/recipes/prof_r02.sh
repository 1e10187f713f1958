# round 2: kernel-trace statistics of the default bench command (all legs), copied to gpurun_out/r02/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02
rm -rf /tmp/prof_r02
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r02 -o bench -- python $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/r02/bench_under_rocprofv3.log 2>&1
f=$(find /tmp/prof_r02 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/r02/rocprofv3_kernel_stats_bench_steps3.csv
head -30 $R/gpurun_out/r02/rocprofv3_kernel_stats_bench_steps3.csv
tail -c 400 $R/gpurun_out/r02/bench_under_rocprofv3.log
