# round 2: kernel-trace statistics of tools/guided_bench.py (grid kernel and dense kernel in one run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02
rm -rf /tmp/prof_guided
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_guided -o g -- python $R/tools/guided_bench.py ${1:-64} ${2:-4096} ${3:-0} > $R/gpurun_out/r02/guided_bench_under_rocprofv3.log 2>&1
f=$(find /tmp/prof_guided -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/r02/rocprofv3_kernel_stats_guided_v1.csv
head -12 $R/gpurun_out/r02/rocprofv3_kernel_stats_guided_v1.csv
tail -6 $R/gpurun_out/r02/guided_bench_under_rocprofv3.log
