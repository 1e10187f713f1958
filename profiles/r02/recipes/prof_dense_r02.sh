# round 2: kernel statistics of the dense-overlap match leg (main leg: 1 step, dense leg: 1 step, no warm-up)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02
rm -rf /tmp/prof_dense
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dense -o dense -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline > $R/gpurun_out/r02/dense_under_rocprofv3.log 2>&1
f=$(find /tmp/prof_dense -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep "amc::" $f | cut -c1-60,200- | sed 's/amc::ImageDev.*FinalizeParams)//' > $R/gpurun_out/r02/rocprofv3_kernel_stats_dense_leg.csv
python - <<PY
import csv,sys
for r in csv.reader(open("$f")):
    if r and r[0].startswith(("amc::","void amc::")):
        print(r[0][:50].ljust(52), "calls",r[1],"avg_ms",round(float(r[3])/1e6,3),"min_ms",round(float(r[5])/1e6,3),"max_ms",round(float(r[6])/1e6,3))
PY
tail -c 700 $R/gpurun_out/r02/dense_under_rocprofv3.log
