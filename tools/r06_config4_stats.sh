# round 6: per-kernel time of BASELINE configs[4] at N = 1 (rocprofv3 --kernel-trace --stats)   bash tools/r06_config4_stats.sh <tag>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}
export TMPDIR=/tmp
D=/tmp/c4stats_$TAG; rm -rf $D
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > $O/config4_under_rocprofv3_$TAG.log 2>&1
F=$(find $D -name '*kernel_stats.csv' | head -1)
cp $F $O/rocprofv3_kernel_stats_config4_$TAG.csv
python - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "amc::" in r["Name"] or "rocclr" in r["Name"]]
for r in rows[:24]:
    n = r["Name"].replace("void ", "").replace("amc::", "").split("(")[0]
    print(f'{n:45s} calls {int(r["Calls"]):6d}  total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms  avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
tail -1 $O/config4_under_rocprofv3_$TAG.log | cut -c1-300
