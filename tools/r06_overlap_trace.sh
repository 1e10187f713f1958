# round 6: kernel timeline of one headline step with the chain beside the next scan (rocprofv3 --kernel-trace)
#   bash tools/r06_overlap_trace.sh <tag> [AMC_CHAIN_CUS]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; export AMC_CHAIN_CUS=${2:-8}
export TMPDIR=/tmp
D=/tmp/ovtrace_$TAG; rm -rf $D
HEAD="--steps 2 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-dense --no-db --no-sift-stats --no-config3 --no-config4"
AMC_MATCH_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -- python bench.py $HEAD > $O/overlap_trace_$TAG.log 2>&1
F=$(find $D -name '*kernel_trace.csv' | head -1)
python - "$F" > $O/overlap_timeline_$TAG.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [r for r in rows if r["Kernel_Name"].startswith(("amc::", "void amc::", "__amd_rocclr"))]
import glob, os
for f in glob.glob(os.path.join(os.path.dirname(sys.argv[1]), "*memory_copy_trace.csv")):
    for r in csv.DictReader(open(f)):
        ks.append({"Kernel_Name": "MEMCPY " + r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?")) + " B", "Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"], "Queue_Id": "-", "Grid_Size_X": "-"})
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the third-last forward scan on
fw = [i for i, r in enumerate(ks) if "match_mfma_kernel<0" in r["Kernel_Name"]]
i0 = fw[-3]
# include the packing kernels in front of it
i0 = max(0, i0 - 12)
t0 = int(ks[i0]["Start_Timestamp"])
for r in ks[i0:]:
    n = r["Kernel_Name"].replace("void ", "").replace("amc::", "").split("(")[0]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:9.3f} ms  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6:8.3f} ms  q{r.get("Queue_Id", "?")}  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?"))}  {n}')
PY
cat $O/overlap_timeline_$TAG.txt | head -150
