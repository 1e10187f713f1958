# timeline of the pipeline leg's last call (amc_match_verify_pairs): kernels, copies, idle gaps
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/ktp; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ktp -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-dense --no-ragged --no-db > /tmp/ktp.json 2> /tmp/ktp.err
k=$(find /tmp/ktp -name "*kernel_trace.csv" | head -1); m=$(find /tmp/ktp -name "*memory_copy_trace.csv" | head -1)
python - <<PY
import csv, collections
ks=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open("$k"))]
ms=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY_"+r["Direction"]) for r in csv.DictReader(open("$m"))] if "$m" else []
ks.sort()
sc=[e for e in ks if "match_mfma_kernel<0" in e[2]]
gg=[e for e in ks if "match_guided_grid_kernel" in e[2]]
end=gg[0][0] if gg else max(e[1] for e in ks)
sc=[e for e in sc if e[0]<end]
ev=sorted(e for e in ks+ms if e[0]>=sc[-3][0]-4_000_000 and e[0]<end)   # the last chained call: its three scans ... its verification
t0=ev[0][0]; t1=max(e[1] for e in ev)
print("window %.1f ms"%((t1-t0)/1e6))
last=t0
for s,e,n in ev:
    if s-last>3e5: print("        idle %.2f ms"%((s-last)/1e6))
    if e-s>3e5: print("%7.1f .. %7.1f  %6.2f ms  %s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n.split("(")[0][:60]))
    last=max(last,e)
PY
tail -1 /tmp/ktp.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['pipeline']['stage_ms_per_step'], d['pipeline']['ms_per_step'])"
