# timeline of the dense leg's last call: kernel-trace (no stats), then per-kernel time and idle gaps inside the call
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/ktd; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ktd -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-db > /tmp/ktd.json 2> /tmp/ktd.err
k=$(find /tmp/ktd -name "*kernel_trace.csv" | head -1); m=$(find /tmp/ktd -name "*memory_copy_trace.csv" | head -1)
python - <<PY
import csv, collections
ks=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open("$k"))]
ms=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY_"+r["Direction"]+"_"+str(int(r.get("Bytes",0) or 0)>>20)+"MB") for r in csv.DictReader(open("$m"))] if "$m" else []
ks.sort()
# the dense leg's calls are the last ones: find the last two "finalize" groups; take events after the 3rd-last finalize's end
fin=[e for e in ks if "finalize_kernel" in e[2]]
# dense leg: steps = 2 + warmup 1 -> 3 calls x 2 batches... take the window of the last call: between the end of the finalize before the last two/three finalizes
nb=3 if len(fin)>=3 else 2
# find start: the last call starts at the first seg_count/launch after the previous call's last finalize
cut=fin[-(nb+1)][1] if len(fin)>nb else 0
ev=[e for e in ks+ms if e[0]>=cut]
ev.sort()
t0=ev[0][0]; t1=max(e[1] for e in ev)
tot=collections.Counter(); cnt=collections.Counter()
for s,e,n in ev:
    key=n.split("(")[0][:48]; tot[key]+=e-s; cnt[key]+=1
print("window %.1f ms, events %d"%((t1-t0)/1e6,len(ev)))
for key,v in tot.most_common(16): print(key.ljust(50), cnt[key], "%.2f ms"%(v/1e6))
# union busy time of kernels (not copies)
iv=sorted((s,e) for s,e,n in ev if not n.startswith("COPY_"))
busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print("kernel-busy union %.1f ms, idle %.1f ms"%(busy/1e6,(t1-t0-busy)/1e6))
for s,e,n in ev:
    if e-s>3e5: print("%7.1f .. %7.1f  %6.2f ms  %s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n.split("(")[0][:60]))
PY
tail -1 /tmp/ktd.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['dense']['stage_ms_per_step'], d['dense']['ms_per_step'])"
