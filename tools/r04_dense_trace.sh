cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/ktd; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/ktd -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-db > /tmp/ktd.json 2> /tmp/ktd.err
f=$(find /tmp/ktd -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:22]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), ("%.2f ms"%(int(r["TotalDurationNs"])/1e6)).rjust(12))
PY
g=$(find /tmp/ktd -name "*memory_copy_stats.csv" | head -1); [ -n "$g" ] && head -8 $g | cut -c1-200
tail -1 /tmp/ktd.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['dense']['stage_ms_per_step'], d['dense']['ms_per_step'])"
