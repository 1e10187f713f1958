# round 4, last evidence run: the full -m gpu suite, the default bench line (traffic / executed fields read the re-taken
# counters), and kernel-trace stats of the HEADLINE-ONLY command (the scan's average there is the headline's own).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O; TAG=${1:-v5}
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 900 python bench.py > $O/bench_line_unprofiled_$TAG.json 2> $O/bench_line_unprofiled_$TAG.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt5 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-db > $GRAFT_REPO_ROOT/$O/bench_line_headline_only_under_rocprofv3_$TAG.json 2> /tmp/kt5.err; f=$(find /tmp/kt5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$O/rocprofv3_kernel_stats_headline_only_$TAG.csv && head -4 $f | cut -c1-200)
python - <<PY
import json
d=json.loads(open("$O/bench_line_unprofiled_$TAG.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("headline", "%.3e"%d["value"], round(d["ms_per_step"],1), "frac", round(r["frac"],4), "avg_kernel_ms", round(r["avg_kernel_ms"],2), "traffic", r["traffic"], "launches", r["launches_per_step"])
v=d["verify"]; print("verify", round(v["value"]), round(v["ms_per_step"],1), "executed", (v["roofline"].get("executed") or {}).get("valu_issue_share") if isinstance(v["roofline"].get("executed"), dict) else v["roofline"].get("executed"))
print("dense", "%.3e"%d["dense"]["value"], "ragged", "%.3e"%d["ragged"]["value"], "pipeline", round(d["pipeline"]["ms_per_step"],1), "db", round(d["db"]["wall_s"],3), round(d["db"]["rerun_wall_s"],3))
u=json.loads(open("$O/bench_line_headline_only_under_rocprofv3_$TAG.json").read().strip().splitlines()[-1])
print("headline-only under rocprofv3: avg_kernel_ms", round(u["roofline"]["avg_kernel_ms"],3), "frac", round(u["roofline"]["frac"],4))
PY
