# round 4, second evidence run (after the verification / host-layer work): default bench line, kernel-trace stats of the
# same command, the verification kernels' counters, stress runs.   bash profiles/r04/recipes/r04_final2.sh v4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
TAG=${1:-v4}
timeout 900 python bench.py > $O/bench_line_unprofiled_$TAG.json 2> $O/bench_line_unprofiled_$TAG.err
tail -c 600 $O/bench_line_unprofiled_$TAG.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt4 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_line_under_rocprofv3_$TAG.json 2> /tmp/kt4.err; f=$(find /tmp/kt4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$O/rocprofv3_kernel_stats_bench_steps3_$TAG.csv && head -8 $f | cut -c1-160)
bash profiles/r04/recipes/pmc_tvg_r04.sh v2 > /dev/null 2>&1
timeout 600 python tools/stress_match.py --rounds 20 > $O/stress_match_$TAG.txt 2>&1
timeout 900 python tools/stress_verify.py > $O/stress_verify_$TAG.txt 2>&1
tail -n 1 $O/stress_match_$TAG.txt; tail -n 1 $O/stress_verify_$TAG.txt
python - <<PY
import json
d=json.loads(open("$O/bench_line_unprofiled_$TAG.json").read().strip().splitlines()[-1])
print("headline", "%.3e"%d["value"], round(d["ms_per_step"],1), "frac", round(d["roofline"]["frac"],4))
for k in ("ragged","dense"):
    print(k, "%.3e"%d[k]["value"], round(d[k]["ms_per_step"],1))
v=d["verify"]; print("verify", round(v["value"]), round(v["ms_per_step"],1), round(v["kernel_ms_per_step"],1))
p=d["pipeline"]; print("pipeline", round(p["ms_per_step"],1), p.get("stage_ms_per_step"))
b=d["db"]; print("db", round(b["wall_s"],3), round(b["rerun_wall_s"],3))
c=d["cpu_baseline"]; print("cpu", "%.2e"%c["value"], "%.2e"%c["optimised"]["value"], c["default_cpu_matcher"])
PY
