# round 6 evidence run, in sections so that every gpurun call stays bounded:
#   bash tools/r06_final.sh suite <tag>   the full -m gpu suite, the default bench line, kernel-trace stats of the
#                                         headline-only command, the two stress tools
#   bash tools/r06_final.sh pmc <tag>     counters: the scan (SQ + HBM -> pmc_hbm_r06.json), the verification kernels
#                                         (-> pmc_tvg_r06.json), the dense set's cross-check stage (FETCH_SIZE)
# Outputs under gpurun_out/r06/; what is to be judged is copied to profiles/r06/ by hand.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; SEC=${1:-suite}; TAG=${2:-v1}
NOLEGS="--no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3 --no-config4"
if [ $SEC = suite ]; then
  timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu_$TAG.log | tail -2
  timeout 700 python bench.py --steps 20 --warmup 5 --detail-json $O/bench_detail_$TAG.json > $O/bench_line_unprofiled_$TAG.json 2> $O/bench_line_unprofiled_$TAG.err; echo "bench rc=$?"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt5 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 $NOLEGS > $GRAFT_REPO_ROOT/$O/bench_line_headline_only_under_rocprofv3_$TAG.json 2> /tmp/kt5.err; f=$(find /tmp/kt5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$O/rocprofv3_kernel_stats_headline_only_$TAG.csv && head -4 $f | cut -c1-200)
  timeout 400 python bench.py --config 4 --steps 2 --warmup 1 --full-line > $O/bench_config4_n1_$TAG.json 2> /dev/null; echo "config4 rc=$?"
  timeout 300 python tools/stress_match.py > $O/stress_match_$TAG.txt 2>&1; tail -2 $O/stress_match_$TAG.txt
  timeout 300 python tools/stress_verify.py > $O/stress_verify_$TAG.txt 2>&1; tail -2 $O/stress_verify_$TAG.txt
  python - <<PY
import json
d=json.loads(open("$O/bench_line_unprofiled_$TAG.json").read().strip().splitlines()[-1])
print("line bytes", len(json.dumps(d)))
r=d["roofline"]; print("headline", "%.4e"%d["value"], round(d["ms_per_step"],2), "frac", round(r["frac"],4), "whole", round(r["whole_step_frac"],4), "avg_kernel_ms", round(r["avg_kernel_ms"],2), "traffic", r["traffic"])
v=d["verify"]; print("verify", round(v["value"]), round(v["ms_per_step"],1), "kernels", round(v["kernel_ms_per_step"],1), "traffic", v["roofline"].get("traffic"), "lanes", v["roofline"].get("lane_utilisation"))
p=d["pipeline"]; print("pipeline", round(p["value"]), p["ms_per_step"], "scan", p["scan_ms"], "non_scan", p["non_scan_ms"], "verify", p["verify_ms"])
c4=d.get("config4") or {}; print("config4", c4.get("value"), c4.get("ms_per_step"), c4.get("scan_frac_of_int8_peak"), c4.get("parts"))
print("dense", "%.3e"%d["dense"]["value"], d["dense"]["resolve_select_reverse_scan_ms"], "ragged", "%.3e"%d["ragged"]["value"], d["ragged"].get("vs_uniform"), "sift", "%.3e"%d["sift_stats"]["value"], "db", d["db"].get("wall_s"), "config3", "%.3e"%d["config3_value"], d["config3_exchange_ms_per_step"])
PY
fi
if [ $SEC = pmc ]; then
  timeout 800 bash tools/pmc_r06.sh $TAG > $O/pmc_r06_$TAG.log 2>&1; grep "rc=" $O/pmc_match_r06_$TAG.txt $O/pmc_hbm_r06_$TAG.txt
  timeout 600 bash tools/pmc_tvg_r06.sh $TAG > /dev/null 2>&1; grep "rc=" $O/pmc_tvg_r06_$TAG.txt; python tools/pmc_tvg_json.py $O/pmc_tvg_r06_$TAG.txt $O/pmc_tvg_r06.json | tail -3
  timeout 800 bash tools/pmc_dense_r06.sh $TAG > /dev/null 2>&1; cat $O/pmc_dense_r06_$TAG.txt | cut -c1-200
fi
