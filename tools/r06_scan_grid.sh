# round 6: the forward scan with fewer workgroups than CUs (AMC_SCAN_GRID) - how much of its rate does a power-bound
# scan lose per CU it gives up?   bash tools/r06_scan_grid.sh <tag>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; OUT=$O/scan_grid_$TAG.txt; : > $OUT
HEAD="--steps 4 --warmup 1 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged --no-db --no-sift-stats --no-config3"
for g in 256 248 240 224 208 192 160 128 256; do
  AMC_SCAN_GRID=$g python bench.py $HEAD 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('grid', $g, 'ms_per_step', round(d['ms_per_step'],2), 'scan_ms', round(r['avg_kernel_ms'],3), 'frac', round(r['frac'],4), 'per_cu_rate_vs_256', 0)" >> $OUT
done
python - <<PY >> $OUT
import re
rows=[(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"grid (\d+) ms_per_step \S+ scan_ms (\S+)", open("$OUT").read())]
base=[t for g,t in rows if g==256]; b=sum(base)/len(base)
for g,t in rows: print(f"grid {g}: scan {t:.2f} ms = x{t/b:.3f} of the 256-CU time; a CU-bound kernel would take x{256/g:.3f}")
PY
cat $OUT
