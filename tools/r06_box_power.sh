# round 6: what the box reports while the forward scan runs (power, clocks, temperature, power cap) - why do the pool's
# boxes differ by 5 % on a power-bound kernel?   bash tools/r06_box_power.sh <tag>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; OUT=$O/box_power_$TAG.txt; : > $OUT; : > $O/box_power_samples_$TAG.txt
echo "== idle" >> $OUT
rocm-smi --showpower --showmaxpower --showclocks --showtemp 2>&1 | grep -v "^=\|^$" | head -40 >> $OUT
HEAD="--steps 40 --warmup 2 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-ragged --no-dense --no-db --no-sift-stats --no-config3 --no-config4"
python bench.py $HEAD > /tmp/bp.json 2>/dev/null &
PID=$!
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -iE "Package Power|sclk|junction" | sed "s/.*: //" | tr "\n" " " >> $O/box_power_samples_$TAG.txt
  echo >> $O/box_power_samples_$TAG.txt
  sleep 0.3
done
echo "== samples above 500 W while bench.py ran (junction C, sclk level, package W)" >> $OUT
awk '{ for (i = 1; i <= NF; ++i) if ($i + 0 > 500 && $i !~ /Mhz/) { print; break } }' $O/box_power_samples_$TAG.txt | head -40 >> $OUT
wait $PID
tail -1 /tmp/bp.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('headline_ms', round(d['ms_per_step'],2), 'scan_ms', round(r['avg_kernel_ms'],3), 'scan_frac', round(r['frac'],4))" >> $OUT
cat $OUT
