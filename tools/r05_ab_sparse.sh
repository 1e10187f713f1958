# Same-box A/B of the headline step's non-scan time (cross-check stage, finalize, gaps): AMC_MATCH_PROFILE lines of the
# headline-only command with the previous library (tools/ab_prev_lib.sh <rev>) and the current one.
#   bash tools/r05_ab_sparse.sh [tag] -> gpurun_out/r05/ab_sparse_<tag>.txt
R=${GRAFT_REPO_ROOT:-.}
TAG=${1:-v1}
mkdir -p $R/gpurun_out/r05
OUT=$R/gpurun_out/r05/ab_sparse_$TAG.txt
: > $OUT
ARGS="--steps 6 --warmup 2 --verify-pairs 0 --no-pipeline --no-ragged --no-sift-stats --no-dense --no-db --no-cpu-baseline --no-config3"
for rep in 1 2 3; do
  for which in prev cur; do
    if [ $which = prev ]; then export AMC_LIB_PATH=$R/pycolmap_amd/csrc/_obj/libamc_prev.so; else unset AMC_LIB_PATH; fi
    AMC_MATCH_PROFILE=1 python $R/bench.py $ARGS 2>&1 | grep "amc match profile" | tail -4 | python -c "
import re,sys
rows=[]
for l in sys.stdin:
    m=re.search(r'wall=([\d.]+) ms.*device events ([\d.]+) ms \(scan ([\d.]+), cross ([\d.]+)\)', l)
    if m: rows.append([float(x) for x in m.groups()])
n=len(rows); a=[sum(r[k] for r in rows)/n for k in range(4)]
print('$which rep $rep: wall %.2f device %.2f scan %.2f cross %.2f other-device %.2f (mean of %d calls)' % (a[0],a[1],a[2],a[3],a[1]-a[2]-a[3],n))" >> $OUT
  done
done
unset AMC_LIB_PATH
cat $OUT
