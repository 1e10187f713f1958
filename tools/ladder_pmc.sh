# PMC passes over tools/bin/ubench_ladder: matrix-pipe busy cycles and the clock each rung holds.
#   bash tools/ladder_pmc.sh [tag]   -> gpurun_out/r04/ladder_pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-v1}
mkdir -p $R/gpurun_out/r04
OUT=$R/gpurun_out/r04/ladder_pmc_$TAG.txt
: > $OUT
run() {
  tag=$1; shift
  for try in 1 2 3; do
    rm -rf /tmp/lad_$tag
    timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/lad_$tag -- $R/tools/bin/ubench_ladder > /tmp/lad_$tag.log 2>&1
    rc=$?
    db=$(find /tmp/lad_$tag -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && break
  done
  echo "rc=$rc tries=$try" >> $OUT
  echo "=== pass $tag: $@" >> $OUT
  [ -n "$db" ] && python $R/tools/pmc_summary.py $db ladder | grep -E "calls=|n=" | grep -v "pmc tables" >> $OUT
  grep rung /tmp/lad_$tag.log >> $OUT
}
run a SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU
run g GRBM_GUI_ACTIVE
python - "$OUT" <<'PY' >> $OUT
import re, sys, collections
txt = open(sys.argv[1]).read()
tot = {}; cnt = collections.defaultdict(dict)
for line in txt.splitlines():
    m = re.match(r"\s+(\S.*?)\s+calls=(\d+)\s+total=(\S+)\s+avg=(\S+)", line)
    if m and "ladder" in m.group(1): tot[m.group(1).strip()] = float(m.group(3))
    m = re.match(r"\s+(\S.*?)\s+(SQ_\w+|GRBM_\w+)\s+n=(\d+)\s+sum=(\S+)", line)
    if m: cnt[m.group(1).strip()][m.group(2)] = float(m.group(4))
print("=== derived (per kernel, all its dispatches): clock = GRBM_GUI_ACTIVE / 8 XCDs / time; pipe busy = MFMA_BUSY / (1024 SIMDs x cycles)")
for k, c in cnt.items():
    t = None
    for kk, v in tot.items():
        if kk[:40] == k[:40]: t = v
    if not t or "GRBM_GUI_ACTIVE" not in c: continue
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    ghz = cycles / t / 1e3   # top_kernels totals are in microseconds
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cycles)
    print("  %-50s time %.3f ms clock %.3f GHz pipe-busy %.3f" % (k[:50], t / 1e3, ghz, busy))
PY
cat $OUT
