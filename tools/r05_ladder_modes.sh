# Operand-encoding sweep on the bottom-up ladder: does feeding one or both sides of the int8 MFMA
# in raw / small-zero-point form hold more clock than the 0x80 zero point?  Y = streamed side (A operand, LDS),
# X = resident side (B operand, registers).  bash tools/r05_ladder_modes.sh [tag] -> gpurun_out/r05/ladder_modes_<tag>.txt
R=${GRAFT_REPO_ROOT:-.}
TAG=${1:-v1}
mkdir -p $R/gpurun_out/r05
OUT=$R/gpurun_out/r05/ladder_modes_$TAG.txt
: > $OUT
for rep in 1 2; do
  for yx in "1 1" "2 1" "5 1" "6 1" "2 2" "1 2" "0 0" "1 1"; do
    set -- $yx
    echo "=== rep $rep: LADDER_DATA=$1 (Y) LADDER_DATA_X=$2 (X)" >> $OUT
    LADDER_QUICK=1 LADDER_DATA=$1 LADDER_DATA_X=$2 timeout 120 $R/tools/bin/ubench_ladder >> $OUT 2>&1
  done
done
cat $OUT
