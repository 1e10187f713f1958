# A/B timing inside ONE gpurun call (box-to-box clock spread is ~3 %): builds one source file of
# pycolmap_amd/csrc as of git revision $1 and links it with the current objects into
# pycolmap_amd/csrc/_obj/libamc_prev.so (select it with AMC_LIB_PATH; tools/diag_run.sh prev base ...).
#   bash tools/ab_build.sh <rev> [file.hip]      default file: match_mfma.hip
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
F=${2:-match_mfma.hip}
STEM=${F%.hip}
git show $REV:pycolmap_amd/csrc/$F > pycolmap_amd/csrc/_obj/${STEM}_prev.hip
cd pycolmap_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I."
/opt/rocm/bin/hipcc $FLAGS -c _obj/${STEM}_prev.hip -o _obj/${STEM}_prev.o
OBJS=""
for o in amc_api amc_comm match_common match_dot4 match_guided match_mfma tvg_e tvg_fh tvg_e_big tvg_fh_big pose camera; do
  if [ $o = $STEM ]; then OBJS="$OBJS _obj/${o}_prev.o"; else OBJS="$OBJS _obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _obj/libamc_prev.so $OBJS
ls -la _obj/libamc_prev.so
