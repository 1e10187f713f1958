# Builds pycolmap_amd/csrc/_obj/libamc_var_<name>.so: the current library with match_mfma.hip compiled with extra flags
# (A/B inside one gpurun call; select with AMC_LIB_PATH).   bash tools/var_build.sh <name> <flags...>
set -e
cd "$(dirname "$0")/../pycolmap_amd/csrc"
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include"
/opt/rocm/bin/hipcc $FLAGS "$@" -c match_mfma.hip -o _obj/match_mfma_var_$NAME.o
OBJS=""
for o in amc_api match_common match_dot4 match_guided tvg_e tvg_fh tvg_e_big tvg_fh_big pose camera; do OBJS="$OBJS _obj/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _obj/libamc_var_$NAME.so $OBJS _obj/match_mfma_var_$NAME.o
