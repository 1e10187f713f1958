# A/B timing of the match kernel inside ONE gpurun call (box-to-box clock spread is ~3 %):
# builds match_mfma.hip as of git revision $1 into pycolmap_amd/csrc/_obj/libamc_prev.so.
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
git show $REV:pycolmap_amd/csrc/match_mfma.hip > pycolmap_amd/csrc/_obj/match_mfma_prev.hip
cd pycolmap_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I."
/opt/rocm/bin/hipcc $FLAGS -c _obj/match_mfma_prev.hip -o _obj/match_mfma_prev.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _obj/libamc_prev.so _obj/amc_api.o _obj/match_common.o _obj/match_dot4.o _obj/match_mfma_prev.o _obj/tvg.o _obj/pose.o
ls -la _obj/libamc_prev.so
