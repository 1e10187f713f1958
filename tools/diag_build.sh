# Timing-diagnostic builds of the match kernel (results are WRONG by construction; only their
# kernel time is of interest).  libamc_diag<N>.so under pycolmap_amd/csrc/_obj/, selected with
# AMC_LIB_PATH.  N bit 0: no row-block epilogue; bit 1: no VALU epilogue inside the scan.
set -e
cd "$(dirname "$0")/../pycolmap_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DAMC_DIAG=$n -c match_mfma.hip -o _obj/match_mfma_diag$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _obj/libamc_diag$n.so _obj/amc_api.o _obj/match_common.o _obj/match_dot4.o _obj/match_guided.o _obj/match_mfma_diag$n.o _obj/tvg_e.o _obj/tvg_fh.o _obj/tvg_e_big.o _obj/tvg_fh_big.o _obj/pose.o _obj/camera.o
done
ls -la _obj/*.so
