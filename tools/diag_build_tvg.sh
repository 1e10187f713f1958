# Diagnostic builds of the verification kernel: libamc_tvgdiag<N>.so under pycolmap_amd/csrc/_obj/ (select with
# AMC_LIB_PATH).  N bit 0: the exact fallback of the counting loop's fast test returns 0 (shows whether it is ever taken);
# bit 1: cycle split of the counting loop; bit 3 (N = 8): every homography model is counted by the FP32 pre-filter AND the
# FP64 path, disagreements counted (run with AMC_TVG_PROFILE=1 to print them).
set -e
cd "$(dirname "$0")/../pycolmap_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DAMC_TVG_DIAG=$n -c tvg.hip -o _obj/tvg_diag$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _obj/libamc_tvgdiag$n.so _obj/amc_api.o _obj/match_common.o _obj/match_dot4.o _obj/match_guided.o _obj/match_mfma.o _obj/tvg_diag$n.o _obj/pose.o _obj/camera.o
done
ls -la _obj/*.so
