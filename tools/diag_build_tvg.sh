# Diagnostic builds of the verification kernel: libamc_tvgdiag<N>.so under pycolmap_amd/csrc/_obj/ (select with
# AMC_LIB_PATH).  N bit 0: the exact fallback of the counting loop's fast test returns 0 (shows whether it is ever taken).
set -e
cd "$(dirname "$0")/../pycolmap_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DAMC_TVG_DIAG=$n -c tvg.hip -o _obj/tvg_diag$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _obj/libamc_tvgdiag$n.so _obj/amc_api.o _obj/match_common.o _obj/match_dot4.o _obj/match_guided.o _obj/match_mfma.o _obj/tvg_diag$n.o _obj/pose.o _obj/camera.o
done
ls -la _obj/*.so
