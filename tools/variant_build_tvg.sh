# A/B builds of the verification kernels inside ONE gpurun call: libamc_<name>.so under pycolmap_amd/csrc/_obj/ with
# extra -D flags on tvg_e.hip, tvg_fh.hip and amc_api.hip (the host sizes the launches from the same constants);
# select with AMC_LIB_PATH.   bash tools/variant_build_tvg.sh fh3 "-DAMC_FH_WAVES=3"
set -e
cd "$(dirname "$0")/../pycolmap_amd/csrc"
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include $*"
for f in tvg_e tvg_fh amc_api; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o _obj/${f}_$NAME.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _obj/libamc_$NAME.so _obj/amc_api_$NAME.o _obj/amc_comm.o _obj/match_common.o _obj/match_dot4.o _obj/match_guided.o _obj/match_mfma.o _obj/tvg_e_$NAME.o _obj/tvg_fh_$NAME.o _obj/tvg_e_big.o _obj/tvg_fh_big.o _obj/pose.o _obj/camera.o
ls -la _obj/libamc_$NAME.so
