# A/B timing inside ONE gpurun call (box-to-box clock spread is ~3 %): builds the WHOLE library as of git revision $1
# into pycolmap_amd/csrc/_obj/libamc_prev.so (select it with AMC_LIB_PATH; tools/diag_run.sh prev base ...).
# Unlike tools/ab_build.sh (one source file against the current objects) this survives changes of the internal
# interfaces between the two revisions.
#   bash tools/ab_prev_lib.sh <rev>
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
D=pycolmap_amd/csrc/_obj/prev_src
rm -rf $D && mkdir -p $D/pycolmap_amd/csrc $D/include
for f in $(git ls-tree --name-only $REV pycolmap_amd/csrc/ | grep -E '\.(hip|h)$'); do git show $REV:$f > $D/$f; done
for f in $(git ls-tree --name-only $REV include/); do git show $REV:$f > $D/$f; done
cd $D/pycolmap_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
OBJS=""
for s in *.hip; do /opt/rocm/bin/hipcc $FLAGS -c $s -o ${s%.hip}.o & OBJS="$OBJS ${s%.hip}.o"; done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../../libamc_prev.so $OBJS
ls -la ../../../libamc_prev.so
