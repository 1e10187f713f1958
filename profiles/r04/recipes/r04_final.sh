# round 4 evidence run: counters + kernel-trace stats + ladder + ladder counters + default bench line + stress
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
TAG=${1:-v2}
mkdir -p tools/bin; [ -x tools/bin/ubench_ladder ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/ubench_ladder tools/ubench_ladder.hip
timeout 900 python bench.py > $O/bench_line_unprofiled_$TAG.json 2> $O/bench_line_unprofiled_$TAG.err
bash profiles/r04/recipes/pmc_r04.sh $TAG > $O/pmc_r04_$TAG.log 2>&1
bash profiles/r04/recipes/pmc_dense_r04.sh $TAG > /dev/null 2>&1
timeout 300 tools/bin/ubench_ladder > $O/ladder_$TAG.txt 2>&1
LADDER_DATA=1 timeout 300 tools/bin/ubench_ladder > $O/ladder_siftlike_$TAG.txt 2>&1
bash tools/ladder_pmc.sh $TAG > /dev/null 2>&1
timeout 600 python tools/stress_match.py --rounds 20 > $O/stress_match_$TAG.txt 2>&1
timeout 900 python tools/stress_verify.py > $O/stress_verify_$TAG.txt 2>&1
tail -n 2 $O/stress_match_$TAG.txt; tail -n 2 $O/stress_verify_$TAG.txt
cat $O/pmc_dense_r04_$TAG.txt | cut -c1-220
