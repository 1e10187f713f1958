cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout 300 tools/bin/ubench_ladder > $O/ladder_v3.txt 2>&1; echo "ladder rc=$?"
cat $O/ladder_v3.txt
bash tools/ladder_pmc.sh v2 > /dev/null 2>&1
grep -A40 "=== derived" $O/ladder_pmc_v2.txt
