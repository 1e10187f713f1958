cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
for m in 0 1 2 3 4 0; do LADDER_DATA=$m timeout 120 tools/bin/ubench_ladder 2>&1 | grep -E "data mode|W8x4 chain  rung [03]|W4x8 chain  rung 3|16x16x64 rung [03]" ; done | tee $O/ladder_data_v1.txt
