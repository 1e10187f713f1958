cd $GRAFT_REPO_ROOT
bash tools/pmc_tvg_r04.sh v1 > /dev/null 2>&1
python tools/pmc_tvg_json.py gpurun_out/r04/pmc_tvg_r04_v1.txt gpurun_out/r04/pmc_tvg_r04.json
bash tools/pmc_dense_r04.sh v1 > /dev/null 2>&1
cat gpurun_out/r04/pmc_dense_r04_v1.txt | cut -c1-200
for m in 0 1 2 3; do echo "LADDER_DATA=$m"; LADDER_DATA=$m timeout 120 tools/bin/ubench_ladder 2>&1 | grep -E "W8x4 chain  rung [03]|W4x8 chain  rung 3|W8 16x16x64 rung [03]"; done > gpurun_out/r04/ladder_data_modes_v1.txt
