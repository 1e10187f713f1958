# round 6: long randomised parity runs with fresh seeds (match, verification, guided), the match one also with the
# chain-beside-scan mode on.   bash tools/r06_stress_long.sh <seed>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; S=${1:-7}
timeout 900 python tools/stress_verify.py --rounds 40 --seed $S > $O/stress_verify_seed$S.txt 2>&1; tail -1 $O/stress_verify_seed$S.txt
timeout 500 python tools/stress_match.py --rounds 40 --seed $S > $O/stress_match_seed$S.txt 2>&1; tail -1 $O/stress_match_seed$S.txt
AMC_MATCH_OVERLAP=1 AMC_MATCH_BATCH_ENTRIES=3000 timeout 500 python tools/stress_match.py --rounds 25 --seed $((S+100)) > $O/stress_match_overlap_seed$S.txt 2>&1; tail -1 $O/stress_match_overlap_seed$S.txt
timeout 500 python tools/stress_guided.py 30 $S > $O/stress_guided_seed$S.txt 2>&1; tail -1 $O/stress_guided_seed$S.txt
