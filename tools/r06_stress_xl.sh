# round 6: the randomised parity tools at ten times their usual length, fresh seeds.   bash tools/r06_stress_xl.sh <seed>
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; S=${1:-8}
timeout 1500 python tools/stress_verify.py --rounds 300 --seed $S > $O/stress_verify_xl_seed$S.txt 2>&1; tail -1 $O/stress_verify_xl_seed$S.txt
timeout 1200 python tools/stress_match.py --rounds 300 --seed $S > $O/stress_match_xl_seed$S.txt 2>&1; tail -1 $O/stress_match_xl_seed$S.txt
timeout 1200 python tools/stress_guided.py 200 $S > $O/stress_guided_xl_seed$S.txt 2>&1; tail -1 $O/stress_guided_xl_seed$S.txt
