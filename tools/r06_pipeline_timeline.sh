# round 6: the chained pipeline leg's host timeline (amc_ctx_last_timeline + Python's clock around the call), several
# processes, 10 steps each: where a step's wall time goes beside the kernels' spans.   bash tools/r06_pipeline_timeline.sh <tag> [reps]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O; TAG=${1:-v1}; REPS=${2:-4}; OUT=$O/pipeline_timeline_$TAG.txt; : > $OUT
PIPE="--steps 3 --warmup 1 --no-cpu-baseline --no-dense --verify-pairs 0 --no-ragged --no-db --no-sift-stats --no-config3 --no-config4 --pipeline-steps ${STEPS:-10} --full-line"
for r in $(seq $REPS); do
  python bench.py $PIPE 2>/dev/null | tail -1 | python -c "
import json,sys; p=json.loads(sys.stdin.read())['pipeline']; s=p['stage_ms_per_step']; h=p['host_timeline_ms_per_step']
print('run $r ms_per_step %.2f  device: match %.2f (scan %.2f cross %.2f) verify %.2f (kernels %.2f)  host: c_call %.2f = setup %.2f + match %.2f + close/launch %.2f + verify wait/pack/download %.2f | python %.2f (free previous %.2f) | hand-over hidden %.2f | slowest step %.2f | steps %s' % (
  p['ms_per_step'], s['match_ms'], s['scan_ms'], s['cross_ms'], s['verify_ms'], s['verify_kernel_ms'], h['c_call_ms'], h['verify_setup_ms'], h['match_call_ms'], h['close_and_launch_ms'],
  h['verify_wait_pack_download_ms'], p['ms_per_step'] - h['c_call_ms'], h['python_free_previous_ms'], h['batch_handover_host_ms_hidden'], h['step_ms_max'], h.get('steps_ms')))" >> $OUT
done
cat $OUT
