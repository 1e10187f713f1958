cd $GRAFT_REPO_ROOT
PIPE="--steps 3 --warmup 1 --no-cpu-baseline --no-dense --verify-pairs 0 --no-ragged --no-db --no-sift-stats --no-config3 --no-config4 --pipeline-steps 6 --full-line"
pipe() { python bench.py $PIPE 2>/dev/null | tail -1 | python -c "import json,sys; p=json.loads(sys.stdin.read())['pipeline']; s=p['stage_ms_per_step']; print('$1', round(p['ms_per_step'],2), 'verify', round(s['verify_ms'],2), 'kernels', round(s['verify_kernel_ms'],2))"; }
for r in 1 2; do
AMC_PIPELINE_SERIAL=1 AMC_TVG_SLICES=1 pipe serial_1slice
AMC_PIPELINE_SERIAL=1 AMC_TVG_SLICES=2 AMC_TVG_MIN_PER_SLICE=1 pipe serial_2slices
pipe shipped
done
