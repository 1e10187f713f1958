for n in 4096 16384 65536; do
  timeout 250 python bench.py --images 40 --steps 2 --warmup 1 --no-cpu-baseline --verify-pairs $n 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['verify']; print($n, round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])"
done
