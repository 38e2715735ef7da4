export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
one() { python bench.py --no-cpu-baseline "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('classify', k['classify'], 'emit', k['emit'], 'step', d['ms_per_step'], 'one-in-flight', d.get('ms_per_step_one_in_flight'))"; }
for rep in 1 2 3 4; do
echo "warmup 10   $(one --steps 100 --warmup 10)"
echo "warmup 300  $(one --steps 100 --warmup 300)"
echo "warmup 1000 $(one --steps 100 --warmup 1000)"
done
echo "steps 1000 warmup 10 $(one --steps 1000 --warmup 10)"
echo "steps 1000 warmup 10 $(one --steps 1000 --warmup 10)"
