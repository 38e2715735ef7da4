export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
for d in 1 2 3 4 5; do python bench.py --no-cpu-baseline --in-flight $d --steps 60 --warmup 6 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in_flight', d['config']['in_flight'], 'ms_per_step', d['ms_per_step'], 'one', d.get('ms_per_step_one_in_flight'))"; done
for d in 1 2 3 4; do python bench.py --no-cpu-baseline --in-flight $d --steps 200 --warmup 20 --slab-of 8 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slab8 in_flight', d['config']['in_flight'], 'ms_per_step', d['ms_per_step'], 'one', d.get('ms_per_step_one_in_flight'))"; done
