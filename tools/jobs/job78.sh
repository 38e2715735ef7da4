for i in 1 2 3 4; do python bench.py --no-cpu-baseline --workload torus | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('torus', d['ms_per_step'], d.get('ms_per_step_one_in_flight'), d['kernel_ms'])"; done
for i in 1 2 3; do python bench.py --no-cpu-baseline --grid-res 512 --equation "x^2+y^2+z^2-1/(x^2+4)" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rational', d['ms_per_step'], d.get('ms_per_step_one_in_flight'), d['kernel_ms'])"; done
