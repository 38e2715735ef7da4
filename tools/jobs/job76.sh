echo "== full gpu suite"; timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gpu76.log 2>&1; tail -2 gpurun_out/gpu76.log
echo "== bench torus default"; python bench.py --no-cpu-baseline --workload torus | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_one_in_flight'), d['kernel_ms'])"
echo "== bench sphere default"; python bench.py --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_one_in_flight'), d['kernel_ms'])"
echo "== bench gyroid"; python bench.py --no-cpu-baseline --workload gyroid --steps 10 --warmup 2 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_one_in_flight'), d['kernel_ms'])"
