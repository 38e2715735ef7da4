timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gpu82.log 2>&1; tail -2 gpurun_out/gpu82.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke82.log 2>&1; tail -1 gpurun_out/smoke82.log
python bench.py > gpurun_out/bench82.json 2> gpurun_out/bench82.err; python -c "
import json; d=json.loads(open('gpurun_out/bench82.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_one_in_flight'], d['kernel_ms'], d['roofline']['frac'], d['cpu_baseline']['value'])"
python bench.py --no-cpu-baseline --workload gyroid --steps 10 --warmup 2 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gyroid', d['ms_per_step'], d.get('ms_per_step_one_in_flight'), d['kernel_ms'])"
