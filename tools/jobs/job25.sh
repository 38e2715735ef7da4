export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
python bench.py > gpurun_out/r03_bench_sphere1024.json 2> gpurun_out/r03_bench_sphere1024.err; tail -c 2500 gpurun_out/r03_bench_sphere1024.json
python bench.py --no-cpu-baseline --grid-res 2000 --steps 20 --warmup 3 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2001^3', d['value'], d['ms_per_step'], d.get('ms_per_step_one_in_flight'), d['kernel_ms'], d['roofline']['kernel'], d['roofline']['frac'], d['second_roofline']['frac'], d['config']['triangles'])"
python bench.py --mode isosweep > gpurun_out/r03_bench_goursat512.json; python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_goursat512.json').read().strip().splitlines()[-1]); print('goursat', d['value'], d['ms_per_step'], d['ms_per_step_one_in_flight'], d['kernel_ms'])"
python bench.py --no-cpu-baseline --workload torus > gpurun_out/r03_bench_torus512.json; python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_torus512.json').read().strip().splitlines()[-1]); print('torus', d['value'], d['ms_per_step'], d['ms_per_step_one_in_flight'], d['kernel_ms'])"
python bench.py --no-cpu-baseline --workload gyroid --steps 20 --warmup 3 > gpurun_out/r03_bench_gyroid1024.json; python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_gyroid1024.json').read().strip().splitlines()[-1]); print('gyroid', d['value'], d['ms_per_step'], d['ms_per_step_one_in_flight'], d['kernel_ms'])"
