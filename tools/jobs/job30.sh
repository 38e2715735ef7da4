export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
for t in 63 0; do
  export MC_FORCE63=$t
  for a in "--slab-of 8 --steps 200 --warmup 20" "--slab-of 4 --steps 100 --warmup 10" "--mode isosweep --steps 60 --warmup 5" "--workload torus"; do
    python bench.py --no-cpu-baseline $a | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ['MC_FORCE63'], '$a', d['ms_per_step'], d.get('ms_per_step_one_in_flight'), d['kernel_ms']['classify'], d['kernel_ms']['emit'])"
  done
done
