export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
run() { python bench.py --no-cpu-baseline --in-flight 1 --steps 40 --warmup 5 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms']['classify'], d['kernel_ms']['emit'], d['ms_per_step'])"; }
for rep in 1 2 3; do echo "new sphere $(run)"; echo "old sphere $(MC_AMD_DEV_LIB=1 run)"; done
for rep in 1 2; do echo "new 512 $(run --grid-res 512)"; echo "old 512 $(MC_AMD_DEV_LIB=1 run --grid-res 512)"; echo "new torus $(run --workload torus)"; echo "old torus $(MC_AMD_DEV_LIB=1 run --workload torus)"; done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "not random" 2>&1 | tail -3
