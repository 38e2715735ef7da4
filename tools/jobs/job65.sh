export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc; export MC_AMD_DEV_LIB=1
one() { python bench.py --no-cpu-baseline --in-flight 1 --steps 40 --warmup 5 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('classify', k['classify'], 'emit', k['emit'], 'step', d['ms_per_step'])"; }
one > /dev/null; one > /dev/null
for args in "--mode isosweep" "--workload torus" "--grid-res 512" ""; do
for rep in 1 2; do
for late in 0 "10,21" "5,21" "20,21" "10,32" "10,10" "4,32"; do
echo "[$args] late=$late  $(MC_LATE=$late one $args)"
done; done; done
echo "== quick parity (default rule on)"; MC_AMD_DEV_LIB=0 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/parity65.log 2>&1; tail -2 gpurun_out/parity65.log
