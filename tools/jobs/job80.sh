export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc; export MC_AMD_DEV_LIB=1
one() { python bench.py --no-cpu-baseline --in-flight 1 --steps 40 --warmup 5 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('classify', k['classify'], 'emit', k['emit'], 'step', d['ms_per_step'])"; }
one > /dev/null; one > /dev/null
for args in "--grid-res 512" "--workload torus" "--mode isosweep" "--slab-of 8"; do
for rep in 1 2; do
for wv in 4 2 1; do
echo "[$args] wpb_c=$wv  $(MC_WPB_CLASSIFY=$wv one $args)"
done; done; done
