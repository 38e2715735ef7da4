export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc; export MC_AMD_DEV_LIB=1
one() { python bench.py --no-cpu-baseline --steps 90 --warmup 6 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('classify', k['classify'], 'emit', k['emit'], 'step', d['ms_per_step'], 'one-in-flight', d.get('ms_per_step_one_in_flight'))"; }
for args in "--mode isosweep" "--workload torus"; do
for rep in 1 2; do
echo "[$args] base $(one $args)"
echo "[$args] t42  $(MC_TILE_H=42 one $args)"
echo "[$args] t32  $(MC_TILE_H=32 one $args)"
echo "[$args] t21  $(MC_TILE_H=21 one $args)"
done; done
