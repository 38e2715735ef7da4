export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc; export MC_AMD_DEV_LIB=1
one() { python bench.py --no-cpu-baseline --in-flight 1 --steps 40 --warmup 5 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('classify', k['classify'], 'emit', k['emit'], 'step', d['ms_per_step'])"; }
for args in "--mode isosweep" "--workload torus" "--grid-res 512" ""; do
for rep in 1 2; do
echo "[$args] base      $(one $args)"
echo "[$args] minw6     $(MC_JIT_EXTRA='#define MC_CLASSIFY_MINW 6' one $args)"
echo "[$args] t21       $(MC_TILE_H=21 one $args)"
echo "[$args] t32       $(MC_TILE_H=32 one $args)"
echo "[$args] minw6 t21 $(MC_TILE_H=21 MC_JIT_EXTRA='#define MC_CLASSIFY_MINW 6' one $args)"
echo "[$args] minw6 t32 $(MC_TILE_H=32 MC_JIT_EXTRA='#define MC_CLASSIFY_MINW 6' one $args)"
done; done
