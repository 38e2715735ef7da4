export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc; export MC_AMD_DEV_LIB=1
one() { python bench.py --no-cpu-baseline --steps 90 --warmup 6 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('classify', k['classify'], 'emit', k['emit'], 'step', d['ms_per_step'], 'one-in-flight', d.get('ms_per_step_one_in_flight'))"; }
for rep in 1 2 3; do
echo "base        $(one)"
echo "minw6       $(MC_JIT_EXTRA='#define MC_CLASSIFY_MINW 6' one)"
echo "wpb_c 2     $(MC_WPB_CLASSIFY=2 one)"
echo "wpb_e 4     $(MC_WPB_EMIT=4 one)"
echo "minw6 wpbe4 $(MC_WPB_EMIT=4 MC_JIT_EXTRA='#define MC_CLASSIFY_MINW 6' one)"
done
