export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc; export MC_AMD_DEV_LIB=1
one() { python bench.py --no-cpu-baseline --steps 90 --warmup 6 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('classify', k['classify'], 'emit', k['emit'], 'step', d['ms_per_step'], 'one-in-flight', d.get('ms_per_step_one_in_flight'))"; }
# warm the box into its sustained state first
one > /dev/null; one > /dev/null
for rep in 1 2; do
echo "base                  $(one)"
echo "C 4/CU (pad 10k)      $(MC_LDS_PAD_C=10240 one)"
echo "C 3/CU (pad 18k)      $(MC_LDS_PAD_C=18432 one)"
echo "E 3/CU (pad 16k)      $(MC_LDS_PAD_E=16384 one)"
echo "E 2/CU (pad 32k)      $(MC_LDS_PAD_E=32768 one)"
echo "C 4/CU + E 2/CU       $(MC_LDS_PAD_C=10240 MC_LDS_PAD_E=32768 one)"
echo "C 3/CU + E 2/CU       $(MC_LDS_PAD_C=18432 MC_LDS_PAD_E=32768 one)"
echo "C 4/CU + E 3/CU       $(MC_LDS_PAD_C=10240 MC_LDS_PAD_E=16384 one)"
done
