export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc; export MC_AMD_DEV_LIB=1
run() { python bench.py --no-cpu-baseline --steps 60 --warmup 6 "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_one_in_flight'), d['kernel_ms']['classify'], d['kernel_ms']['emit'])"; }
echo "2 in flight, no masks: $(run --in-flight 2)"
F=ffffffff; Z=00000000
echo "2 in flight, halves (low 128 / high 128 bits): $(MC_CU_MASK_0=$F,$F,$F,$F,$Z,$Z,$Z,$Z MC_CU_MASK_1=$Z,$Z,$Z,$Z,$F,$F,$F,$F run --in-flight 2)"
echo "2 in flight, interleaved words: $(MC_CU_MASK_0=$F,$Z,$F,$Z,$F,$Z,$F,$Z MC_CU_MASK_1=$Z,$F,$Z,$F,$Z,$F,$Z,$F run --in-flight 2)"
echo "2 in flight, interleaved bits: $(MC_CU_MASK_0=55555555,55555555,55555555,55555555,55555555,55555555,55555555,55555555 MC_CU_MASK_1=aaaaaaaa,aaaaaaaa,aaaaaaaa,aaaaaaaa,aaaaaaaa,aaaaaaaa,aaaaaaaa,aaaaaaaa run --in-flight 2)"
echo "1 in flight, half the chip (low 128 bits): $(MC_CU_MASK_0=$F,$F,$F,$F,$Z,$Z,$Z,$Z run --in-flight 1)"
echo "1 in flight, every other bit: $(MC_CU_MASK_0=55555555,55555555,55555555,55555555,55555555,55555555,55555555,55555555 run --in-flight 1)"
