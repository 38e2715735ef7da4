set -x
export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
timeout -k 10 300 python tools/ab.py --reps 2 xcc= "static=#define MC_EMIT_STATIC 1" "xccwg3=env:MC_EMIT_WG_PER_CU=3" "xccwg6=env:MC_EMIT_WG_PER_CU=6" "xccminw8=#define MC_EMIT_MINW 8" > gpurun_out/r3_ab8.log 2>&1
cat gpurun_out/r3_ab8.log
timeout -k 10 200 python tools/ab.py --reps 1 --bench-args "--grid-res|512" xcc= "static=#define MC_EMIT_STATIC 1" > gpurun_out/r3_ab9.log 2>&1
cat gpurun_out/r3_ab9.log
timeout -k 10 200 python tools/ab.py --reps 1 --bench-args "--mode|isosweep|--steps|30" xcc= "static=#define MC_EMIT_STATIC 1"  > gpurun_out/r3_ab10.log 2>&1
cat gpurun_out/r3_ab10.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3_t3.log 2>&1; tail -3 gpurun_out/r3_t3.log
