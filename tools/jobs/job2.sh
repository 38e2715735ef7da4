set -x
export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3_t2.log 2>&1; tail -3 gpurun_out/r3_t2.log
timeout -k 10 400 python tools/ab.py --reps 2 new= "minw8=#define MC_EMIT_MINW 8" "wg3=env:MC_EMIT_WG_PER_CU=3" "wg6=env:MC_EMIT_WG_PER_CU=6" "wg8minw8=env:MC_EMIT_WG_PER_CU=8;#define MC_EMIT_MINW 8" "w4wg8=env:MC_WPB_EMIT=4;env:MC_EMIT_WG_PER_CU=8" "w16wg2=env:MC_WPB_EMIT=16;env:MC_EMIT_WG_PER_CU=2" > gpurun_out/r3_ab5.log 2>&1
cat gpurun_out/r3_ab5.log
timeout -k 10 200 python tools/ab.py --reps 1 --bench-args "--grid-res|512" new= "minw8=#define MC_EMIT_MINW 8" > gpurun_out/r3_ab6.log 2>&1
cat gpurun_out/r3_ab6.log
timeout -k 10 200 python tools/ab.py --reps 1 --bench-args "--mode|isosweep|--steps|30" new= "minw8=#define MC_EMIT_MINW 8" > gpurun_out/r3_ab7.log 2>&1
cat gpurun_out/r3_ab7.log
