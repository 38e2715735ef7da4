export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3_t12.log 2>&1; tail -3 gpurun_out/r3_t12.log
timeout -k 10 300 python tools/ab.py --reps 2 share= "w4=env:MC_WPB_EMIT=4" "w16=env:MC_WPB_EMIT=16" > gpurun_out/r3_ab12.log 2>&1; cat gpurun_out/r3_ab12.log
timeout -k 10 200 python tools/ab.py --reps 1 --bench-args "--grid-res|512" share= "w16=env:MC_WPB_EMIT=16" >> gpurun_out/r3_ab12.log 2>&1; tail -2 gpurun_out/r3_ab12.log
timeout -k 10 200 python tools/ab.py --reps 1 --bench-args "--mode|isosweep|--steps|30" share= "w16=env:MC_WPB_EMIT=16" >> gpurun_out/r3_ab12.log 2>&1; tail -2 gpurun_out/r3_ab12.log
