export MC_JIT_CACHE=/tmp/jc; mkdir -p /tmp/jc
timeout -k 10 300 python tools/ab.py --reps 3 pre= > gpurun_out/r3_ab14.log 2>&1; cat gpurun_out/r3_ab14.log
timeout -k 10 200 python tools/ab.py --reps 2 --bench-args "--grid-res|512" pre= >> gpurun_out/r3_ab14.log 2>&1; tail -2 gpurun_out/r3_ab14.log
timeout -k 10 200 python tools/ab.py --reps 2 --bench-args "--mode|isosweep|--steps|30" pre= >> gpurun_out/r3_ab14.log 2>&1; tail -2 gpurun_out/r3_ab14.log
